// do hipLaunchKernel calls of different host threads (own stream each) overlap?  hipcc --offload-arch=gfx950 -O2 launch_mt.hip -o launch_mt -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
struct Big { char b[256]; };
__global__ void k(Big a, int* out) { if (out && threadIdx.x == 1000) out[0] = a.b[0]; }
int main() {
    const int iters = 20000;
    for (int nt : {1, 2, 4, 8}) {
        std::vector<hipStream_t> st(nt);
        for (auto& s : st) hipStreamCreate(&s);
        auto work = [&](int i) {
            Big a = {};
            void* args[2]; int* out = nullptr; args[0] = &a; args[1] = &out;
            for (int j = 0; j < iters; j++) hipLaunchKernel((const void*)k, dim3(1), dim3(64), args, 0, st[i]);
        };
        work(0); hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int i = 0; i < nt; i++) th.emplace_back(work, i);
        for (auto& t : th) t.join();
        auto t1 = std::chrono::steady_clock::now();
        hipDeviceSynchronize();
        auto t2 = std::chrono::steady_clock::now();
        double us = std::chrono::duration<double, std::micro>(t1 - t0).count();
        double us2 = std::chrono::duration<double, std::micro>(t2 - t0).count();
        printf("%d threads x %d launches: %.2f us per launch per thread (host), %.2f us per launch aggregate; drained after %.2f us per launch aggregate\n",
               nt, iters, us / iters, us / iters / nt, us2 / iters / nt);
        for (auto& s : st) hipStreamDestroy(s);
    }
    return 0;
}
