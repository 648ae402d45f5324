// host cost of a launch against the size of its by-value argument struct
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
template <int B> struct Arg { char b[B]; };
template <int B> __global__ void k(Arg<B> a, int* p) { if (p && a.b[0] == 77) *p = 1; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <int B> void run(hipStream_t s) {
    Arg<B> a = {};
    const int N = 20000;
    for (int i = 0; i < 500; i++) hipLaunchKernelGGL(k<B>, dim3(1), dim3(64), 0, s, a, (int*)nullptr);
    (void)hipStreamSynchronize(s);
    double t0 = now();
    for (int i = 0; i < N; i++) hipLaunchKernelGGL(k<B>, dim3(1), dim3(64), 0, s, a, (int*)nullptr);
    double t1 = now();
    (void)hipStreamSynchronize(s);
    double t2 = now();
    printf("%4d-byte struct: enqueue %.2f us/launch, drained at %.2f us/launch\n", B, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6);
}
int main() {
    hipStream_t s; (void)hipStreamCreate(&s);
    run<8>(s); run<32>(s); run<64>(s); run<96>(s); run<128>(s); run<160>(s); run<200>(s); run<256>(s); run<400>(s); run<1024>(s);
    run<200>(s); run<8>(s);
    return 0;
}
