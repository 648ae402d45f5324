// host cost of one kernel launch on this box: N launches of an empty kernel (and of one with a 200-byte argument struct),
// enqueue time only, then the drain time
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { char b[200]; };
__global__ void empty_kernel() {}
__global__ void big_arg_kernel(Big a, int* p) { if (p && a.b[0] == 77) *p = 1; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s; hipStreamCreate(&s);
    const int N = 20000;
    for (int i = 0; i < 1000; i++) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s);
    hipStreamSynchronize(s);
    double t0 = now();
    for (int i = 0; i < N; i++) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s);
    double t1 = now();
    hipStreamSynchronize(s);
    double t2 = now();
    printf("empty kernel: enqueue %.2f us/launch, drained at %.2f us/launch\n", (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6);
    Big b = {};
    t0 = now();
    for (int i = 0; i < N; i++) hipLaunchKernelGGL(big_arg_kernel, dim3(1), dim3(64), 0, s, b, (int*)nullptr);
    t1 = now();
    hipStreamSynchronize(s);
    t2 = now();
    printf("200-byte args: enqueue %.2f us/launch, drained at %.2f us/launch\n", (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6);
    // alternating between two streams with an event dependency every 4 launches (the wgrad side-stream pattern)
    hipStream_t s2; hipStreamCreate(&s2);
    hipEvent_t e; hipEventCreateWithFlags(&e, hipEventDisableTiming);
    t0 = now();
    for (int i = 0; i < N; i++) {
        hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s);
        if ((i & 3) == 3) { hipEventRecord(e, s); hipStreamWaitEvent(s2, e, 0); hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s2); }
    }
    t1 = now();
    hipDeviceSynchronize();
    t2 = now();
    printf("two streams, event every 4 launches: enqueue %.2f us/launch, drained at %.2f us/launch (%d launches)\n",
           (t1 - t0) / (N * 1.25) * 1e6, (t2 - t0) / (N * 1.25) * 1e6, (int)(N * 1.25));
    return 0;
}
