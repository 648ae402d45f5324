// host cost of the pieces of a cross-stream dependency: hipEventRecord, hipStreamWaitEvent, and a launch on another stream
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void empty_kernel() {}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define Q(x) (void)(x)
int main() {
    hipStream_t s, s2; Q(hipStreamCreate(&s)); Q(hipStreamCreate(&s2));
    hipEvent_t e; Q(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const int N = 5000;
    for (int i = 0; i < 1000; i++) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s); hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s2); }
    Q(hipDeviceSynchronize());
    double tl = 0, tr = 0, tw = 0, tl2 = 0;
    for (int i = 0; i < N; i++) {
        double a = now();
        for (int j = 0; j < 4; j++) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s);
        double b = now();
        Q(hipEventRecord(e, s));
        double c = now();
        Q(hipStreamWaitEvent(s2, e, 0));
        double d = now();
        hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s2);
        double f = now();
        tl += b - a; tr += c - b; tw += d - c; tl2 += f - d;
    }
    Q(hipDeviceSynchronize());
    printf("4 launches on s: %.2f us | hipEventRecord: %.2f us | hipStreamWaitEvent: %.2f us | launch on s2: %.2f us\n",
           tl / N * 1e6, tr / N * 1e6, tw / N * 1e6, tl2 / N * 1e6);
    // alternate launches between two streams without events
    double t0 = now();
    for (int i = 0; i < N; i++) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s); hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s2); }
    double t1 = now();
    Q(hipDeviceSynchronize());
    printf("alternating two streams, no events: %.2f us/launch\n", (t1 - t0) / (2 * N) * 1e6);
    // events with timing enabled (torch.cuda.Event(enable_timing=False) is the default; for reference)
    hipEvent_t et; Q(hipEventCreate(&et));
    t0 = now();
    for (int i = 0; i < N; i++) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s); Q(hipEventRecord(et, s)); }
    t1 = now();
    Q(hipDeviceSynchronize());
    printf("launch + timing-event record on one stream: %.2f us/pair\n", (t1 - t0) / N * 1e6);
    t0 = now();
    for (int i = 0; i < N; i++) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s); Q(hipEventRecord(e, s)); }
    t1 = now();
    Q(hipDeviceSynchronize());
    printf("launch + no-timing-event record on one stream: %.2f us/pair\n", (t1 - t0) / N * 1e6);
    // same-stream wait (a wait on an event of the same stream)
    t0 = now();
    for (int i = 0; i < N; i++) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s); Q(hipEventRecord(e, s)); Q(hipStreamWaitEvent(s, e, 0)); }
    t1 = now();
    Q(hipDeviceSynchronize());
    printf("launch + record + same-stream wait: %.2f us/triple\n", (t1 - t0) / N * 1e6);
    return 0;
}
