// GPU-side cost of an event record in a chain of small dependent kernels, by event flags (the chain is enqueued behind a
// 60 ms kernel, so the host is never the limit)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(long long cycles, float* p) { long long t0 = wall_clock64(); while (wall_clock64() - t0 < cycles) {} if (p && cycles < 0) *p = 1.f; }
__global__ void touch(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define Q(x) (void)(x)
static void run(const char* name, unsigned flags, bool with_wait, float* buf, int n) {
    hipStream_t s, s2; Q(hipStreamCreate(&s)); Q(hipStreamCreate(&s2));
    hipEvent_t e; Q(hipEventCreateWithFlags(&e, flags));
    hipEvent_t t0, t1; Q(hipEventCreate(&t0)); Q(hipEventCreate(&t1));
    const int N = 2000;
    for (int rep = 0; rep < 2; rep++) {
        Q(hipDeviceSynchronize());
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 6000000LL, (float*)nullptr);     // 60 ms at 100 MHz: the host enqueues the chain behind it
        Q(hipEventRecord(t0, s));
        for (int i = 0; i < N; i++) {
            hipLaunchKernelGGL(touch, dim3(n / 256), dim3(256), 0, s, buf, n);
            if (flags != 0xffffffffu) {
                Q(hipEventRecord(e, s));
                if (with_wait) { Q(hipStreamWaitEvent(s2, e, 0)); hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, s2, buf + n, 64); }
            }
        }
        Q(hipEventRecord(t1, s));
        Q(hipDeviceSynchronize());
        float ms = 0; Q(hipEventElapsedTime(&ms, t0, t1));
        if (rep == 1) printf("%-58s %.2f us per kernel on the main stream\n", name, ms * 1e3 / N);
    }
    Q(hipStreamDestroy(s)); Q(hipStreamDestroy(s2));
}
int main() {
    const int n = 1 << 20;          // 4 MB: lives in L2 / MALL between the dependent kernels
    float* buf; Q(hipMalloc(&buf, (n + 64) * sizeof(float))); Q(hipMemset(buf, 0, (n + 64) * sizeof(float)));
    run("no events", 0xffffffffu, false, buf, n);
    run("record, hipEventDisableTiming (torch.cuda.Event)", hipEventDisableTiming, false, buf, n);
    run("record, DisableTiming | hipEventDisableSystemFence", hipEventDisableTiming | hipEventDisableSystemFence, false, buf, n);
    run("record, DisableTiming | hipEventReleaseToDevice", hipEventDisableTiming | hipEventReleaseToDevice, false, buf, n);
    run("record + other stream waits and runs a kernel, DisableTiming", hipEventDisableTiming, true, buf, n);
    run("record + wait + kernel, DisableTiming | DisableSystemFence", hipEventDisableTiming | hipEventDisableSystemFence, true, buf, n);
    run("record + wait + kernel, DisableTiming | ReleaseToDevice", hipEventDisableTiming | hipEventReleaseToDevice, true, buf, n);
    run("no events", 0xffffffffu, false, buf, n);
    return 0;
}
