"""When does a stream that waits on an event recorded BETWEEN two long kernels of another stream get released: after the first
kernel (what the schedule assumes) or after the second?  Variants: event without / with timing."""
import torch
dev = torch.device('cuda:0')
a = torch.randn(8192, 8192, device=dev)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()


def long_kernel():
    return a @ a            # ~2-3 ms


for timing in (False, True):
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True)
        e_mid = torch.cuda.Event(enable_timing=True)
        e_end = torch.cuda.Event(enable_timing=True)
        e_b = torch.cuda.Event(enable_timing=True)
        rel = torch.cuda.Event(enable_timing=timing)
        with torch.cuda.stream(sA):
            t0.record()
            long_kernel()
            rel.record()
            e_mid.record()
            long_kernel()
            e_end.record()
        with torch.cuda.stream(sB):
            sB.wait_event(rel)
            x = torch.zeros(16, device=dev) + 1
            e_b.record()
        torch.cuda.synchronize()
    print('release event timing=%s: first kernel ends %.2f ms, second ends %.2f ms, waiting stream ran at %.2f ms' % (
        timing, t0.elapsed_time(e_mid), t0.elapsed_time(e_end), t0.elapsed_time(e_b)))

# second experiment: the work behind the event is a chain of 600 tiny kernels (CUs free) instead of a chip-filling GEMM
small = torch.zeros(1024, device=dev)
for _ in range(2):
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); e_mid = torch.cuda.Event(enable_timing=True)
    e_end = torch.cuda.Event(enable_timing=True); e_b = torch.cuda.Event(enable_timing=True)
    rel = torch.cuda.Event()
    with torch.cuda.stream(sA):
        t0.record()
        long_kernel()
        rel.record()
        e_mid.record()
        for _ in range(600):
            small.add_(1.0)
        e_end.record()
    with torch.cuda.stream(sB):
        sB.wait_event(rel)
        x = torch.zeros(16, device=dev) + 1
        e_b.record()
    torch.cuda.synchronize()
print('tiny-kernel chain behind the event: first kernel ends %.2f ms, chain ends %.2f ms, waiting stream ran at %.2f ms' % (
    t0.elapsed_time(e_mid), t0.elapsed_time(e_end), t0.elapsed_time(e_b)))
# third: the waiting stream's kernel is itself chip-filling
for _ in range(2):
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); e_mid = torch.cuda.Event(enable_timing=True)
    e_end = torch.cuda.Event(enable_timing=True); e_b0 = torch.cuda.Event(enable_timing=True); e_b = torch.cuda.Event(enable_timing=True)
    rel = torch.cuda.Event()
    with torch.cuda.stream(sA):
        t0.record()
        long_kernel()
        rel.record()
        e_mid.record()
        long_kernel()
        e_end.record()
    with torch.cuda.stream(sB):
        sB.wait_event(rel)
        e_b0.record()
        y = a @ a
        e_b.record()
    torch.cuda.synchronize()
print('GEMM behind the wait: first ends %.2f, second ends %.2f; waiting stream passed the wait at %.2f, its GEMM ends %.2f ms' % (
    t0.elapsed_time(e_mid), t0.elapsed_time(e_end), t0.elapsed_time(e_b0), t0.elapsed_time(e_b)))
