"""write / copy bandwidth floor for the thin-layer outputs (67 MB bf16)"""
import torch
def t(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for mb in (17, 34, 67, 134, 268):
    n = mb * 1000 * 1000 // 2
    y = torch.empty(n, dtype=torch.bfloat16, device='cuda'); x = torch.randn(n, device='cuda').bfloat16()
    tf = t(lambda: y.fill_(1.0)); tc = t(lambda: y.copy_(x))
    print('%4d MB: fill %6.1f us (%.2f TB/s)   copy %6.1f us (%.2f TB/s r+w)' % (mb, tf, mb / tf * 1e-6 * 1e6 / 1e6 * 1e0, tc, 2 * mb / tc))
