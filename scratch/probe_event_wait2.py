"""Stream B waits on an event recorded in the middle of stream A's chain of one-wave chip-filling igemm launches (PatchGAN L4
forward: 244 workgroups, ~205 us each), then runs a chain of small kernels.  When does B start and how fast does it progress?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcc_amd import ops
dev = torch.device('cuda:0')
ops.lib()
x = ops.new_act(16, 512, 32, 32, dev); x.normal_()
m = (torch.randn(1024, 512, 4, 4, device=dev) * 0.02).contiguous(memory_format=torch.channels_last)
wp, _ = ops.pack_weights(m)
y = ops.new_act(16, 1024, 31, 31, dev)
small = torch.zeros(4096, device=dev)
import os
PRI = int(os.environ.get('PRI', '0'))
sA, sB = torch.cuda.Stream(), torch.cuda.Stream(priority=PRI)
print('priority of B:', PRI, 'range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else '?')


def big():
    ops.conv_fprop(x, wp, 1024, 4, 1, 1, out=y)


for nsmall, what in ((100, 'fill kernels'),):
    for rep in range(3):
        torch.cuda.synchronize()
        ev = lambda: torch.cuda.Event(enable_timing=True)
        t0, e_mid, e_end, b0, b1 = ev(), ev(), ev(), ev(), ev()
        rel = torch.cuda.Event()
        with torch.cuda.stream(sA):
            t0.record()
            for _ in range(3):
                big()
            rel.record()
            e_mid.record()
            for _ in range(10):
                big()
            e_end.record()
        with torch.cuda.stream(sB):
            sB.wait_event(rel)
            b0.record()
            for _ in range(nsmall):
                ops.fill(small, 1.0)
            b1.record()
        torch.cuda.synchronize()
    print('A: 3 igemm end %.2f ms, 10 more end %.2f ms | B (%d %s): passed the wait at %.2f ms, chain done at %.2f ms' % (
        t0.elapsed_time(e_mid), t0.elapsed_time(e_end), nsmall, what, t0.elapsed_time(b0), t0.elapsed_time(b1)))
# the same with B alone
torch.cuda.synchronize()
b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(sB):
    b0.record()
    for _ in range(100):
        ops.fill(small, 1.0)
    b1.record()
torch.cuda.synchronize()
print('B alone: 100 fill kernels %.2f ms' % b0.elapsed_time(b1))
