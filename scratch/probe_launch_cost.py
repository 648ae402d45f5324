"""Host cost of one launch through the Python wrapper / raw ctypes / a no-op C call."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcc_amd import ops
dev = torch.device('cuda:0')
L = ops.lib()
alpha = torch.rand(64, device=dev); mask = torch.zeros(64, device=dev)
x = ops.new_act(1, 64, 64, 64, dev); y = ops.new_act(1, 64, 64, 64, dev)
n = 20000


def t(fn, name):
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    te = time.time() - t0
    torch.cuda.synchronize()
    tt = time.time() - t0
    print('%-46s host %.2f us/call   (with GPU drain %.2f us)' % (name, te / n * 1e6, tt / n * 1e6), flush=True)


t(lambda: ops.gate_mask(alpha, 0.1, mask), 'ops.gate_mask (wrapper + ctypes + 1 launch)')
ap, mp, s = alpha.data_ptr(), mask.data_ptr(), ops.stream()
t(lambda: L.gcc_gate_mask(ap, 0.1, mp, 64, s), 'raw ctypes gcc_gate_mask (1 launch)')
t(lambda: L.gcc_strerror(0), 'raw ctypes gcc_strerror (no launch)')
t(lambda: ops.stream(), 'ops.stream()')
t(lambda: ops.geom(x), 'ops.geom(x)')
t(lambda: ops.bnact_fwd(x, y, act=ops.ACT_RELU), 'ops.bnact_fwd (1 launch, struct + 13 args)')
t(lambda: ops.nhwc_copy(x, 0, y, 0, 64), 'ops.nhwc_copy (1 launch)')
