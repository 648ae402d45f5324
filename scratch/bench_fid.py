"""FID arithmetic at the real size (2048 Inception features): statistics of 500 fakes, Frechet distance against
full-rank real statistics; GPU time and agreement with scipy (CPU oracle, timed beside it)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcc_amd.metric import fid_score as F
from oracle import metric_oracle as M
d = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rng = np.random.RandomState(0)
basis = rng.randn(d, d) / np.sqrt(d)
real = ((rng.randn(3000, d) * (0.1 + rng.rand(d))) @ basis).astype(np.float32)
fake = ((rng.randn(500, d) * (0.1 + rng.rand(d))) @ basis.T).astype(np.float32) + 0.05
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    m1, s1 = F.activation_statistics(real); m2, s2 = F.activation_statistics(fake)
    torch.cuda.synchronize(); t1 = time.time()
    fid, resid = F.calculate_frechet_distance(m1, s1, m2, s2, return_residual=True)
    t2 = time.time()
print('GPU: statistics %.1f ms, frechet %.1f ms (2 x 60 Newton-Schulz steps), fid %.10g, last-step trace change %.1e' % (
    (t1 - t0) * 1e3, (t2 - t1) * 1e3, fid, resid))
t0 = time.time()
a, b = M.activation_statistics(real), M.activation_statistics(fake)
t1 = time.time()
ref = M.calculate_frechet_distance(a[0], a[1], b[0], b[1])
t2 = time.time()
print('CPU oracle (numpy / scipy sqrtm): statistics %.1f ms, frechet %.1f ms, fid %.10g, rel diff %.2e' % (
    (t1 - t0) * 1e3, (t2 - t1) * 1e3, ref, abs(ref - fid) / abs(ref)))
flops = 2 * 60 * 3 * 2.0 * d ** 3
