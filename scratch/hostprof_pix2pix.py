"""cProfile of the host side of the Pix2Pix GCC step (bench.py's model): where the ~10 us per launch go."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
model, opt = bench.build(16)
train, val = bench.synthetic(16, 0, model.device)
for _ in range(5):
    bench.one_step(model, train, val)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    bench.one_step(model, train, val)
t_enq = (time.perf_counter() - t0) / 10
torch.cuda.synchronize()
print('host enqueue %.2f ms / step (wall incl. GPU drain %.2f)' % (t_enq * 1e3, (time.perf_counter() - t0) / 10 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    bench.one_step(model, train, val)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(32)
print(s.getvalue()[:6500])
