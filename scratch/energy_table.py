"""Per-kernel-family energy: socket power sampled (hwmon power1_average / rocm-smi fallback) while ONE family loops alone.
Usage: python scratch/energy_table.py [seconds per family]   ->  W (mean over >= 100 samples), us / launch, J / launch, pJ / FLOP"""
import glob
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gcc_amd import ops

DEV = 'cuda:0'
SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0


def power_files():
    return sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/power1_average') +
                  glob.glob('/sys/class/hwmon/hwmon*/power1_average'))


def read_power():
    best = 0.0
    for f in power_files():
        try:
            best = max(best, int(open(f).read().strip()) / 1e6)
        except Exception:
            pass
    if best > 0:
        return best
    try:
        out = subprocess.run(['rocm-smi', '--showpower'], capture_output=True, text=True, timeout=5).stdout
        for line in out.splitlines():
            if 'Power' in line and 'W' in line:
                return float(line.split(':')[-1].strip().split()[0])
    except Exception:
        pass
    return float('nan')


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.samples, self.stop_flag = [], False

    def run(self):
        while not self.stop_flag:
            self.samples.append(read_power())
            time.sleep(0.01 if power_files() else 0.05)


def measure(name, fn, flop):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    time.sleep(0.5)
    s = Sampler()
    n = 0
    t0 = time.perf_counter()
    s.start()
    while time.perf_counter() - t0 < SECS:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s.stop_flag = True
    s.join()
    smp = [v for v in s.samples[len(s.samples) // 5:] if v == v]      # drop the ramp
    w = sum(smp) / max(len(smp), 1)
    us = dt / n * 1e6
    print('%-34s %5d samples %7.1f W  %8.1f us/launch %8.4f J/launch %s' % (
        name, len(smp), w, us, w * us * 1e-6, ('%6.2f pJ/FLOP' % (w * us * 1e-6 / flop * 1e12)) if flop else ''), flush=True)


def conv_ops(N, H, W, Ci, Co, k, s, p):
    g = torch.Generator().manual_seed(0)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = ops.new_act(N, Ci, H, W, DEV)
    x.copy_(torch.randn(N, Ci, H, W, generator=g).bfloat16().to(DEV))
    dy = ops.new_act(N, Co, Ho, Wo, DEV)
    dy.copy_(torch.randn(N, Co, Ho, Wo, generator=g).bfloat16().to(DEV))
    m = (torch.randn(Co, Ci, k, k, generator=g) * 0.05).to(DEV).contiguous(memory_format=torch.channels_last)
    w, wt = ops.pack_weights(m)
    y = ops.new_act(N, Co, Ho, Wo, DEV)
    dx = ops.new_act(N, Ci, H, W, DEV)
    dw = torch.zeros_like(m)
    fl = 2.0 * N * Ho * Wo * Co * k * k * Ci
    return (lambda: ops.conv_fprop(x, w, Co, k, s, p, out=y), lambda: ops.conv_dgrad(dy, wt, Ci, H, W, k, s, p, out=dx),
            lambda: ops.conv_wgrad(x, dy, dw, k, s, p, accumulate=True), fl)


print('power source: %s' % (power_files() or 'rocm-smi'))
time.sleep(1.0)
idle = [read_power() for _ in range(50) if not time.sleep(0.01)]
print('%-34s %5d samples %7.1f W' % ('idle', len(idle), sum(idle) / len(idle)))
f4, d4, w4, fl4 = conv_ops(16, 32, 32, 512, 1024, 4, 1, 1)
f2, d2, w2, fl2 = conv_ops(16, 128, 128, 128, 256, 4, 2, 1)
measure('igemm L4 forward (halo, 256 WGs)', f4, fl4)
measure('igemm L4 data gradient (128 WGs)', d4, fl4)
measure('igemm L2 forward (halo)', f2, fl2)
measure('igemm L2 data gradient (halo)', d2, fl2)
measure('wgrad L4', w4, fl4)
measure('wgrad L2', w2, fl2)
big = ops.new_act(16, 256, 64, 64, DEV)
out = ops.new_act(16, 256, 64, 64, DEV)
measure('streaming copy 33 MB (bnact-like)', lambda: ops.nhwc_copy(big, 0, out, 0, 256), 0)
import bench
model, opt = bench.build(16)
train, val = bench.synthetic(16, 0, model.device)
model.set_stream_schedule(False, 'production')


def g_fwd():
    model.set_input(train)
    model.forward()


measure('student U-Net forward (small chain)', g_fwd, 16 * 3.10e9)
model.set_stream_schedule(True)
measure('whole step, production schedule', lambda: bench.one_step(model, train, val), 16 * 679.3e9)
