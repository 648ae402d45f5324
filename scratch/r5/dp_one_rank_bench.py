"""What the data-parallel plumbing costs a rank, measured where it can be: ONE rank of an RCCL (nccl) process group on a one-GPU box,
the model taking every world > 1 code path (bucketed all-reduces issued from the weight-gradient side streams, the asynchronous
teacher-generator bucket, the summed arch terms).  With one rank a collective moves no bytes over xGMI, but its stream, its launches
and the hardware-queue pressure are real: ProcessGroupNCCL's own stream is a FIFTH busy stream on the four hardware queues.
python3 scratch/r5/dp_one_rank_bench.py <route: none|torch|native> [steps]     (schedule switches come from the environment)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
route = sys.argv[1] if len(sys.argv) > 1 else 'none'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
if route != 'none':
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=os.environ.get('MASTER_PORT', '29577'), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
                      HSA_ENABLE_IPC_MODE_LEGACY='0', GCC_DP_FORCE_BUCKETS='1', GCC_DP_COMM=route)
import torch

import bench
from gcc_amd import ops

if route != 'none':
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group(backend='nccl', rank=0, world_size=1)
model, opt = bench.build(16)
if route != 'none':
    model._world = model.teacher_model._world = 2
train, val = bench.synthetic(16, 0, model.device)
for _ in range(8):
    bench.one_step(model, train, val)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    bench.one_step(model, train, val)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
red = getattr(model.optimizer_D, 'reducer', None)
print('route %-6s buckets %-5s %s: %.2f ms/step, %.1f images/s' % (route, red is not None, os.environ.get('GCC_R5_TAG', ''), dt * 1e3, 16 / dt), flush=True)
if route != 'none':
    dist.barrier()
    dist.destroy_process_group()
