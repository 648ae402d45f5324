"""does Work.wait() of a sub-group's collective block the HOST?  (one-rank nccl group; a long kernel chain in front of the collective)"""
import os, sys, time
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29591', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group(backend='nccl', rank=0, world_size=1)
sub = dist.new_group(ranks=[0])
dev = torch.device('cuda:0')
junk = torch.empty(256 << 20, dtype=torch.float32, device=dev)
buf = torch.zeros(1 << 20, dtype=torch.float32, device=dev)
side = torch.cuda.Stream()
for name, group in (('default', None), ('sub-group', sub), ('default', None), ('sub-group', sub)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        for i in range(40):
            junk.fill_(float(i))                  # ~10 ms of device work in front of the collective
        t1 = time.perf_counter()
        h = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=True)
        t2 = time.perf_counter()
        h.wait()
        t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print('%-10s host: enqueue fills %.2f ms, all_reduce call %.2f ms, wait() %.2f ms; device done after %.2f ms' % (
        name, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t0) * 1e3), flush=True)
dist.barrier()
dist.destroy_process_group()
