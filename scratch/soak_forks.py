"""Full-size soak of the stream forks: the bench's Pix2Pix iteration (N = 16, 256 x 256, dropout on) for `steps` iterations with
the forks of backward_G's tail / the architecture step off and on -- every parameter, BatchNorm buffer and Adam moment must end on
the same bits (the forks only change which stream a chain runs on; a race would show as a difference).
    python scratch/soak_forks.py [steps]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gcc_amd.models import Pix2Pix as P

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150


def digest(model):
    h = hashlib.sha256()
    for tag, m in (('s', model), ('t', model.teacher_model)):
        for k, v in sorted(m.state_dict().items()):
            h.update(k.encode()); h.update(v.detach().float().cpu().numpy().tobytes())
        for name in sorted(dir(m)):
            o = getattr(m, name, None)
            for pn in ('plan', 'plan_dup'):
                p = getattr(o, pn, None) if name.startswith('optimizer') else None
                if p is not None:
                    for a, b in zip(p.m, p.v):
                        h.update(a.detach().cpu().numpy().tobytes()); h.update(b.detach().cpu().numpy().tobytes())
    return h.hexdigest()


out = {}
for forks in ((False, False, False), (True, True, False), (True, True, True)):
    P.DISTILL_FORK, P.ARCH_FORK, P.ARCH_EARLY = forks
    model, opt = bench.build(16)
    train, val = bench.synthetic(16, 0, model.device)
    for i in range(steps):
        bench.one_step(model, train if i % 2 == 0 else val, val if i % 2 == 0 else train)
    torch.cuda.synchronize()
    losses = {k: round(v, 6) for k, v in model.get_current_losses().items()}
    out[forks] = (digest(model), losses)
    print('forks (distill, arch, arch-early) = %s: %s  %s' % (forks, out[forks][0][:16], losses), flush=True)
    del model
    torch.cuda.empty_cache()
ds = {v[0] for v in out.values()}
print('IDENTICAL' if len(ds) == 1 else 'DIFFERENT: %d distinct states' % len(ds))
sys.exit(0 if len(ds) == 1 else 1)
