"""FID arithmetic on the GPU (metric/fid_score.py:150-214, 219-284, 327-328): activation statistics in f64 and the Frechet
distance with a Newton-Schulz matrix square root.  No CPU path: a missing library / GPU raises GccError."""
import numpy as np
import torch

from .. import ops
from .._lib import GccError, check

NEWTON_SCHULZ_ITERATIONS = 60       # each step is three d^3 f64 GEMMs; small eigenvalues converge linearly (x1.5 per step)
SHIFT_REL = 1e-13                   # delta / |s1 s2|_F of the two shifted solves (see csrc/metric.hip)


def _dev():
    if not torch.cuda.is_available():
        raise GccError('gcc_amd runs on MI355X only (no CPU path): need a visible GPU')
    return torch.device('cuda', torch.cuda.current_device())


def _f64(a, dev):
    t = a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
    return t.to(device=dev, dtype=torch.float64).contiguous()


def activation_statistics(act):
    """mu = np.mean(act, axis=0), sigma = np.cov(act, rowvar=False) (:327-328) of [n, d] activations (numpy or tensor,
    fp32 or f64) -> (mu [d], sigma [d, d]) f64 device tensors"""
    dev = _dev()
    t = act if torch.is_tensor(act) else torch.from_numpy(np.ascontiguousarray(act))
    if t.dtype not in (torch.float32, torch.float64):
        t = t.double()
    t = t.to(dev).contiguous()
    n, d = t.shape
    mu = torch.empty(d, dtype=torch.float64, device=dev)
    sigma = torch.empty((d, d), dtype=torch.float64, device=dev)
    L = ops.lib()
    ws = torch.empty(L.gcc_activation_stats_workspace(n, d), dtype=torch.uint8, device=dev)
    check(L.gcc_activation_stats(t.data_ptr(), int(t.dtype == torch.float64), n, d, mu.data_ptr(), sigma.data_ptr(),
                                 ws.data_ptr(), ws.numel(), ops.stream()), 'gcc_activation_stats')
    return mu, sigma


def calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6, iterations=None, return_residual=False):
    """metric/fid_score.py:219-284: d^2 = |mu1 - mu2|^2 + Tr(C1 + C2 - 2 sqrt(C1 C2)).  The reference retries with
    eps * I added to both covariances when scipy's sqrtm returns non-finite entries; here that happens when the
    iteration has not produced a finite trace."""
    dev = _dev()
    mu1, mu2 = _f64(np.atleast_1d(mu1) if not torch.is_tensor(mu1) else mu1, dev), _f64(
        np.atleast_1d(mu2) if not torch.is_tensor(mu2) else mu2, dev)
    sigma1, sigma2 = _f64(np.atleast_2d(sigma1) if not torch.is_tensor(sigma1) else sigma1, dev), _f64(
        np.atleast_2d(sigma2) if not torch.is_tensor(sigma2) else sigma2, dev)
    assert mu1.shape == mu2.shape, 'Training and test mean vectors have different lengths'
    assert sigma1.shape == sigma2.shape, 'Training and test covariances have different dimensions'
    d = mu1.numel()
    L = ops.lib()
    ws = torch.empty(L.gcc_frechet_workspace(d), dtype=torch.uint8, device=dev)
    out = torch.zeros(2, dtype=torch.float64, device=dev)

    def run(s1, s2):
        check(L.gcc_frechet_distance(mu1.data_ptr(), s1.data_ptr(), mu2.data_ptr(), s2.data_ptr(), d,
                                     int(iterations or NEWTON_SCHULZ_ITERATIONS), SHIFT_REL, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                     ops.stream()), 'gcc_frechet_distance')
        return out.cpu().numpy()
    val = run(sigma1, sigma2)
    if not np.isfinite(val).all():
        print('fid calculation produces singular product; adding %s to diagonal of cov estimates' % eps)
        off = torch.eye(d, dtype=torch.float64, device=dev) * eps
        val = run(sigma1 + off, sigma2 + off)
    return (float(val[0]), float(val[1])) if return_residual else float(val[0])


def get_activations_from_ims(ims, model, batch_size=50, dims=2048, device=None, verbose=False, use_tqdm=True):
    """metric/fid_score.py:150-214: ``model(batch)[0]`` on [0, 1] NCHW batches, global average pooling if the map is not
    1x1; the activations stay on the device (f32)."""
    if hasattr(model, 'eval'):
        model.eval()
    n = len(ims)
    out = []
    for start in range(0, n, batch_size):
        images = np.array(ims[start:start + batch_size], dtype=np.float64)
        if images.shape[1] != 3:
            images = images.transpose((0, 3, 1, 2))
        images = images / 255
        batch = torch.from_numpy(images).type(torch.FloatTensor).to(device)
        with torch.no_grad():
            pred = model(batch)[0]
        if pred.shape[2] != 1 or pred.shape[3] != 1:
            pred = pred.mean((2, 3), keepdim=True)
        out.append(pred.reshape(pred.shape[0], -1).float())
    return torch.cat(out, 0)


def _compute_statistics_of_ims(ims, model, batch_size, dims, device, use_tqdm=True):
    act = get_activations_from_ims(ims, model, batch_size, dims, device, verbose=False, use_tqdm=use_tqdm)
    return activation_statistics(act)
