"""Evaluation arithmetic: the CPU oracle against the reference's own functions
(tests/golden/metric.npz), and the HIP entry points against the oracle and the golden values."""
import os

import numpy as np
import pytest
import torch


def _z(golden_dir):
    return np.load(os.path.join(golden_dir, 'metric.npz'))


def test_metric_oracle_vs_reference_golden(golden_dir):
    from oracle import metric_oracle as M
    z = _z(golden_dir)
    for tag in ('full', 'wide', 'rank'):
        m1, s1 = M.activation_statistics(z[tag + '.act1'])
        m2, s2 = M.activation_statistics(z[tag + '.act2'])
        assert np.array_equal(m1, z[tag + '.mu1']) and np.array_equal(s1, z[tag + '.sigma1'])
        assert abs(M.calculate_frechet_distance(m1, s1, m2, s2) - float(z[tag + '.fid'])) <= 1e-9 * float(z[tag + '.fid'])
    hist = M.fast_hist(z['iou.pred'], z['iou.label'], 19)
    assert np.array_equal(hist, z['iou.hist'])
    assert np.array_equal(z['iou.scores'].argmax(axis=1).reshape(-1), z['iou.pred'])
    iu = M.per_class_iu(hist.astype(np.float64))
    assert np.allclose(iu, z['iou.per_class'], rtol=0, atol=0, equal_nan=True)
    assert round(float(np.nanmean(iu * 100)), 2) == float(z['iou.miou'])
    assert np.allclose(M.y_channel(z['psnr.fake']), z['psnr.fake_y'], rtol=1e-6, atol=1e-4)
    assert np.allclose(M.y_channel(z['psnr.real']), z['psnr.real_y'], rtol=1e-6, atol=1e-4)
    ref_mse = np.mean((z['psnr.fake_y'].astype(np.float64) - z['psnr.real_y'].astype(np.float64)) ** 2)
    assert abs(M.psnr_y(z['psnr.fake'], z['psnr.real']) - 10 * np.log10(255. ** 2 / ref_mse)) < 1e-4


@pytest.mark.gpu
def test_miou_kernels_exact(golden_dir):
    from gcc_amd.metric import mIoU_score as G
    z = _z(golden_dir)
    dev = torch.device('cuda:0')
    pred = G.argmax_classes(torch.from_numpy(z['iou.scores']).to(dev))
    assert np.array_equal(pred.cpu().numpy().reshape(-1).astype(np.int64), z['iou.pred'])
    s = torch.from_numpy(z['iou.scores'][:1, :, :4, :4].copy())
    s[0, 5, 1, 1] = float('nan')
    s[0, 7, 1, 1] = float('nan')
    assert np.array_equal(G.argmax_classes(s.to(dev)).cpu().numpy(), s.numpy().argmax(axis=1))       # first NaN wins
    hist = G.fast_hist(z['iou.pred'], z['iou.label'], 19)
    assert hist.dtype == torch.int64 and np.array_equal(hist.cpu().numpy(), z['iou.hist'])
    hist = G.fast_hist(z['iou.pred'], z['iou.label'], 19, hist)                                        # accumulates
    assert np.array_equal(hist.cpu().numpy(), 2 * z['iou.hist'])
    iu = G.per_class_iu(torch.from_numpy(z['iou.hist']))
    assert np.allclose(iu, z['iou.per_class'], rtol=0, atol=0, equal_nan=True)
    big = torch.randint(0, 19, (3_000_000,), dtype=torch.int32, device=dev)
    lab = torch.randint(-2, 25, (3_000_000,), dtype=torch.int32, device=dev)
    from oracle import metric_oracle as M
    assert np.array_equal(G.fast_hist(big, lab, 19).cpu().numpy(), M.fast_hist(big.cpu().numpy(), lab.cpu().numpy(), 19))

    class Seg(torch.nn.Module):                     # stands in for the external DRN: returns (scores, features)
        def forward(self, x):
            return x, None
    sc = torch.from_numpy(z['iou.scores'])
    lab2 = torch.from_numpy(z['iou.label'][:4000]).reshape(2, 40, 50)
    val = G.test(None, None, Seg(), dev, num_classes=19, dataset=[(sc, lab2)])
    assert val == float(z['iou.miou'])


@pytest.mark.gpu
def test_fid_kernels_vs_reference_golden(golden_dir):
    from gcc_amd.metric import fid_score as F
    z = _z(golden_dir)
    for tag, tol in (('full', 1e-7), ('wide', 1e-7), ('rank', 1e-7)):
        for dtype in (np.float32, np.float64):
            mu, sigma = F.activation_statistics(z[tag + '.act1'].astype(dtype))
            assert np.allclose(mu.cpu().numpy(), z[tag + '.mu1'], rtol=1e-12, atol=1e-13)
            assert np.allclose(sigma.cpu().numpy(), z[tag + '.sigma1'], rtol=1e-11, atol=1e-13)
        m2, s2 = F.activation_statistics(torch.from_numpy(z[tag + '.act2']).cuda())
        fid, resid = F.calculate_frechet_distance(mu, sigma, m2, s2, return_residual=True)
        ref = float(z[tag + '.fid'])
        print('%s: fid %.12g reference %.12g (rel %.2e), last Newton-Schulz step moved the trace by %.1e' % (
            tag, fid, ref, abs(fid - ref) / ref, resid))
        # 'rank': fewer samples than dimensions, singular covariance product (scipy's own result is 4e-9 from the
        # eigenvalue-based value there)
        assert abs(fid - ref) <= tol * ref, (tag, fid, ref)
    assert abs(F.calculate_frechet_distance(z['full.mu1'], z['full.sigma1'], z['full.mu1'], z['full.sigma1'])) < 1e-6   # 2 tr(sigma) = 55


@pytest.mark.gpu
def test_psnr_y_kernel(golden_dir):
    from gcc_amd import ops
    from gcc_amd._lib import check
    from oracle import metric_oracle as M
    z = _z(golden_dir)
    dev = torch.device('cuda:0')
    fake, real = torch.from_numpy(z['psnr.fake']).to(dev), torch.from_numpy(z['psnr.real']).to(dev)
    L = ops.lib()
    sse = torch.zeros(1, dtype=torch.float64, device=dev)
    ws = torch.empty(L.gcc_psnr_workspace(), dtype=torch.uint8, device=dev)
    N, _, H, W = fake.shape
    for acc in (0, 1):
        check(L.gcc_psnr_y_sse(fake.data_ptr(), real.data_ptr(), N, H, W, sse.data_ptr(), acc, ws.data_ptr(), ws.numel(),
                               ops.stream()), 'gcc_psnr_y_sse')
    ref_sse = np.sum((z['psnr.fake_y'].astype(np.float64) - z['psnr.real_y'].astype(np.float64)) ** 2)
    assert abs(sse.item() - 2 * ref_sse) <= 1e-5 * ref_sse
    psnr = 10 * np.log10(255. ** 2 / (sse.item() / 2 / (N * (H - 8) * (W - 8))))
    assert abs(psnr - M.psnr_y(z['psnr.fake'], z['psnr.real'])) < 1e-4
    # SSIM on the same luminance images (skimage's definition restated in the oracle: unpinned, like PSNR)
    acc = torch.zeros(1, dtype=torch.float64, device=dev)
    check(L.gcc_ssim_y_sum(fake.data_ptr(), real.data_ptr(), N, H, W, acc.data_ptr(), 0, ws.data_ptr(), ws.numel(), ops.stream()),
          'gcc_ssim_y_sum')
    ssim = acc.item() / (N * (H - 14) * (W - 14))
    ref = M.ssim_y(z['psnr.fake'], z['psnr.real'])
    assert 0.0 < ref < 1.0 and abs(ssim - ref) < 1e-9, (ssim, ref)
    same = torch.zeros(1, dtype=torch.float64, device=dev)
    check(L.gcc_ssim_y_sum(real.data_ptr(), real.data_ptr(), N, H, W, same.data_ptr(), 0, ws.data_ptr(), ws.numel(), ops.stream()),
          'gcc_ssim_y_sum')
    assert abs(same.item() / (N * (H - 14) * (W - 14)) - 1.0) < 1e-12
