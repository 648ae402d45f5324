"""CPU restatement of the reference's evaluation arithmetic (TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this; the product path never does).

Pinned against the reference's own functions run in the build container (tests/golden/make_fixtures.py::fixture_metric
-> tests/golden/metric.npz): calculate_frechet_distance (scipy.linalg.sqrtm), fast_hist, per_class_iu, convert_image.
skimage is absent from the image: peak_signal_noise_ratio is restated from its published definition,
10 log10(data_range^2 / mean((a - b)^2)) in float64, and structural_similarity from the definition its documentation gives
(Wang et al. 2004 with a uniform 7 x 7 window, sample covariance, K1 = 0.01, K2 = 0.03, the mean taken over the SSIM map
with a border of (7 - 1) / 2 pixels cropped): PARITY UNPINNED for these two formulas (no skimage here to check against)."""
import numpy as np
from scipy import linalg


def calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """metric/fid_score.py:219-284"""
    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    diff = mu1 - mu2
    covmean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(covmean).all():
        offset = np.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if np.iscomplexobj(covmean):
        covmean = covmean.real
    return diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean)


def activation_statistics(act):
    """metric/fid_score.py:327-328"""
    act = np.asarray(act, dtype=np.float64)
    return np.mean(act, axis=0), np.cov(act, rowvar=False)


def fast_hist(pred, label, n):
    """metric/mIoU_score.py:163-167"""
    k = (label >= 0) & (label < n)
    return np.bincount(n * label[k].astype(int) + pred[k], minlength=n ** 2).reshape(n, n)


def per_class_iu(hist):
    """metric/mIoU_score.py:108-109"""
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.diag(hist) / (hist.sum(1) + hist.sum(0) - np.diag(hist))


def y_channel(img):
    """data/sr_dataset.py:36-37, 58-62 on an NCHW float32 array in [-1, 1] -> [N, H-8, W-8] float32"""
    x = (img.astype(np.float32) + np.float32(1.)) / np.float32(2.)
    x = np.float32(255.) * np.transpose(x, (0, 2, 3, 1))[:, 4:-4, 4:-4, :]
    w = np.array([65.481, 128.553, 24.966], dtype=np.float32)
    return (x @ w) / np.float32(255.) + np.float32(16.)


def ssim_y(fake, real, data_range=255.):
    """models/SRGAN.py:659-661 with skimage.metrics.structural_similarity(real_y, fake_y, data_range=255.) restated
    (defaults: win_size 7, uniform filter, use_sample_covariance, K1 .01, K2 .03); mean over the batch"""
    from scipy.ndimage import uniform_filter
    X, Y = y_channel(real).astype(np.float64), y_channel(fake).astype(np.float64)
    out = []
    for x, y in zip(X, Y):
        ux, uy = uniform_filter(x, 7), uniform_filter(y, 7)
        uxx, uyy, uxy = uniform_filter(x * x, 7), uniform_filter(y * y, 7), uniform_filter(x * y, 7)
        cn = 49. / 48.
        vx, vy, vxy = cn * (uxx - ux * ux), cn * (uyy - uy * uy), cn * (uxy - ux * uy)
        C1, C2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
        S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
        out.append(S[3:-3, 3:-3].mean())
    return float(np.mean(out))


def psnr_y(fake, real):
    """models/SRGAN.py:653-657 with skimage.metrics.peak_signal_noise_ratio(data_range=255.) restated"""
    a, b = y_channel(fake).astype(np.float64), y_channel(real).astype(np.float64)
    mse = np.mean((a - b) ** 2)
    return 10 * np.log10(255. ** 2 / mse)
