"""CPU restatement of the reference's paired-image input pipeline (TEST INFRASTRUCTURE: only tests/ may import this).

data/aligned_dataset.py:40-53 + data/base_dataset.py:81-112 with torchvision's transforms spelled out (torchvision is
absent from the image): Resize -> PIL ``img.resize((w, h), BICUBIC)``, __crop -> ``img.crop``, __flip ->
``transpose(FLIP_LEFT_RIGHT)``, ToTensor -> uint8 / 255 as CHW float32, Normalize -> (x - 0.5) / 0.5.
``resample_bicubic`` restates Pillow's Resample.c (two integer passes, 22-bit coefficients); it is pinned to PIL 12.2.0
itself in tests/test_pipeline.py and through tests/golden/pipeline.npz."""

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _bilinear(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def _coeffs(in_size, out_size, filt='bicubic'):
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = (2.0 if filt == 'bicubic' else 1.0) * filterscale
    _bicubic = globals()['_bicubic'] if filt == 'bicubic' else _bilinear
    out = []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = [_bicubic((x + xmin - center + 0.5) / filterscale) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        kk = [int(-0.5 + (w / ww) * (1 << PRECISION_BITS)) if w / ww < 0 else int(0.5 + (w / ww) * (1 << PRECISION_BITS)) for w in k]
        out.append((xmin, np.array(kk, dtype=np.int64)))
    return out


def _pass(img, out_size, axis, filt='bicubic'):
    img = np.moveaxis(img, axis, 0).astype(np.int64)
    res = np.empty((out_size,) + img.shape[1:], dtype=np.uint8)
    for xx, (xmin, kk) in enumerate(_coeffs(img.shape[0], out_size, filt)):
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk, img[xmin:xmin + len(kk)], axes=(0, 0))
        res[xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(res, 0, axis)


def resample_bicubic(img, out_h, out_w, filt='bicubic'):
    """uint8 [h, w, 3] -> uint8 [out_h, out_w, 3], horizontal pass first (Pillow's ImagingResample)"""
    if img.shape[1] != out_w:
        img = _pass(img, out_w, 1, filt)
    if img.shape[0] != out_h:
        img = _pass(img, out_h, 0, filt)
    return img.copy()


def sr_item(img, crop, scale, left, top):
    """ImageTransforms (data/sr_dataset.py:86-121), train split, with the crop position given: (lr imagenet-norm, hr [-1, 1])"""
    hr = img[top:top + crop, left:left + crop]
    lr = resample_bicubic(hr, crop // scale, crop // scale)
    t = lambda a: np.transpose(a.astype(np.float32) / np.float32(255.), (2, 0, 1))
    mean = np.array([0.485, 0.456, 0.406], dtype=np.float32).reshape(3, 1, 1)
    std = np.array([0.229, 0.224, 0.225], dtype=np.float32).reshape(3, 1, 1)
    return (t(lr) - mean) / std, np.float32(2.) * t(hr) - np.float32(1.)


def sa_item(img, size, center_crop):
    """SADataset.get_transform (data/sa_dataset.py:26-38): CenterCrop(160), bilinear Resize, ToTensor, Normalize(.5, .5)"""
    if center_crop:
        h, w = img.shape[:2]
        top, left = int(round((h - 160) / 2.0)), int(round((w - 160) / 2.0))
        img = img[top:top + 160, left:left + 160]
    r = resample_bicubic(img, size, size, 'bilinear')
    t = np.transpose(r.astype(np.float32) / np.float32(255.), (2, 0, 1))
    return (t - np.float32(0.5)) / np.float32(0.5)


def aligned_item(AB, load_size, crop_size, crop_pos, flip, resize=True):
    """AlignedDataset.__getitem__ on a decoded uint8 [h, 2w, 3] image -> (A, B) float32 [3, crop, crop]"""
    w2 = int(AB.shape[1] / 2)
    out = []
    for img in (AB[:, :w2], AB[:, w2:]):
        if resize:
            img = resample_bicubic(img, load_size, load_size)
        x, y = crop_pos
        if img.shape[1] > crop_size or img.shape[0] > crop_size:
            img = img[y:y + crop_size, x:x + crop_size]
        if flip:
            img = img[:, ::-1]
        t = np.transpose(img.astype(np.float32) / np.float32(255.), (2, 0, 1))
        out.append((t - np.float32(0.5)) / np.float32(0.5))
    return out
