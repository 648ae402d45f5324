"""Input pipeline of the aligned (paired A|B) datasets on MI355X (data/aligned_dataset.py:27-56,
data/base_dataset.py:63-112): a decoded 8-bit RGB image is split, each half resized with PIL's BICUBIC, cropped,
flipped, scaled and normalised on the GPU -- the per-image work of the reference's DataLoader workers, which 8 workers
cannot deliver at the rate the HIP step consumes images.  Decoding the files stays on the host (out of scope).

The augmentation parameters are drawn exactly as the reference draws them (``random.randint`` x 2, ``random.random``),
so a seeded run sees the same crops and flips."""
import math
import random

import numpy as np
import torch

from .. import ops
from .._lib import GccError, check

PRECISION_BITS = 32 - 8 - 2


def get_params(opt, size):
    """data/base_dataset.py:63-78"""
    w, h = size
    new_h, new_w = h, w
    if opt.preprocess == 'resize_and_crop':
        new_h = new_w = opt.load_size
    elif opt.preprocess == 'scale_width_and_crop':
        new_w = opt.load_size
        new_h = opt.load_size * h // w
    x = random.randint(0, int(np.maximum(0, new_w - opt.crop_size)))
    y = random.randint(0, int(np.maximum(0, new_h - opt.crop_size)))
    flip = random.random() > 0.5
    return {'crop_pos': (x, y), 'flip': flip}


PREPROCESS_MODES = ('resize_and_crop', 'crop', 'scale_width', 'scale_width_and_crop', 'resize', 'none', 'none_exact')


def resize_target(opt, h, w):
    """The (height, width) the resizing stage of get_transform (data/base_dataset.py:81-131) gives an h x w image, or None when it
    leaves the image alone: 'resize*' -> load_size x load_size (:85-87); 'scale_width*' -> width load_size, height
    int(max(load_size * h / w, crop_size)) unless the width already fits and the height covers the crop (__scale_width, :123-129);
    'none' -> both sides rounded to a multiple of 4 with Python's round (__make_power_2, :112-120); 'crop' -> nothing."""
    pre = getattr(opt, 'preprocess', 'resize_and_crop')
    if 'resize' in pre:
        return (opt.load_size, opt.load_size) if (h, w) != (opt.load_size, opt.load_size) else None
    if 'scale_width' in pre:
        if w == opt.load_size and h >= opt.crop_size:
            return None
        return int(max(opt.load_size * h / w, opt.crop_size)), opt.load_size
    if pre == 'none':
        nh, nw = int(round(h / 4) * 4), int(round(w / 4) * 4)
        return None if (nh, nw) == (h, w) else (nh, nw)
    return None


def _bicubic(x):
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _bilinear(x):
    if x < 0.0:
        x = -x
    if x < 1.0:
        return 1.0 - x
    return 0.0


_FILTERS = {'bicubic': (_bicubic, 2.0), 'bilinear': (_bilinear, 1.0)}
_coeff_cache = {}


def resample_coeffs(in_size, out_size, filter='bicubic'):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc (Resample.c) for the BICUBIC (or BILINEAR) filter over the whole
    axis: (bounds int32 [out, 2], coefficients int32 [out, ksize], ksize).  Double arithmetic in Pillow's operation order."""
    key = (in_size, out_size, filter)
    if key in _coeff_cache:
        return _coeff_cache[key]
    _bicubic, base_support = _FILTERS[filter]
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = base_support * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coef = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        for x in range(xmax):
            v = k[x] / ww if ww != 0.0 else k[x]
            coef[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    _coeff_cache[key] = (bounds, coef, ksize)
    return _coeff_cache[key]


class AlignedGpuPipeline:
    """``pipe(AB)`` -> {'A', 'B'}: what AlignedDataset.__getitem__ returns for one decoded image, as device tensors
    (NCHW fp32 in [-1, 1], the batch-dict contract of set_input)."""

    def __init__(self, opt, device=None):
        if not torch.cuda.is_available():
            raise GccError('gcc_amd runs on MI355X only (no CPU path): need a visible GPU')
        self.opt = opt
        self.device = device or torch.device('cuda', torch.cuda.current_device())
        self._tables = {}
        if getattr(opt, 'preprocess', 'resize_and_crop') not in PREPROCESS_MODES:
            raise ValueError('unknown --preprocess %s (data/base_dataset.py:81-112 knows %s)' % (opt.preprocess, ', '.join(PREPROCESS_MODES[:6])))

    def pre_resize(self, img):
        """the resizing stage of get_transform for every --preprocess mode (resize_target), BICUBIC"""
        t = resize_target(self.opt, img.shape[0], img.shape[1])
        return img if t is None else self.resize(img, t[0], t[1])

    def _dev_tables(self, n_in, n_out, filter='bicubic'):
        key = (n_in, n_out, filter)
        if key not in self._tables:
            b, c, k = resample_coeffs(n_in, n_out, filter)
            self._tables[key] = (torch.from_numpy(b).to(self.device), torch.from_numpy(c).to(self.device), k)
        return self._tables[key]

    def resize(self, img, out_h, out_w, filter='bicubic'):
        """img: uint8 device tensor view [h, w, 3] (rows may be strided) -> uint8 [out_h, out_w, 3]"""
        h, w, _ = img.shape
        assert img.stride(2) == 1 and img.stride(1) == 3
        dst = torch.empty((out_h, out_w, 3), dtype=torch.uint8, device=self.device)
        hb, hc, hk = self._dev_tables(w, out_w, filter) if out_w != w else (None, None, 0)
        vb, vc, vk = self._dev_tables(h, out_h, filter) if out_h != h else (None, None, 0)
        tmp = torch.empty((h, out_w, 3), dtype=torch.uint8, device=self.device) if (hk and vk) else None
        p = lambda t: t.data_ptr() if t is not None else None
        check(ops.lib().gcc_resample_u8(img.data_ptr(), h, w, img.stride(0), dst.data_ptr(), out_h, out_w, p(hb), p(hc), hk,
                                        p(vb), p(vc), vk, p(tmp), ops.stream()), 'gcc_resample_u8')
        return dst

    def finish(self, img, crop_pos, crop, flip, nhwc=None):
        """crop + flip + ToTensor + Normalize of a uint8 [h, w, 3] device image -> fp32 [3, crop, crop]"""
        h, w, _ = img.shape
        x, y = crop_pos
        tw = th = crop
        if not (w > tw or h > th):          # __crop (:137-143) only crops when the image is larger
            x, y, tw, th = 0, 0, w, h
        out = torch.empty((3, th, tw), dtype=torch.float32, device=self.device)
        check(ops.lib().gcc_crop_flip_normalize(img.data_ptr(), h, w, img.stride(0), x, y, th, tw, int(bool(flip)), out.data_ptr(),
                                                nhwc.data_ptr() if nhwc is not None else None,
                                                nhwc.stride(3) if nhwc is not None else 0, ops.stream()), 'gcc_crop_flip_normalize')
        return out

    def __call__(self, AB, params=None):
        """AB: uint8 [h, 2w, 3] (host or device).  params: get_params() result; drawn here if None."""
        opt = self.opt
        AB = AB.to(self.device, non_blocking=True)
        if AB.dtype != torch.uint8 or AB.dim() != 3 or AB.shape[2] != 3:
            raise GccError('expected a decoded RGB image, uint8 [h, w, 3]')
        AB = AB.contiguous()
        h, w = AB.shape[0], AB.shape[1]
        w2 = int(w / 2)
        halves = (AB[:, :w2], AB[:, w2:w])                         # AB.crop((0, 0, w2, h)), AB.crop((w2, 0, w, h))
        if params is None:
            params = get_params(opt, (w2, h))
        out = {}
        for name, img in zip(('A', 'B'), halves):
            img = self.pre_resize(img)
            flip = (not opt.no_flip) and params['flip']
            if 'crop' in opt.preprocess:
                out[name] = self.finish(img, params['crop_pos'], opt.crop_size, flip)
            else:
                out[name] = self.finish(img, (0, 0), max(img.shape[0], img.shape[1]) + 1, flip)
        return out

    def batch(self, images, paths=None):
        """a list of decoded AB images -> the batch dict of set_input"""
        items = [self(im) for im in images]
        paths = paths or [''] * len(items)
        return {'A': torch.stack([i['A'] for i in items]), 'B': torch.stack([i['B'] for i in items]),
                'A_paths': list(paths), 'B_paths': list(paths)}


class UnalignedGpuPipeline(AlignedGpuPipeline):
    """``pipe(A_img, B_img)`` -> {'A', 'B'}: the per-image work of UnalignedDataset.__getitem__
    (data/unaligned_dataset.py:45-75) after decoding: each image goes through get_transform(opt) with params=None, i.e.
    Resize, torchvision's RandomCrop and RandomHorizontalFlip.  Their random draws are restated from torchvision
    (absent from this image, so that sequence is unpinned): RandomCrop.get_params draws
    ``torch.randint(0, h - th + 1, (1,))`` then ``torch.randint(0, w - tw + 1, (1,))`` unless the image already has the
    crop size, RandomHorizontalFlip flips when ``torch.rand(1) < 0.5``.  ``index_B`` is the caller's
    (``random.randint(0, B_size - 1)`` unless serial_batches, :62-65)."""

    def _one(self, img):
        opt = self.opt
        img = img.to(self.device, non_blocking=True).contiguous()
        if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
            raise GccError('expected a decoded RGB image, uint8 [h, w, 3]')
        img = self.pre_resize(img)
        x = y = 0
        crop = opt.crop_size
        h, w = img.shape[0], img.shape[1]
        if 'crop' in opt.preprocess:
            if not (h == crop and w == crop):
                if h < crop or w < crop:
                    raise ValueError('Required crop size %s is larger than input image size %s' % ((crop, crop), (h, w)))
                y = int(torch.randint(0, h - crop + 1, size=(1,)).item())
                x = int(torch.randint(0, w - crop + 1, size=(1,)).item())
        else:
            crop = max(h, w) + 1
        flip = (not opt.no_flip) and bool(torch.rand(1) < 0.5)
        if 'crop' in opt.preprocess:
            out = torch.empty((3, crop, crop), dtype=torch.float32, device=self.device)
            check(ops.lib().gcc_crop_flip_normalize(img.data_ptr(), h, w, img.stride(0), x, y, crop, crop, int(flip),
                                                    out.data_ptr(), None, 0, ops.stream()), 'gcc_crop_flip_normalize')
            return out
        return self.finish(img, (0, 0), crop, flip)

    def __call__(self, A_img, B_img):
        return {'A': self._one(A_img), 'B': self._one(B_img)}


# ------------------------------------------------------------------------------------------------------------------
# Loader glue (data/image_folder.py:13-33, data/aligned_dataset.py:20-58, data/__init__.py:52-100): file listing and
# decoding stay on the host (PIL; out of scope), everything after the decode runs on the GPU.
# ------------------------------------------------------------------------------------------------------------------
IMG_EXTENSIONS = ['.jpg', '.JPG', '.jpeg', '.JPEG', '.png', '.PNG', '.ppm', '.PPM', '.bmp', '.BMP', '.tif', '.TIF', '.tiff', '.TIFF']


def make_dataset(dir, max_dataset_size=float('inf')):
    """data/image_folder.py:24-33"""
    import os
    assert os.path.isdir(dir), '%s is not a valid directory' % dir
    images = []
    for root, _, fnames in sorted(os.walk(dir)):
        for fname in fnames:
            if any(fname.endswith(e) for e in IMG_EXTENSIONS):
                images.append(os.path.join(root, fname))
    return images[:min(max_dataset_size, len(images))]


class _GpuFileLoader:
    """File listing + host decode (PIL on ``decode_threads`` threads; it releases the GIL) + GPU transforms + collation.
    Order: ``serial_batches`` -> index order, else a permutation drawn like torch's RandomSampler inside a single-process
    DataLoader (the per-epoch base seed first, then the sampler's seed from the default generator, ``torch.randperm`` on a
    generator of its own).  Per-item random draws happen in item order, as with num_workers = 0."""

    def __init__(self, opt, decode_threads=8):
        self.opt = opt
        self.threads = int(decode_threads)

    @staticmethod
    def decode(path):
        from PIL import Image
        return torch.from_numpy(np.asarray(Image.open(path).convert('RGB')).copy())

    def order(self):
        n = len(self)
        if self.opt.serial_batches:
            return list(range(n))
        torch.empty((), dtype=torch.int64).random_()          # DataLoader's per-epoch base seed for workers: drawn first
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
        g = torch.Generator()
        g.manual_seed(seed)
        return torch.randperm(n, generator=g).tolist()

    def item(self, index, pool):
        raise NotImplementedError

    @staticmethod
    def collate(items):
        out = {}
        for k in items[0]:
            v = [it[k] for it in items]
            out[k] = torch.stack(v) if torch.is_tensor(v[0]) else list(v)
        return out

    def rank_order(self):
        """this rank's share of the epoch: ONE permutation (rank 0's draw, broadcast) dealt round-robin, truncated so that
        every rank sees the same number of items -- an epoch is one pass over the dataset for the whole job, which keeps
        the per-epoch LR / EMA-beta / arch-LR schedules those of the reference"""
        from .. import dist as gdist
        order = self.order()
        w = gdist.world_size()
        # only TRAINING loaders are dealt over the ranks: an evaluation loader (a non-train phase, or `self.shard = False` set by
        # whoever builds it) is walked whole by whoever iterates it -- no collective inside, so one rank may evaluate alone
        # (ADVICE r2).  --serial_batches only fixes the ORDER (a data-parallel training run with it used to feed the
        # whole dataset to every rank).
        shard = bool(getattr(self.opt, 'isTrain', False)) and getattr(self.opt, 'phase', 'train') == 'train' and \
            getattr(self, 'shard', True)
        if w > 1 and shard:
            import torch.distributed as dist
            box = [order]
            dist.broadcast_object_list(box, src=0)
            order = box[0]
            assert len(order) >= w, 'dataset of %d items cannot be dealt over %d ranks' % (len(order), w)
            order = order[:len(order) // w * w][gdist.rank()::w]
        return order

    def _produce(self, fn):
        """run the GPU part of a batch on the loader's own stream and hand the batch over with a 'ready' event (the
        models' set_input orders every consuming stream behind it): the online teacher's stream does not have to wait
        for the student's queue, and the batch is never read before its producer finished"""
        if getattr(self, '_stream', None) is None:
            self._stream = torch.cuda.Stream()
        with torch.cuda.stream(self._stream):
            batch = fn()
            ev = torch.cuda.Event()
            ev.record(self._stream)
        batch['ready'] = ev
        return batch

    def __iter__(self):
        from concurrent.futures import ThreadPoolExecutor
        order, bs = self.rank_order(), int(self.opt.batch_size)
        with ThreadPoolExecutor(max(1, self.threads)) as pool:
            for b in range(0, len(order), bs):
                yield self._produce(lambda: self.collate([self.item(i, pool) for i in order[b:b + bs]]))


class AlignedGpuDataLoader(_GpuFileLoader):
    """What ``create_dataset(opt)`` returns for ``--dataset_mode aligned`` (data/aligned_dataset.py:20-58): iterating
    yields the batch dicts of set_input; augmentation parameters come from get_params in item order."""

    def __init__(self, opt, device=None, decode_threads=8):
        import os
        super().__init__(opt, decode_threads)
        self.paths = sorted(make_dataset(os.path.join(opt.dataroot, opt.phase), opt.max_dataset_size))
        self.pipe = AlignedGpuPipeline(opt, device)

    def __len__(self):
        return len(self.paths)

    LOOKAHEAD = 4          # batches decoded ahead of the GPU (bounded: a whole epoch of decoded images does not fit in host memory)

    def __iter__(self):
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        order, bs = self.rank_order(), int(self.opt.batch_size)
        with ThreadPoolExecutor(max(1, self.threads)) as pool:
            window, nxt = deque(), 0
            for b in range(0, len(order), bs):
                while nxt < len(order) and nxt < b + (self.LOOKAHEAD + 1) * bs:
                    window.append(pool.submit(self.decode, self.paths[order[nxt]]))
                    nxt += 1
                imgs = [window.popleft().result() for _ in order[b:b + bs]]
                paths = [self.paths[i] for i in order[b:b + bs]]
                yield self._produce(lambda: self.pipe.batch(imgs, paths))


class UnalignedGpuDataLoader(_GpuFileLoader):
    """``--dataset_mode unaligned`` (data/unaligned_dataset.py:20-78): <phase>A / <phase>B directories, B drawn with
    ``random.randint(0, B_size - 1)`` unless serial_batches, length max(A_size, B_size)"""

    def __init__(self, opt, device=None, decode_threads=8):
        import os
        super().__init__(opt, decode_threads)
        self.A_paths = sorted(make_dataset(os.path.join(opt.dataroot, opt.phase + 'A'), opt.max_dataset_size))
        self.B_paths = sorted(make_dataset(os.path.join(opt.dataroot, opt.phase + 'B'), opt.max_dataset_size))
        self.pipe = UnalignedGpuPipeline(opt, device)

    def __len__(self):
        return max(len(self.A_paths), len(self.B_paths))

    def item(self, index, pool):
        a = self.A_paths[index % len(self.A_paths)]
        ib = index % len(self.B_paths) if self.opt.serial_batches else random.randint(0, len(self.B_paths) - 1)
        b = self.B_paths[ib]
        fa, fb = pool.submit(self.decode, a), pool.submit(self.decode, b)
        it = self.pipe(fa.result(), fb.result())
        it['A_paths'], it['B_paths'] = a, b
        return it


class SRGpuDataLoader(_GpuFileLoader):
    """``--dataset_mode sr`` (data/sr_dataset.py:123-184): every file of <dataroot>/<phase>, sorted"""

    def __init__(self, opt, device=None, decode_threads=8):
        import os
        super().__init__(opt, decode_threads)
        self.folder = os.path.join(opt.dataroot, opt.phase)
        self.names = sorted(os.listdir(self.folder))
        self.pipe = SRGpuPipeline(opt, device)

    def __len__(self):
        return len(self.names)

    def item(self, index, pool):
        import os
        it = self.pipe(self.decode(os.path.join(self.folder, self.names[index])))
        it['lr_names'] = it['hr_names'] = self.names[index]
        return it


class SAGpuDataLoader(_GpuFileLoader):
    """``--dataset_mode sa`` (data/sa_dataset.py:9-55)"""

    def __init__(self, opt, device=None, decode_threads=8):
        import os
        super().__init__(opt, decode_threads)
        self.folder = os.path.join(opt.dataroot, opt.phase)
        self.names = sorted(os.listdir(self.folder))
        self.pipe = SAGpuPipeline(opt, device)

    def __len__(self):
        return len(self.names)

    def item(self, index, pool):
        import os
        it = self.pipe(self.decode(os.path.join(self.folder, self.names[index])))
        it['img_path'] = self.names[index]
        return it


def create_dataset(opt, device=None):
    """data/__init__.py:52-57 for the four dataset modes"""
    cls = {'aligned': AlignedGpuDataLoader, 'unaligned': UnalignedGpuDataLoader, 'sr': SRGpuDataLoader, 'sa': SAGpuDataLoader}
    if opt.dataset_mode not in cls:
        raise NotImplementedError('dataset_mode %s' % opt.dataset_mode)
    return cls[opt.dataset_mode](opt, device)


IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def _convert(img, form, device, mean=None, std=None):
    """uint8 [h, w, 3] device image -> fp32 [3, h, w] through gcc_crop_convert (the whole image, no flip)"""
    import ctypes
    h, w, _ = img.shape
    out = torch.empty((3, h, w), dtype=torch.float32, device=device)
    arr = lambda v: (ctypes.c_float * 3)(*v) if v is not None else None
    check(ops.lib().gcc_crop_convert(img.data_ptr(), h, w, img.stride(0), 0, 0, h, w, 0, form, arr(mean), arr(std),
                                     out.data_ptr(), None, 0, ops.stream()), 'gcc_crop_convert')
    return out


class SRGpuPipeline(AlignedGpuPipeline):
    """ImageTransforms of data/sr_dataset.py:66-121 on a decoded image: a random (train) or largest divisible centre (test)
    crop is the HR image, PIL's BICUBIC downscale by ``upscale_factor`` the LR image; HR -> '[-1, 1]', LR ->
    'imagenet-norm' (the reference's defaults, options.py:119-120).  ``pipe(img)`` -> {'lr', 'hr'}."""

    def __init__(self, opt, device=None):
        self.opt = opt
        if not torch.cuda.is_available():
            raise GccError('gcc_amd runs on MI355X only (no CPU path): need a visible GPU')
        self.device = device or torch.device('cuda', torch.cuda.current_device())
        self._tables = {}
        if (opt.lr_img_type, opt.hr_img_type) != ('imagenet-norm', '[-1, 1]'):
            raise NotImplementedError('the MI355X pipeline covers the reference defaults: lr imagenet-norm, hr [-1, 1]')

    def __call__(self, img):
        opt = self.opt
        img = img.to(self.device, non_blocking=True).contiguous()
        h, w = img.shape[0], img.shape[1]
        s, crop = int(opt.upscale_factor), int(opt.image_size)
        if str(opt.phase).lower() == 'train':
            left = random.randint(1, w - crop)
            top = random.randint(1, h - crop)
            hr = img[top:top + crop, left:left + crop]
        else:
            xr, yr = w % s, h % s
            hr = img[yr // 2:yr // 2 + (h - yr), xr // 2:xr // 2 + (w - xr)]
        lr = self.resize(hr, int(hr.shape[0] / s), int(hr.shape[1] / s))
        return {'lr': _convert(lr, 0, self.device, IMAGENET_MEAN, IMAGENET_STD), 'hr': _convert(hr, 1, self.device)}


class SAGpuPipeline(AlignedGpuPipeline):
    """SADataset's transform (data/sa_dataset.py:26-48): CenterCrop(160) if --center_crop, Resize((crop_size, crop_size))
    with torchvision's default BILINEAR (PIL, antialiased), ToTensor, Normalize(.5, .5); z ~ torch.randn(z_dim) per item.
    ``pipe(img)`` -> {'z', 'real_img'}."""

    def __init__(self, opt, device=None):
        self.opt = opt
        if not torch.cuda.is_available():
            raise GccError('gcc_amd runs on MI355X only (no CPU path): need a visible GPU')
        self.device = device or torch.device('cuda', torch.cuda.current_device())
        self._tables = {}

    def __call__(self, img):
        opt = self.opt
        img = img.to(self.device, non_blocking=True).contiguous()
        h, w = img.shape[0], img.shape[1]
        if getattr(opt, 'center_crop', False):
            th = tw = 160
            if h < th or w < tw:
                raise NotImplementedError('CenterCrop(160) pads images smaller than the crop: not on the MI355X path')
            top, left = int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))       # torchvision.transforms.functional.center_crop
            img = img[top:top + th, left:left + tw]
        size = int(opt.crop_size)
        img = self.resize(img, size, size, filter='bilinear')
        z = torch.randn(int(opt.z_dim))
        return {'z': z.to(self.device), 'real_img': self.finish(img, (0, 0), size + 1, False)}
