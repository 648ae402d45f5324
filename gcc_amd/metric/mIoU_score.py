"""mIoU arithmetic on the GPU (metric/mIoU_score.py:108-109, 163-167, 196-218): class argmax, confusion matrix, IoU."""
import numpy as np
import torch

from .. import ops
from .._lib import GccError, check


def _dev():
    if not torch.cuda.is_available():
        raise GccError('gcc_amd runs on MI355X only (no CPU path): need a visible GPU')
    return torch.device('cuda', torch.cuda.current_device())


def argmax_classes(scores):
    """scores [N, C, H, W] fp32 device tensor -> int32 [N, H, W] (numpy.argmax(axis=1) semantics)"""
    s = scores.float().contiguous()
    N, C, H, W = s.shape
    pred = torch.empty((N, H, W), dtype=torch.int32, device=s.device)
    check(ops.lib().gcc_argmax_channels(s.data_ptr(), N, C, H * W, pred.data_ptr(), ops.stream()), 'gcc_argmax_channels')
    return pred


def fast_hist(pred, label, n, hist=None):
    """metric/mIoU_score.py:163-167 on flat int arrays / tensors; returns (and accumulates into) an int64 [n, n] device
    tensor"""
    dev = _dev()
    to = lambda a: (a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))).to(dev).to(torch.int32).contiguous().reshape(-1)
    p, l = to(pred), to(label)
    assert p.numel() == l.numel()
    if hist is None:
        hist = torch.zeros((n, n), dtype=torch.int64, device=dev)
    check(ops.lib().gcc_confusion_hist(p.data_ptr(), l.data_ptr(), p.numel(), n, hist.data_ptr(), ops.stream()),
          'gcc_confusion_hist')
    return hist


def per_class_iu(hist):
    """metric/mIoU_score.py:108-109 (a class absent from both prediction and label gives nan, as there)"""
    h = hist.detach().cpu().numpy().astype(np.float64) if torch.is_tensor(hist) else np.asarray(hist, dtype=np.float64)
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.diag(h) / (h.sum(1) + h.sum(0) - np.diag(h))


def test(fakes, names, model, device, table_path='datasets/table.txt', data_dir='database/cityscapes', batch_size=1,
         num_workers=8, num_classes=19, use_tqdm=True, dataset=None):
    """metric/mIoU_score.py:196-218.  The reference's SegList reads label images from disk and resizes the score maps to
    2048x1024 with PIL; ``dataset`` yields (image batch, label batch [N, H, W] int) directly and the labels' size is the
    evaluation size."""
    if dataset is None:
        raise NotImplementedError('pass dataset=: the label files of the reference\'s SegList are host I/O (out of scope)')
    if hasattr(model, 'eval'):
        model.eval()
    hist = None
    with torch.no_grad():
        for image, label in dataset:
            final = model(image.to(device))[0]
            if final.shape[-2:] != label.shape[-2:]:
                final = torch.nn.functional.interpolate(final, size=label.shape[-2:], mode='bilinear', align_corners=False)
            hist = fast_hist(argmax_classes(final), label, num_classes, hist)
    ious = per_class_iu(hist) * 100
    return round(float(np.nanmean(ious)), 2)
