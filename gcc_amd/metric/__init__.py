"""Evaluation arithmetic on MI355X -- the reference's ``metric`` package surface (metric/__init__.py:8-25) over
libgcc_hip.so.  The evaluator networks (Inception-v3 for FID, DRN for mIoU) are external inputs, exactly as their
weights are in the reference: ``model`` is any callable with the reference models' output convention."""
import torch

from .fid_score import _compute_statistics_of_ims, calculate_frechet_distance
from .mIoU_score import test
from ..utils import util


def get_fid(fakes, model, npz, device, batch_size=1, use_tqdm=True):
    """metric/__init__.py:8-14"""
    m1, s1 = npz['mu'], npz['sigma']
    fakes = torch.cat(fakes, dim=0)
    m2, s2 = _compute_statistics_of_ims(util.tensor2imgs(fakes).astype(float), model, batch_size, 2048, device,
                                        use_tqdm=use_tqdm)
    return float(calculate_frechet_distance(m1, s1, m2, s2))


def get_mIoU(fakes, names, model, device, table_path='datasets/table.txt', data_dir='database/cityscapes', batch_size=1,
             num_workers=8, num_classes=19, use_tqdm=True, dataset=None):
    """metric/__init__.py:16-25.  ``dataset`` replaces the reference's SegList (image files on disk, out of scope): an
    iterable of (image batch, label batch)."""
    fakes = torch.cat(fakes, dim=0)
    return float(test(util.tensor2imgs(fakes), names, model, device, table_path=table_path, data_dir=data_dir,
                      batch_size=batch_size, num_workers=num_workers, num_classes=num_classes, use_tqdm=use_tqdm,
                      dataset=dataset))
