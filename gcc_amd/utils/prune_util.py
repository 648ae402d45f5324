"""Pruning driver (reference utils/prune_util.py:20-63).  The per-layer integer logic (cfg lists) is
Pix2PixModel.scale_prune_cfg; the MAC budget search needs the thop-convention counter, which is
row (f1) of the scope table and not built yet."""


def prune(model, opt, logger):
    raise NotImplementedError('budgeted pruning (binary search on the threshold against a thop-convention MAC '
                              'count) is scheduled after the hot path (SURVEY.md section 8 f1); build the pruned '
                              'student directly with filter_cfgs/channel_cfgs from Pix2PixModel.scale_prune_cfg()')
