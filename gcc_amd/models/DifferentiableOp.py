"""DifferentiableOP: per-channel selective-activation gate (reference models/DifferentiableOp.py:34-59).

Only the parameter (alpha), the threshold and the host-visible helpers live here; the gate itself
(y = x * m[c], straight-through d/dalpha = sum dy*x, dx = dy*m) is fused into the BatchNorm /
activation HIP kernels (gcc_bnact_fwd / gcc_bnact_bwd), see gcc_amd/engine.py."""
import torch
import torch.nn as nn

from .. import ops


class DifferentiableOP(nn.Module):
    def __init__(self, output_channel, threshold=0.5):
        super().__init__()
        self.alpha = nn.Parameter(torch.ones(output_channel), requires_grad=True)
        self.threshold = threshold * torch.ones(1)      # plain attribute: not in the state_dict (:40)

    def clip_alpha(self):
        """alpha.clip_(0, 1)  (:51-53)"""
        if self.alpha.is_cuda:
            ops.clamp_(self.alpha.data, 0.0, 1.0)
        else:
            self.alpha.data.clamp_(0, 1)

    def anneal_threshold(self):
        pass

    def get_current_mask(self):
        """(sign(alpha - tau) + 1) / 2   (:58-59)"""
        tau = float(self.threshold)
        if self.alpha.is_cuda:
            m = torch.empty_like(self.alpha.data)
            ops.gate_mask(self.alpha.data, tau, m)
            return m
        return (torch.sign(self.alpha.detach() - tau) + 1) / 2
