"""`NeRFSemSegFPNHead` -- eval branch of reference gpnerf/semantic_branch.py (ctor :13-43, forward
:89-107) with the reference's state-dict names (`scale_heads.{i}.{j}.weight`,
`scale_heads.{i}.{j}.norm.{weight,bias}`, `predictor.{weight,bias}`).

Runs once per image on the rendered [1,512,h,w] feature image (about 8 GFLOP) as a
plain torch module on the HIP device: conv / GroupNorm / bilinear resize are library ops, not part of
the per-ray hot path that the HIP kernels of this package replace.  detectron2's `Conv2d(norm=...,
activation=...)` is restated as conv -> GroupNorm(32,128) -> ReLU and fvcore's c2_msra_fill as
kaiming_normal_(fan_out, relu).  The training-only distillation / depth-guided losses (:60-88) are
out of scope (loss code) and raise.  The reference hard-codes a 120x160 feature image (:95); here
the size is taken from the input (identical for 120x160).
"""
import ctypes as C

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from . import _lib


def semantic_labels(logits, dtype=torch.int64):
    """Label image of the head's logits [1,C,H,W] (or [C,H,W]): what the reference obtains as
    `sems.permute(0,2,3,1).argmax(dim=-1)` (eval_gpnerf.py:201-206, loss.py:293-300), in one HIP
    kernel over the NCHW logits.  dtype int64 (reference) or uint8 (1 byte per pixel, for gathers)."""
    x = _lib.dev_f32(_lib.as_f32c(logits), "logits")
    if x.dim() == 4:
        if x.shape[0] != 1:
            raise ValueError("semantic_labels: one image at a time (batch %d)" % x.shape[0])
        x = x[0]
    if x.dim() != 3:
        raise ValueError("semantic_labels: expected [1,C,H,W] or [C,H,W]")
    Cn, H, W = x.shape
    if dtype not in (torch.int64, torch.uint8):
        raise ValueError("semantic_labels: dtype must be torch.int64 or torch.uint8")
    out = torch.empty((H, W), dtype=dtype, device=x.device)
    p64 = _lib.ptr(out) if dtype == torch.int64 else None
    p8 = _lib.ptr(out) if dtype == torch.uint8 else None
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().gpn_argmax_labels(_lib.ptr(x), Cn, H * W, p64, p8, _lib.stream()),
                   "gpn_argmax_labels")
    return out


class _ConvGNReLU(nn.Conv2d):
    def __init__(self, cin, cout):
        super().__init__(cin, cout, kernel_size=3, stride=1, padding=1, bias=False)
        self.norm = nn.GroupNorm(32, cout)
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        return F.relu(self.norm(super().forward(x)))


class NeRFSemSegFPNHead(nn.Module):
    def __init__(self, args, feature_strides=[2, 4, 8, 16], feature_channels=[128, 128, 128, 128],
                 num_classes=20):
        super().__init__()
        self.n_p = getattr(args, "n_p", 0)
        conv_dims = 128
        self.common_stride = 2
        self.scale_heads = nn.ModuleList()
        for stride, channels in zip(feature_strides, feature_channels):
            ops = []
            head_length = max(1, int(np.log2(stride) - np.log2(self.common_stride)))
            for k in range(head_length):
                ops.append(_ConvGNReLU(channels if k == 0 else conv_dims, conv_dims))
                if stride != self.common_stride:
                    ops.append(nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False))
            self.scale_heads.append(nn.Sequential(*ops))
        self.predictor = nn.Conv2d(conv_dims, num_classes + 1, kernel_size=1, stride=1, padding=0)

    @torch.no_grad()
    def forward(self, deep_feats, agg_sem_feats=None, select_inds=None):
        """deep_feats [1,512,h,w] -> logits [1,num_classes+1,2h,2w] (reference :89-107)."""
        if select_inds is not None:
            raise NotImplementedError("training branch (distillation / depth-guided losses, reference "
                                      "semantic_branch.py:60-88) is outside the rendering hot path")
        h, w = deep_feats.shape[-2:]
        flat = deep_feats.reshape(1, deep_feats.shape[1], -1).squeeze(0).permute(1, 0)
        x = None
        for i, chunk in enumerate(torch.chunk(flat, 4, dim=1)):
            chunk = chunk.reshape(h, w, chunk.shape[-1]).permute(2, 0, 1).unsqueeze(0)
            if i == 0:
                x = self.scale_heads[i](chunk)
            else:
                chunk = F.interpolate(chunk, scale_factor=1 / (2 ** i), mode="bilinear",
                                      align_corners=True, recompute_scale_factor=True)
                x = x + self.scale_heads[i](chunk)
        out = self.predictor(x)
        return F.interpolate(out, scale_factor=2, mode="bilinear", align_corners=True)

    @torch.no_grad()
    def labels(self, deep_feats, dtype=torch.int64):
        """Eval-branch logits followed by the label epilogue (`semantic_labels`): [2h, 2w] labels."""
        return semantic_labels(self.forward(deep_feats), dtype)
