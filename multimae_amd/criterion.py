"""Masked losses (mirror of the reference's ``multimae/criterion.py`` API) on HIP kernels.

Reference: MaskedCrossEntropyLoss criterion.py:23-57, MaskedMSELoss :60-114, MaskedL1Loss
:117-171.  Semantics reproduced: per-sample masked mean of the
channel-mean error, then the mean over samples that have at least one masked token (nanmean);
norm_pix uses the unbiased patch variance with eps 1e-6 inside the sqrt.

Deliberate deviation (documented in DESIGN.md): when NO token of the batch is masked the
reference returns ``torch.tensor(0)`` (int64, after a host sync); this engine returns a float32
zero with zero gradient and never synchronises.
"""
from __future__ import annotations

import torch
from torch import nn

from .functions import MaskedCEFn, MaskedCEPatFn, MaskedPixelLossFn, MaskedPixelLossPatFn
from .lazy import materialize as _materialize


def _pat_handle(x: torch.Tensor, patch: int, ce: bool = False):
    """The adapter's patch rows behind prediction x (functions.PatHandle), if x is exactly what an output adapter returned
    (``preds[task].float()`` of an f32 tensor is the same object) and the loss's patch grid is the adapter's.  Anything else --
    a modified prediction, a copy made any other way than ``clone()`` -- takes the image-domain loss (same value, gradient rows
    through the f32 image instead of the adapter's activation dtype).  The CLONES torch DDP's output sink returns with
    find_unused_parameters=True keep the side channel (lazy.LazyPrediction.clone), so the reference loop under DDP runs the same
    loss kernels as the native loop (tests/test_reference_loop_gpu.py)."""
    h = getattr(x, '_mmae_pat', None)
    if h is None or not h.matches(x, patch) or not getattr(x, 'unmodified', True):
        return None
    if ce and (patch * patch > 64 or (patch * patch) & (patch * patch - 1)):
        return None
    return h


def _ones_mask(x: torch.Tensor, scale: int) -> torch.Tensor:
    H, W = x.shape[-2:]
    return torch.ones((x.shape[0], (H // scale) * (W // scale)), device=x.device, dtype=torch.int64)


class MaskedCrossEntropyLoss(nn.Module):
    """Cross-entropy loss with masking (patch_size, stride, label_smoothing)."""

    def __init__(self, patch_size: int = 16, stride: int = 1, label_smoothing: float = 0.0):
        super().__init__()
        self.patch_size = patch_size
        self.stride = stride
        self.scale_factor = patch_size // stride
        self.label_smoothing = label_smoothing

    def forward(self, input, target, mask=None):
        if mask is None:
            mask = _ones_mask(input, self.scale_factor)      # plain mean == masked mean with an all-ones mask
        h = _pat_handle(input, self.scale_factor, ce=True)
        if h is not None:
            return MaskedCEPatFn.apply(h.token, h, target, mask, self.scale_factor, float(self.label_smoothing))
        _materialize(input)                                 # image-domain loss: the kernels read the image itself
        return MaskedCEFn.apply(input, target, mask, self.scale_factor, float(self.label_smoothing))


class _MaskedPixelLoss(nn.Module):
    kind = 0

    def __init__(self, patch_size: int = 16, stride: int = 1, norm_pix=False):
        super().__init__()
        self.patch_size = patch_size
        self.stride = stride
        self.scale_factor = patch_size // stride
        self.norm_pix = norm_pix

    def forward(self, input, target, mask=None):
        if mask is None:
            mask = _ones_mask(input, self.scale_factor)
        h = _pat_handle(input, self.scale_factor)
        if h is not None:
            return MaskedPixelLossPatFn.apply(h.token, h, target, mask, self.kind, bool(self.norm_pix), self.scale_factor)
        _materialize(input)
        return MaskedPixelLossFn.apply(input, target, mask, self.kind, bool(self.norm_pix), self.scale_factor)


class MaskedMSELoss(_MaskedPixelLoss):
    """MSE loss with masking (patch_size, stride, norm_pix)."""
    kind = 0


class MaskedL1Loss(_MaskedPixelLoss):
    """L1 loss with masking (patch_size, stride, norm_pix)."""
    kind = 1
