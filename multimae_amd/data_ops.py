"""Batch preparation steps the training loop performs right before the model call.

``truncated_depth_standardize`` replaces run_pretraining_multimae.py:487-492 -- a full ``torch.sort`` of the 50 176 values of
every depth map each step -- with one selection kernel (``mmae_depth_standardize``, csrc/depth.hip).  Drop-in for the loop:

    if standardize_depth and 'depth' in tasks_dict:
        tasks_dict['depth'] = truncated_depth_standardize(tasks_dict['depth'])
"""
from __future__ import annotations

import torch

from . import _lib, ops


def truncated_depth_standardize(depth: torch.Tensor, lo: float = 0.1, hi: float = 0.9, eps: float = 1e-6) -> torch.Tensor:
    """(B, C, H, W) depth -> standardised with the mean / unbiased variance of each sample's values of rank
    [int(lo*n), int(hi*n)), n = C*H*W."""
    ops._require_gpu(depth, 'depth')
    B = depth.shape[0]
    x = depth.contiguous().float()
    n = x.numel() // B
    y = torch.empty_like(x)
    _lib.check(_lib.load().mmae_depth_standardize(x.data_ptr(), y.data_ptr(), B, n, int(lo * n), int(hi * n), eps, ops._stream()),
               'depth_standardize')
    return y.view(depth.shape)
