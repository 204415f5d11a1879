"""Prediction images that are written only when somebody reads them.

``MultiMAE.forward`` returns ``preds[task]`` as a (B, C, H, W) image (output_adapters.py:277-280 of the reference: the rearrangement
of out_proj's patch rows).  The training loop hands that tensor straight to a masked loss (run_pretraining_multimae.py:508-520), and
this engine's losses work on the adapter's patch rows (functions.PatHandle) -- nothing on the hot path ever reads the image.  At the
bench geometry the four images are 786 MB of f32 per step (semseg alone 427 MB: 133 classes at 56 x 56), written by
``unpatchify`` kernels for the API only (profiles/r03_pmc_traffic.json: 0.3 ms per step, 880 MB of traffic for the semseg image).

``LazyPrediction`` is a ``torch.Tensor`` subclass whose storage is allocated by the adapter but filled on first use: every torch
function that receives it first runs the deferred ``unpatchify`` (once, ordered behind the adapter's stream), except a whitelist of
metadata accessors (shape, dtype, device, size(), ...), ``record_stream`` and the no-op ``.float()`` of an f32 tensor -- exactly the
calls the loop makes on its way to the loss.  The patch rows it is filled from stay alive with it (they live in the adapter's
activation slab), so reading the image after ``backward()`` -- or a step later -- still works.  Results are plain tensors.
``engine.set_lazy_predictions(False)`` restores the eager write.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

_META_PROPS = ('shape', 'dtype', 'device', 'requires_grad', 'grad_fn', 'is_cuda', 'ndim', 'is_leaf', 'layout', '_version', 'output_nr',
               'is_sparse', 'is_quantized', 'is_meta', 'names', 'grad')
_META_METHODS = ('size', 'dim', 'numel', 'nelement', 'stride', 'is_contiguous', 'element_size', 'record_stream', 'is_floating_point',
                 'is_complex', 'get_device', '__len__', '__hash__', 'storage_offset', 'has_names', 'register_hook', 'retain_grad',
                 'requires_grad_', 'is_same_size', '_is_view', 'is_pinned', 'is_shared', 'dim_order', 'is_inference')


def _passthrough():
    s = set()
    for n in _META_PROPS:
        p = getattr(torch.Tensor, n, None)
        if p is not None and hasattr(p, '__get__'):
            s.add(p.__get__)
    for n in _META_METHODS:
        m = getattr(torch.Tensor, n, None)
        if m is not None:
            s.add(m)
    return s


_PASS = _passthrough()


class _LazyCloneFn(torch.autograd.Function):
    """clone() of a prediction that has not been written yet: new (unwritten) storage, identity gradient"""

    @staticmethod
    def forward(ctx, x):
        return torch.empty_like(x)

    @staticmethod
    def backward(ctx, g):
        return g


class LazyPrediction(torch.Tensor):
    """See the module docstring.  ``_mmae_fill``: callable that writes the image into this tensor's storage (None once done).
    ``clone()`` keeps the wrapper, the laziness and the patch-row side channel (``_mmae_pat``): torch DDP's output sink hands the
    training loop CLONES of the model's outputs (find_unused_parameters=True), and the loop's losses should still find the
    adapter's patch rows behind them."""

    @staticmethod
    def wrap(img: torch.Tensor, fill: Optional[Callable[[torch.Tensor], None]]) -> 'LazyPrediction':
        out = img.as_subclass(LazyPrediction)
        out._mmae_plain = img                  # the same storage as a plain tensor: what the deferred kernel writes through
        out._mmae_fill = fill
        out._mmae_v0 = img._version            # an in-place edit of the image detaches it from the patch rows (criterion._pat_handle)
        return out

    def _clone_keep(self) -> 'LazyPrediction':
        fill = getattr(self, '_mmae_fill', None)
        with torch._C.DisableTorchFunctionSubclass():
            new = _LazyCloneFn.apply(self) if fill is not None else torch.Tensor.clone(self)
        out = LazyPrediction.wrap(new, fill)
        if self.unmodified and hasattr(self, '_mmae_pat'):
            out._mmae_pat = self._mmae_pat
        return out

    @property
    def unmodified(self) -> bool:
        return torch.Tensor._version.__get__(self) == getattr(self, '_mmae_v0', -1)

    def materialize(self) -> None:
        fill = getattr(self, '_mmae_fill', None)
        if fill is not None:
            self._mmae_fill = None
            fill(self._mmae_plain)

    @property
    def materialized(self) -> bool:
        return getattr(self, '_mmae_fill', None) is None

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _PASS:
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        if func is torch.Tensor.float and not kwargs and len(args) == 1 and isinstance(args[0], LazyPrediction) \
                and torch.Tensor.dtype.__get__(args[0]) == torch.float32:
            return args[0]                      # preds[task].float() of an f32 prediction: the same object, still lazy
        if func is torch.Tensor.clone and len(args) == 1 and isinstance(args[0], LazyPrediction) and \
                kwargs.get('memory_format', torch.preserve_format) in (torch.preserve_format, torch.contiguous_format):
            return args[0]._clone_keep()

        def visit(a):
            if isinstance(a, LazyPrediction):
                a.materialize()
            elif isinstance(a, (list, tuple)):
                for b in a:
                    visit(b)
            elif isinstance(a, dict):
                for b in a.values():
                    visit(b)
        visit(args)
        visit(kwargs)
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)


def materialize(x) -> None:
    """Fill x now if it is a LazyPrediction that has not been written yet (entry points that hand x.data_ptr() to a kernel)."""
    if isinstance(x, LazyPrediction):
        x.materialize()
