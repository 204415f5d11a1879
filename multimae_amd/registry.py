"""Model registry (timm-style ``register_model`` / ``create_model``), the boundary the
reference's training script uses: utils/registry.py:26-51, utils/model_builder.py:29-76,
call site run_pretraining_multimae.py:285-291.

When the reference's own ``utils.registry`` is importable (drop-in deployment next to the
reference tree) the factories are ALSO registered there, so the unmodified
``utils.create_model('pretrain_multimae_base', ...)`` resolves to this engine.
"""
from __future__ import annotations

import sys
from typing import Callable, Dict

_entrypoints: Dict[str, Callable] = {}


def register_model(fn: Callable) -> Callable:
    _entrypoints[fn.__name__] = fn
    mod = sys.modules.get(fn.__module__)
    if mod is not None:
        names = getattr(mod, '__all__', None)
        if names is None:
            mod.__all__ = [fn.__name__]
        elif fn.__name__ not in names:
            names.append(fn.__name__)
    ext = sys.modules.get('utils.registry')
    if ext is None:
        # deployed next to the reference tree: its registry is importable as utils.registry (the reference's own
        # multimae/multimae.py:28 imports it at this point too)
        try:
            import importlib
            ext = importlib.import_module('utils.registry')
        except Exception:
            ext = None
    if ext is not None and hasattr(ext, '_model_entrypoints'):
        ext._model_entrypoints[fn.__name__] = fn        # live reference registry: take over the name
    return fn


def is_model(name: str) -> bool:
    return name in _entrypoints


def model_entrypoint(name: str) -> Callable:
    return _entrypoints[name]


def list_models():
    return sorted(_entrypoints)


def create_model(model_name: str, pretrained: bool = False, checkpoint_path: str = '', **kwargs):
    """create_model(name, input_adapters=..., output_adapters=..., num_global_tokens=..., drop_path_rate=...)."""
    if pretrained or checkpoint_path:
        raise NotImplementedError('pretrained / checkpoint_path loading is outside the pre-training hot path; '
                                  'use model.load_state_dict on a reference-format checkpoint')
    kwargs = {k: v for k, v in kwargs.items() if v is not None}
    if not is_model(model_name):
        raise RuntimeError('Unknown model (%s)' % model_name)
    return model_entrypoint(model_name)(**kwargs)
