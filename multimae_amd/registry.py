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
    _into_reference_registry(fn)
    return fn


def _reference_registry():
    ext = sys.modules.get('utils.registry')
    if ext is None:
        # deployed next to the reference tree: its registry is importable as utils.registry (the reference's own
        # multimae/multimae.py:28 imports it at this point too)
        try:
            import importlib
            ext = importlib.import_module('utils.registry')
        except Exception:
            ext = None
    return ext if ext is not None and hasattr(ext, '_model_entrypoints') else None


def _into_reference_registry(fn: Callable) -> None:
    ext = _reference_registry()
    if ext is not None:
        ext._model_entrypoints[fn.__name__] = fn        # live reference registry: take over the name


def sync_reference_registry() -> int:
    """(Re-)insert every registered factory into the reference's ``utils.registry`` if it is importable NOW.  ``register_model`` does
    this at decoration time, which is too early when the engine was imported before the reference tree was on ``sys.path`` (found by
    running the unmodified script, tools/run_reference_script_dryrun.py); the drop-in ``multimae.multimae`` module calls this when the
    training script imports it -- run_pretraining_multimae.py:34-36 imports ``utils`` first.  Returns the number of names placed."""
    ext = _reference_registry()
    if ext is None:
        return 0
    for name, fn in _entrypoints.items():
        ext._model_entrypoints[name] = fn
    return len(_entrypoints)


def is_model(name: str) -> bool:
    return name in _entrypoints


def model_entrypoint(name: str) -> Callable:
    return _entrypoints[name]


def list_models():
    return sorted(_entrypoints)


def create_model(model_name: str, pretrained: bool = False, checkpoint_path: str = '', scriptable=None, exportable=None, no_jit=None,
                 **kwargs):
    """create_model(name, input_adapters=..., output_adapters=..., num_global_tokens=..., drop_path_rate=...) with the signature of
    the reference's builder (utils/model_builder.py:29-76): the timm-era arguments are accepted -- `drop_connect_rate` is mapped to
    `drop_path_rate` with the reference's warning, `bn_tf` / `bn_momentum` / `bn_eps` are dropped, `scriptable` / `exportable` /
    `no_jit` are ignored (as there).  The reference builder accepts `pretrained` and `checkpoint_path` and then never uses them;
    here `checkpoint_path` (a reference-format checkpoint file: {'model': state_dict, ...} or a bare state_dict) IS loaded after
    construction (checkpoint.load_checkpoint: strict key match, position tables resized), `pretrained=True` has nothing to fetch
    for these factories and is ignored with a warning."""
    if '/' in model_name or ':' in model_name:                 # "source:name" / hub prefixes of the reference's split_model_name
        model_name = model_name.replace(':', '/').split('/')[-1]
    for k in ('bn_tf', 'bn_momentum', 'bn_eps'):
        kwargs.pop(k, None)
    drop_connect_rate = kwargs.pop('drop_connect_rate', None)
    if drop_connect_rate is not None and kwargs.get('drop_path_rate', None) is None:
        print("WARNING: 'drop_connect' as an argument is deprecated, please use 'drop_path'."
              " Setting drop_path to %f." % drop_connect_rate)
        kwargs['drop_path_rate'] = drop_connect_rate
    if not is_model(model_name):
        raise RuntimeError('Unknown model (%s)' % model_name)
    model = model_entrypoint(model_name)(**kwargs)
    if pretrained:
        import warnings
        warnings.warn(f'create_model({model_name!r}, pretrained=True): no pretrained weights are registered for this factory '
                      '(the reference builder ignores the flag too); pass checkpoint_path=... to load a checkpoint')
    if checkpoint_path:
        from .checkpoint import load_checkpoint
        load_checkpoint(checkpoint_path, model)
    return model
