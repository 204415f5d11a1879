"""drop-in alias of multimae_amd.multimae (same public names as the reference's multimae/multimae.py)"""
from multimae_amd.multimae import *  # noqa: F401,F403
from multimae_amd import multimae as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})

# the side effect the training script relies on (run_pretraining_multimae.py:36 "from multimae import multimae" registers the
# factories): make sure they sit in the reference's own utils.registry even if the engine was imported before that package existed
from multimae_amd.registry import sync_reference_registry as _sync
_sync()
