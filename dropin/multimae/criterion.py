"""drop-in alias of multimae_amd.criterion (same public names as the reference's multimae/criterion.py)"""
from multimae_amd.criterion import *  # noqa: F401,F403
from multimae_amd import criterion as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
