"""drop-in alias of multimae_amd.multimae_utils (same public names as the reference's multimae/multimae_utils.py)"""
from multimae_amd.multimae_utils import *  # noqa: F401,F403
from multimae_amd import multimae_utils as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
