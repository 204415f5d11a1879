"""drop-in alias of multimae_amd.multimae (same public names as the reference's multimae/multimae.py)"""
from multimae_amd.multimae import *  # noqa: F401,F403
from multimae_amd import multimae as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
