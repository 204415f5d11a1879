"""drop-in alias of multimae_amd.output_adapters (same public names as the reference's multimae/output_adapters.py)"""
from multimae_amd.output_adapters import *  # noqa: F401,F403
from multimae_amd import output_adapters as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
