"""multimae_amd -- MI355X (gfx950)-native engine for the MultiMAE pre-training hot path.

Public surface = the reference's ``multimae`` package surface for that path
(multimae/__init__.py): criterion, input adapters, MultiMAE / MultiViT + factories,
SpatialOutputAdapter; plus the engine controls (precision, parameter arena, fused optimiser,
data-parallel gradient reducer).
"""
from . import engine  # noqa: F401
from .data_ops import truncated_depth_standardize  # noqa: F401
from .criterion import MaskedCrossEntropyLoss, MaskedL1Loss, MaskedMSELoss  # noqa: F401
from .input_adapters import PatchedInputAdapter, SemSegInputAdapter  # noqa: F401
from .multimae import (MultiMAE, MultiViT, multivit_base, multivit_large,  # noqa: F401
                       pretrain_multimae_base, pretrain_multimae_large)
from .output_adapters import LinearOutputAdapter, SpatialOutputAdapter  # noqa: F401
from .registry import create_model, register_model  # noqa: F401

__version__ = '0.1.0'
