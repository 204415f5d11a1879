"""Drop-in package: put this directory ahead of the reference's on sys.path and the unmodified
``run_pretraining_multimae.py`` imports the MI355X-native engine:

    from multimae import multimae                     # registers pretrain_multimae_* / multivit_* (also in utils.registry)
    from multimae.criterion import MaskedCrossEntropyLoss, MaskedL1Loss, MaskedMSELoss
    from multimae.input_adapters import PatchedInputAdapter, SemSegInputAdapter
    from multimae.output_adapters import SpatialOutputAdapter

(reference import surface: run_pretraining_multimae.py:36-40, multimae/__init__.py)."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _root not in sys.path:
    sys.path.insert(0, _root)

from multimae_amd.criterion import MaskedCrossEntropyLoss, MaskedL1Loss, MaskedMSELoss  # noqa: E402,F401
from multimae_amd.input_adapters import PatchedInputAdapter, SemSegInputAdapter  # noqa: E402,F401
from multimae_amd.multimae import MultiMAE, MultiViT  # noqa: E402,F401
from multimae_amd.output_adapters import SpatialOutputAdapter  # noqa: E402,F401
