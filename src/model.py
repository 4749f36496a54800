"""Drop-in for the reference's ``src/model.py``: same names (``M2FNet``, ``FusionAttentionModule``), same
constructor / forward signatures and state_dict keys; the arithmetic runs in the gfx950 HIP kernels of
``multimodal-emotion-recognition_amd`` (there is no CPU path)."""
import os
import sys

_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

import mer_amd  # noqa: E402,F401
from mer_amd.model import M2FNet, FusionAttentionModule  # noqa: E402,F401

__all__ = ["M2FNet", "FusionAttentionModule"]
