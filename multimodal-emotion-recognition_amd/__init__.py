"""MI355X-native M2FNet fusion-transformer training path (gfx950 HIP kernels behind a C ABI).

Import as ``mer_amd`` (see ``/mer_amd.py`` at the repo root).  Sub-modules:
  layout   - model config + flat parameter layout (reference state_dict order)
  runtime  - ctypes binding of ``csrc/libm2fnet_hip.so`` (include/m2fnet_hip.h); fails loudly if absent
  model    - ``M2FNet`` / ``FusionAttentionModule`` nn.Module mirrors driving the HIP plan
  dp       - dialogue-sharded data parallelism (RCCL all-reduce of the flat gradient buffer)
"""
from . import layout  # noqa: F401

__all__ = ["layout"]
