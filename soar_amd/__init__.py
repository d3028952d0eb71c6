"""soar_amd -- MI355X-native (gfx950) implementation of SOAR's per-frame avatar training path:
SMPL-X LBS warp of canonical Gaussian surfels -> Gaussian-surfel rasterizer forward -> backward.

Host code is Python on PyTorch-ROCm; all compute runs in hand-written HIP behind the C-ABI
library declared in include/soar_hip.h (built by ``soar_amd.build``).
"""
__version__ = "0.1.0"
