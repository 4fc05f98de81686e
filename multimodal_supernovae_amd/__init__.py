"""MI355X-native contrastive training path for multimodal supernova models (gfx950 only).

Host side mirrors the reference's Python surface for this path (LightCurveImageCLIP, its
encoder slots, clip_loss_multimodal, the 9-tuple batch, the state_dict names); all arithmetic
runs in hand-written HIP kernels behind the C-ABI of include/msn_hip.h.
"""
__version__ = "0.1.0"

import os as _os

# RCCL needs dmabuf IPC on this pool's host driver; the variable has to be in the environment before the HIP runtime
# starts (see distributed.py), so the package sets it at import, whichever module a user reaches first.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
