"""MI355X-native contrastive training path for multimodal supernova models (gfx950 only).

Host side mirrors the reference's Python surface for this path (LightCurveImageCLIP, its
encoder slots, clip_loss_multimodal, the 9-tuple batch, the state_dict names); all arithmetic
runs in hand-written HIP kernels behind the C-ABI of include/msn_hip.h.
"""
__version__ = "0.1.0"
