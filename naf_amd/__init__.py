"""naf_amd -- MI355X-native (gfx950 / CDNA4) implementation of NAF's cross-scale neighbourhood
attention filtering forward, behind the reference's own API (valeoai/NAF: src/model/naf.py).

    from naf_amd import NAF            # same constructor / state_dict as the reference
    naf = NAF().to("cuda").eval()
    hr = naf(image, lr_features, (H, W))

Kernels live in naf_amd/csrc (HIP, C ABI in include/naf_hip.h); build with ``python -m naf_amd.build``.
"""
from .model import NAF, CrossAttention, GraphedForward, ImageEncoder, RoPE  # noqa: F401

__all__ = ["NAF", "CrossAttention", "GraphedForward", "ImageEncoder", "RoPE"]
__version__ = "0.1.0"
