"""torch.hub entry point -- same surface as the reference's hubconf.py:8-24.

    naf = torch.hub.load("<this repo>", "naf", pretrained=False, device="cuda", source="local")
    hr_features = naf(image, lr_features, (H, W))
"""
dependencies = ["torch"]

import torch

from naf_amd import NAF

CHECKPOINT_URL = "https://github.com/valeoai/NAF/releases/download/model/naf_release.pth"


def naf(pretrained: bool = True, device="cpu"):
    """NAF upsampler on MI355X-native HIP kernels -- the FORWARD NEEDS A ROCm DEVICE: with the reference's default
    ``device="cpu"`` the module is built and loaded, but calling it raises RuntimeError until it is moved (``.to("cuda")``);
    there is no CPU path in this library (pass ``device="cuda"``).

    Builds the default model (dim 256, 4 heads, kernel 9) and, if ``pretrained``, loads the reference's
    released weights (identical ``state_dict`` keys, strict).  The model can be constructed and
    loaded on any device; its forward needs a ROCm device.  Like the reference the module comes back in training
    mode: call ``.eval()`` (README usage) or run under ``torch.no_grad()`` for the fused inference path.
    """
    model = NAF().to(device)
    if pretrained:
        state = torch.hub.load_state_dict_from_url(CHECKPOINT_URL, progress=True, map_location=device)
        model.load_state_dict(state)
    return model          # like the reference (hubconf.py:20-24): the caller decides on .eval(), README usage calls it
