"""The README usage snippet, runnable: PYTHONPATH=. python tools/usage_check.py (MI355X)."""
import torch, naf_amd
naf = naf_amd.NAF().cuda().eval()
image = torch.randn(1, 3, 1024, 1024, device="cuda")
feats = torch.randn(1, 768, 64, 64, device="cuda", dtype=torch.bfloat16)
up = naf(image, feats, (1024, 1024)); print(up.shape, up.dtype)
replay = naf.capture(image, feats, (1024, 1024)); up2 = replay(); print(torch.equal(up, up2))
loss = naf.forward_train(image, feats.float(), (1024, 1024)).square().mean(); loss.backward(); print(float(loss.detach()))
m = torch.hub.load(".", "naf", source="local", pretrained=False, device="cuda"); print(type(m).__name__)
o, w = naf(image[:, :, :256, :256], feats[:, :, :16, :16], (256, 256), return_weights=True); print(o.shape, w.shape)
