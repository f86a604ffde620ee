import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import naf_oracle as O
import test_gpu_parity as T
dev = torch.device("cuda:0")
p = O.make_params(seed=41)
m = T._load_model(dev, p, kernel_size=5)
img = O.hash_normal((2, 3, 96, 128), 961).to(dev)
ft = O.hash_normal((2, 128, 6, 8), 962).to(dev).to(torch.bfloat16)
a = m(img, ft, (96, 128)); a2 = m(img, ft, (96, 128))
m.single_call = False
b = m(img, ft, (96, 128)); b2 = m(img, ft, (96, 128))
print("single vs single", torch.equal(a, a2), "composed vs composed", torch.equal(b, b2), "single vs composed", torch.equal(a, b),
      "differing", int((a != b).sum()), "of", a.numel(), "max", float((a.float() - b.float()).abs().max()))
enc = m.image_encoder
g1 = enc._stem_hip(img)
print("stem twice equal", torch.equal(g1, enc._stem_hip(img)))
