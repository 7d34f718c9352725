"""Stage timings of SF3D's image estimator (CLIP ViT-B/32 tower + heads) on one MI355X."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sculptmate_amd import synth
from sculptmate_amd.sf3d import estimators as est

dev = torch.device("cuda:0")
sd = synth.sf3d_estimator_state(0)
e = est.ClipBasedHeadEstimator(None, "bf16").load_state_dict(sd).to(dev)
rgba = synth.image_rgba(7, 512).astype(np.float32) / 255.0
rgb = torch.from_numpy(rgba[..., :3].copy()).to(dev)
mask = torch.from_numpy(rgba[..., 3].copy()).to(dev)


def t(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


print("encode_image: %.3f ms" % t(lambda: e.encode_image(rgb, mask)))
f = e.encode_image(rgb, mask)[None].clone()
print("heads_forward: %.3f ms" % t(lambda: e.heads_forward(f)))
print("whole call: %.3f ms" % t(lambda: e(rgb[None], mask=mask[None])))
g = est.MultiHeadEstimator(None, "bf16").load_state_dict(sd).to(dev)
tok = torch.randn(3 * 96 * 96, 1024, device=dev)
print("global estimator (96^2 x 3072 -> 24 amplitudes): %.3f ms" % t(lambda: g([tok], 96), n=5))
