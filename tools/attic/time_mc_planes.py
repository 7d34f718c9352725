"""Marching cubes at 256^3 with the count phase over the whole volume against the count phase from sign planes
(sculpt_mc_count_launch_signed), on a smooth closed surface (few active bricks) and on white noise (every brick active).
    python tools/time_mc_planes.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sculptmate_amd import ops

dev = torch.device("cuda:0")
R = 256
ax = torch.linspace(-1, 1, R, device=dev)
z, y, x = torch.meshgrid(ax, ax, ax, indexing="ij")
fields = {"smooth blob (radius 0.6 + ripples)": 0.6 + 0.05 * torch.sin(9 * x) * torch.sin(7 * y) * torch.sin(8 * z) - torch.sqrt(x * x + y * y + z * z),
          "band-limited noise": torch.nn.functional.interpolate(torch.randn(1, 1, 40, 40, 40, device=dev, generator=torch.Generator(device=dev).manual_seed(0)),
                                                                size=(R, R, R), mode="trilinear", align_corners=True)[0, 0] - 0.8}


def planes_of(v):
    bits = (v > 0).reshape(R * R, R // 32, 32).to(torch.int64)
    w = (bits << torch.arange(32, device=dev)).sum(-1)
    return (w & 0xffffffff).to(torch.int64).to(torch.int32) if False else torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32)


for name, vol in fields.items():
    vol = vol.contiguous()
    pl = planes_of(vol)
    res = {}
    for rnd in range(6):
        for key, kw in (("whole volume", {}), ("sign planes", {"sign_planes": pl})):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            v, f = ops.marching_cubes(vol, 0.0, reference_order=True, vert_div=R - 1.0, **kw)
            e1.record(); torch.cuda.synchronize()
            if rnd:
                res.setdefault(key, []).append(e0.elapsed_time(e1) * 1e3)
            res.setdefault(key + " mesh", (v.clone(), f.clone()))
    same = torch.equal(res["whole volume mesh"][0], res["sign planes mesh"][0]) and torch.equal(res["whole volume mesh"][1], res["sign planes mesh"][1])
    print("%-36s %8d vertices: count over the whole volume %.0f us, from sign planes %.0f us; identical %s"
          % (name, res["sign planes mesh"][0].shape[0], np.median(res["whole volume"]), np.median(res["sign planes"]), same), flush=True)
