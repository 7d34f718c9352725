#!/usr/bin/env python
"""Generate golden vectors by IMPORTING THE REFERENCE's own hot-path modules (build container only).

    python tests/golden/make_reference_goldens.py [query] [grid] [upsample] [block] [tiny] [preproc]

The reference (/root/reference) cannot travel to the GPU box, so the vectors are committed here as
small .npz fixtures.  Inputs/weights are regenerated from seeds by sculptmate_amd.synth on both
sides; the fixtures hold only what cannot be regenerated (expected outputs, special points).
Stand-ins for omegaconf/bpy/skimage: tests/golden/_reference_shims.py (no arithmetic in them).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import _reference_shims  # noqa: E402

_reference_shims.install()

from sculptmate_amd import synth  # noqa: E402

torch.manual_seed(0)
torch.set_grad_enabled(False)


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def load_np_state(module, sd, prefix):
    own = module.state_dict()
    new = {}
    for k in own:
        new[k] = T(sd[prefix + k]).reshape(own[k].shape)
    module.load_state_dict(new, strict=True)


def make_query():
    """G4: TriplaneNeRFRenderer.query_triplane + NeRFMLP (nerf_renderer.py:41-91)."""
    from tsr.models.nerf_renderer import TriplaneNeRFRenderer
    from tsr.models.network_utils import NeRFMLP

    dec = NeRFMLP({"in_channels": 120, "n_neurons": 64, "n_hidden_layers": 9, "activation": "silu"})
    load_np_state(dec, synth.decoder_state(seed=1), "decoder.")
    ren = TriplaneNeRFRenderer({"radius": 0.87, "feature_reduction": "concat", "density_activation": "exp",
                                "density_bias": -1.0, "num_samples_per_ray": 128})
    ren.set_chunk_size(8192)
    tri = synth.triplane(seed=2, scale=4.0)
    rng = np.random.default_rng(3)
    r = np.float32(0.87)
    pts = (rng.random((12000, 3), dtype=np.float32) * 2 - 1) * r
    # special points: exact borders, corners, texel centres/edges, out-of-range (zero padding)
    sp = []
    for a in (-r, r, 0.0):
        for b in (-r, r, 0.0):
            for c in (-r, r, 0.0):
                sp.append((a, b, c))
    e = np.float32(2 * 0.87 / 64)
    for k in range(-3, 4):
        sp.append((-r + k * e * 0.5, 0.1, r - k * e * 0.25))
        sp.append((0.3, -r + k * e, -0.2))
    for a in (-1.2, 1.05, -0.9, 0.95, 2.0):
        sp.append((a, 0.0, 0.1)); sp.append((0.2, a, -0.3)); sp.append((-0.4, 0.5, a)); sp.append((a, a, a))
    pts = np.concatenate([pts, np.array(sp, np.float32)], 0).astype(np.float32)
    out = ren.query_triplane(dec, T(pts), T(tri))
    np.savez_compressed(os.path.join(HERE, "query_triplane.npz"), pts=pts,
                        **{k: v.numpy() for k, v in out.items()},
                        meta=np.array("decoder_state(seed=1); triplane(seed=2, scale=4.0); radius 0.87"))
    print("query:", pts.shape, {k: float(v.abs().max()) for k, v in out.items()})


def make_grid():
    """G5: MarchingCubeHelper.grid_vertices + the two scale_tensor steps
    (isosurface.py:25-39, system.py:177-181, nerf_renderer.py:52-54)."""
    from tsr.models.isosurface import MarchingCubeHelper
    from tsr.utils import scale_tensor

    out = {}
    rng = np.random.default_rng(5)
    for R in (8, 128, 256):
        g = MarchingCubeHelper(R).grid_vertices
        p = scale_tensor(g, (0, 1), (-0.87, 0.87))
        q = scale_tensor(p, (-0.87, 0.87), (-1, 1))
        n = R ** 3
        idx = np.unique(np.concatenate([np.arange(min(64, n)), np.arange(max(0, n - 64), n),
                                        rng.integers(0, n, 512)])).astype(np.int64)
        out["R%d_idx" % R] = idx
        out["R%d_p" % R] = p[idx].numpy()
        out["R%d_q" % R] = q[idx].numpy()
        out["R%d_psum" % R] = p.double().sum(0).numpy()
        # per-axis coordinate tables (the lattice is separable): p along each axis index
        out["R%d_axis_p" % R] = p.view(R, R, R, 3)[:, 0, 0, 0].numpy()
        out["R%d_axis_q" % R] = q.view(R, R, R, 3)[:, 0, 0, 0].numpy()
    np.savez_compressed(os.path.join(HERE, "grid_vertices.npz"), **out)
    print("grid ok")


def make_upsample():
    """G6: TriplaneUpsampleNetwork (network_utils.py:11-32) on seeded tokens."""
    from tsr.models.network_utils import TriplaneUpsampleNetwork

    up = TriplaneUpsampleNetwork({"in_channels": 1024, "out_channels": 40})
    rng = np.random.default_rng([7, 15])
    w = synth._uniform(rng, (1024, 40, 2, 2), 1.0 / np.sqrt(160.0))
    b = synth._uniform(rng, (40,), 1.0 / np.sqrt(160.0))
    up.load_state_dict({"upsample.weight": T(w), "upsample.bias": T(b)})
    x = np.random.default_rng(8).standard_normal((1, 3, 1024, 32, 32), dtype=np.float32)
    y = up(T(x)).numpy()
    idx = np.unique(np.random.default_rng(9).integers(0, y.size, 8192)).astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "upsample.npz"), idx=idx, y=y.reshape(-1)[idx].astype(np.float32),
                        ysum=y.astype(np.float64).sum((0, 3, 4)),
                        meta=np.array("w,b = _uniform(default_rng([7,15])); x = default_rng(8).standard_normal"))
    print("upsample:", y.shape, float(np.abs(y).max()))


if __name__ == "__main__":
    which = sys.argv[1:] or ["query", "grid", "upsample"]
    for w in which:
        globals()["make_" + w]()
