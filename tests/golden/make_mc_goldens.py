#!/usr/bin/env python
"""Golden vectors for marching cubes from the REAL scikit-image (the reference's third-party
dependency, called at /root/reference/TripoSR/tsr/models/isosurface.py:46-48).

Run in the build container with the interpreter that has scikit-image installed:

    /opt/conda/bin/python3.9 tests/golden/make_mc_goldens.py

(scikit-image 0.18.3 there; the reference leaves the version unpinned.)  Writes
tests/golden/mc_skimage.npz holding input volumes AND skimage.measure.marching_cubes(vol, 0.0)
outputs (verts float32, faces int32), plus the reference's own post-processing of them
(MarchingCubeHelper.forward: faces[:, [1,0,2]], verts / (R-1)).
"""
import os
import warnings

import numpy as np

warnings.filterwarnings("ignore")
from skimage import measure  # noqa: E402
import skimage  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def volumes():
    rng = np.random.default_rng(20240531)
    vols = {}
    # white noise: exercises every ambiguous Lewiner case (3,4,6,7,10,12,13) and the centre vertex
    vols["noise_a"] = rng.standard_normal((12, 10, 14)).astype(np.float32)
    vols["noise_b"] = (rng.random((9, 16, 11)) - 0.5).astype(np.float32)
    # integer-valued: exact zeros at corners and exactly-degenerate saddles (A*C == B*D)
    vols["ints"] = rng.integers(-2, 3, (8, 9, 7)).astype(np.float32)
    # analytic fields
    g = np.linspace(-1, 1, 24)
    x, y, z = np.meshgrid(g, g, g, indexing="ij")
    vols["sphere"] = (0.6 - np.sqrt(x * x + y * y + z * z)).astype(np.float32)
    vols["torus"] = (0.25 - np.sqrt((np.sqrt(x * x + y * y) - 0.55) ** 2 + z * z)).astype(np.float32)
    g2 = np.linspace(0, 2 * np.pi, 20)
    x, y, z = np.meshgrid(g2, g2, g2, indexing="ij")
    vols["gyroid"] = (np.sin(x) * np.cos(y) + np.sin(y) * np.cos(z) + np.sin(z) * np.cos(x)).astype(np.float32)
    # density-like field as the reference feeds it: density_act - 25 (system.py:184, isosurface.py:45)
    g3 = np.linspace(-0.87, 0.87, 20)
    x, y, z = np.meshgrid(g3, g3, g3, indexing="ij")
    dens = 25.0 * np.exp(9.0 * (0.5 - np.sqrt(x * x + 1.3 * y * y + 0.8 * z * z)))
    vols["density"] = (dens.astype(np.float32) - np.float32(25.0)).astype(np.float32)
    # surface touching the volume border (open surface), anisotropic shape
    vols["slab"] = (rng.standard_normal((5, 6, 17)) * 0.3 + np.linspace(-1, 1, 17)[None, None, :]).astype(np.float32)
    vols["tiny"] = np.array([[[1, -1], [-1, 1]], [[-1, 1], [1, -2]]], np.float32)
    return vols


def main():
    out = {"skimage_version": np.array(skimage.__version__)}
    for k, v in volumes().items():
        verts, faces, _, _ = measure.marching_cubes(v, 0.0)
        out[k + "_vol"] = v
        out[k + "_verts"] = verts.astype(np.float32)
        out[k + "_faces"] = faces.astype(np.int32)
        print(k, v.shape, verts.shape, faces.shape)
    np.savez_compressed(os.path.join(HERE, "mc_skimage.npz"), **out)
    # error behaviour
    for name, vol in (("all_positive", np.ones((4, 4, 4), np.float32)),):
        try:
            measure.marching_cubes(vol, 0.0)
            print(name, "no error")
        except Exception as e:
            print(name, type(e).__name__, e)


if __name__ == "__main__":
    main()
