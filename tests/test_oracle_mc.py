"""The marching-cubes oracle (oracle/mc_lewiner.c) against scikit-image goldens + analytic checks."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import capi

G = np.load(os.path.join(GOLDEN, "mc_skimage.npz"))
NAMES = sorted(k[:-4] for k in G.files if k.endswith("_vol"))


@pytest.mark.parametrize("name", NAMES)
def test_bit_exact_vs_skimage_golden(name):
    v, f = capi.marching_cubes(G[name + "_vol"], 0.0)
    assert f.shape == G[name + "_faces"].shape and v.shape == G[name + "_verts"].shape
    assert np.array_equal(f, G[name + "_faces"])
    assert np.array_equal(v.view(np.uint32), G[name + "_verts"].view(np.uint32))


def _edges(faces):
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]])
    return e


def _euler(verts, faces):
    e = np.sort(_edges(faces), axis=1)
    ue, cnt = np.unique(e, axis=0, return_counts=True)
    return len(np.unique(faces)) - len(ue) + len(faces), cnt


@pytest.mark.parametrize("name,chi", [("sphere", 2), ("torus", 0)])
def test_closed_manifold_and_euler_characteristic(name, chi):
    v, f = capi.marching_cubes(G[name + "_vol"], 0.0)
    x, cnt = _euler(v, f)
    assert (cnt == 2).all(), "every edge must be shared by exactly two faces"
    assert x == chi
    # consistent orientation: each directed edge appears once
    de = _edges(f)
    assert len(np.unique(de, axis=0)) == len(de)


def test_sphere_vertices_lie_on_lattice_edges_near_the_surface():
    vol = G["sphere_vol"]
    v, f = capi.marching_cubes(vol, 0.0)
    frac = v - np.floor(v)
    assert ((frac != 0).sum(1) <= 1).all(), "a vertex sits on one lattice edge"
    h = 2.0 / 23
    r = np.linalg.norm(v * h - 1.0, axis=1)
    assert np.abs(r - 0.6).max() < h * h * 2


def test_reference_winding_points_outward():
    """After skimage's descent flip and the reference's [1,0,2] reorder (isosurface.py:52)."""
    v, f = capi.reference_isosurface(-G["sphere_vol"], 24)
    p = v * 2 - 1
    n = np.cross(p[f[:, 1]] - p[f[:, 0]], p[f[:, 2]] - p[f[:, 0]])
    c = p[f].mean(1)
    # density-like field (positive inside): the reference's winding, whichever way it points,
    # must be the same for every face
    s = np.sign((n * c).sum(1))
    assert (s == s[0]).all()
    assert f.dtype == np.int64 and v.dtype == np.float32 and v.min() >= 0 and v.max() <= 1


def test_errors_match_skimage():
    with pytest.raises(ValueError):
        capi.marching_cubes(np.ones((4, 4, 4), np.float32), 0.0)
    with pytest.raises(ValueError):
        capi.marching_cubes(np.ones((1, 4, 4), np.float32), 1.0)
    # level inside the range but no sign change with the strict '> 0' test: all values in {0, -1}
    vol = -np.ones((3, 3, 3), np.float32)
    vol[1, 1, 1] = 0.0
    with pytest.raises(RuntimeError):
        capi.marching_cubes(vol, 0.0)


def test_all_256_sign_patterns_are_consistent_under_value_negation():
    """Triangle count of every corner-sign pattern; a cell and its complement have surfaces."""
    rng = np.random.default_rng(0)
    for idx in range(1, 255):
        signs = np.array([1.0 if idx >> k & 1 else -1.0 for k in range(8)])
        mag = rng.random(8) + 0.25
        vals = signs * mag
        vol = np.empty((2, 2, 2), np.float32)
        order = [(0, 0, 0), (0, 0, 1), (0, 1, 1), (0, 1, 0), (1, 0, 0), (1, 0, 1), (1, 1, 1), (1, 1, 0)]
        for k, (z, y, x) in enumerate(order):
            vol[z, y, x] = vals[k]
        v, f = capi.marching_cubes(vol, 0.0)
        assert 1 <= len(f) <= 12
        assert f.max() == len(v) - 1


CONDA = "/opt/conda/bin/python3.9"


@pytest.mark.skipif(not os.path.exists(CONDA), reason="scikit-image interpreter not present")
def test_live_fuzz_against_skimage(tmp_path):
    """Bit-exact on fresh random volumes / single cells against the real scikit-image (build container)."""
    rng = np.random.default_rng(int.from_bytes(os.urandom(4), "little"))
    vols = {}
    for i in range(12):
        vols["r%d" % i] = rng.standard_normal(tuple(rng.integers(2, 10, 3))).astype(np.float32)
    for i in range(600):
        vols["c%d" % i] = rng.standard_normal((2, 2, 2)).astype(np.float32)
    for i in range(200):
        vols["i%d" % i] = rng.integers(-3, 4, (2, 2, 2)).astype(np.float32)
    np.savez(tmp_path / "in.npz", **vols)
    script = tmp_path / "sk.py"
    script.write_text(
        "import sys, numpy as np, warnings\nwarnings.filterwarnings('ignore')\n"
        "from skimage import measure\nd=np.load(sys.argv[1]); o={}\n"
        "for k in d.files:\n"
        "    try:\n        v,f,_,_=measure.marching_cubes(d[k],0.0); o[k+'_v']=v; o[k+'_f']=f\n"
        "    except Exception as e:\n        o[k+'_e']=np.array(type(e).__name__)\n"
        "np.savez(sys.argv[2], **o)\n")
    subprocess.check_call([CONDA, str(script), str(tmp_path / "in.npz"), str(tmp_path / "out.npz")],
                          cwd=str(tmp_path), stderr=subprocess.DEVNULL)
    o = np.load(tmp_path / "out.npz")
    for k, vol in vols.items():
        if k + "_e" in o.files:
            with pytest.raises((ValueError, RuntimeError)):
                capi.marching_cubes(vol, 0.0)
            continue
        v, f = capi.marching_cubes(vol, 0.0)
        assert np.array_equal(f, o[k + "_f"]), k
        assert np.array_equal(v.view(np.uint32), o[k + "_v"].astype(np.float32).view(np.uint32)), k


def test_ambiguous_sign_patterns_are_the_checkerboard_and_diagonal_ones():
    """The two-pass density grid (csrc/density_filter.hip::filter_cells_kernel) decides with bit logic which cells have all 8 corner
    values read by marching cubes: a face whose diagonal corners agree and whose neighbours differ, or exactly two minority corners
    at the ends of a space diagonal.  Against the case table for all 256 sign patterns: exactly the patterns of cases 3, 4, 6, 7,
    10, 12, 13 -- the ones whose classification runs face / interior tests (oracle/mc_lewiner.c)."""
    from _mcneeds import CORNERS, ambiguous_patterns

    lut = ambiguous_patterns()
    assert int(lut.sum()) == 128
    bit = lambda idx, c: (idx >> CORNERS.index(c)) & 1  # noqa: E731
    faces = [[(0, 0, 0), (0, 1, 0), (0, 1, 1), (0, 0, 1)], [(1, 0, 0), (1, 1, 0), (1, 1, 1), (1, 0, 1)],
             [(0, 0, 0), (1, 0, 0), (1, 0, 1), (0, 0, 1)], [(0, 1, 0), (1, 1, 0), (1, 1, 1), (0, 1, 1)],
             [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0)], [(0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]]
    diagonals = [((0, 0, 0), (1, 1, 1)), ((0, 0, 1), (1, 1, 0)), ((0, 1, 0), (1, 0, 1)), ((0, 1, 1), (1, 0, 0))]
    for idx in range(256):
        g = any(bit(idx, a) == bit(idx, d) and bit(idx, b) == bit(idx, e) and bit(idx, a) != bit(idx, b) for a, b, d, e in faces)
        ones = bin(idx).count("1")
        g = g or any(ones == (2 if s else 6) and bit(idx, p) == s and bit(idx, q) == s for p, q in diagonals for s in (0, 1))
        assert g == bool(lut[idx]), idx

