"""The C-ABI library loads and exports every symbol include/sculpt_hip.h declares (no GPU needed)."""
import os
import re

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "sculpt_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(sculpt_[a-z0-9_]+)\s*\(", src))
    # texture_baker.dll's and uv_unwrapper.dll's own export names
    names |= set(re.findall(r"\bvoid\s+((?:rasterize|interpolate)_cpu|assign_faces_uv_to_atlas_index)\s*\(", src))
    return sorted(names)


def test_header_declares_something():
    names = _declared()
    assert "sculpt_density_grid" in names and "sculpt_mc_emit" in names and len(names) >= 15


def test_library_exports_every_declared_symbol():
    from sculptmate_amd import _lib

    for name in _declared():
        assert hasattr(_lib.lib, name), "libsculpt_hip.so does not export %s" % name
        assert name in _lib.SIGNATURES, "no ctypes signature for %s" % name
    assert sorted(_lib.SIGNATURES) == _declared()


def test_version_and_error_string():
    from sculptmate_amd import _lib

    import re

    hdr = open(os.path.join(ROOT, "include", "sculpt_hip.h")).read()
    want = int(re.search(r"#define\s+SCULPT_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert want == 4 and _lib.lib.sculpt_version() == want   # 4: the two-pass grid reports 12 statistics words (round 6)
    assert isinstance(_lib.last_error(), str)
    assert _lib.lib.sculpt_device_count() >= 0


def test_library_was_built_from_the_sources_beside_it():
    """sculpt_source_digest(): the library carries the sha256 of the HIP sources, headers and flags it was compiled from;
    `_lib` refuses (or rebuilds) a library that does not match, so a stale kernel cannot be what the tests measured."""
    from sculptmate_amd import _lib, build

    have = _lib.lib.sculpt_source_digest().decode()
    assert re.fullmatch(r"[0-9a-f]{32}", have), have
    assert have == build.source_digest() == build.built_digest()
    assert build.is_fresh()
    # the digest moves with any source byte and with the flags
    assert build.source_digest("other flags") != have


def test_no_cpu_fallback_for_cpu_tensors():
    import pytest
    import torch

    from sculptmate_amd import ops

    with pytest.raises(ops.SculptError):
        ops.marching_cubes(torch.zeros(4, 4, 4))


def test_product_never_imports_oracle():
    """The product path must not route through the oracle (test infrastructure only)."""
    pkg = os.path.join(ROOT, "sculptmate_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "oracle/" not in txt or f == "mc_luts.h", f


def test_host_mesh_entry_points_refuse_bad_arguments():
    """sculpt_mesh_* take HOST pointers and run without a GPU: null / oversized / inconsistent arguments come back as an error
    code with a message, never a crash; a released or null handle is harmless."""
    import ctypes

    import numpy as np

    from sculptmate_amd import _lib

    lib = _lib.lib
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float64)
    f = np.array([[0, 2, 1], [0, 1, 3], [1, 2, 3], [2, 0, 3]], np.int32)
    out = ctypes.c_void_p()
    assert lib.sculpt_mesh_decimate(v.ctypes.data, 4, f.ctypes.data, 4, 0, None) != 0 and "null result" in _lib.last_error()
    assert lib.sculpt_mesh_decimate(None, 4, f.ctypes.data, 4, 0, ctypes.byref(out)) != 0 and out.value is None
    assert lib.sculpt_mesh_subdivide(v.ctypes.data, 4, f.ctypes.data, 4, 13, ctypes.byref(out)) != 0      # iters out of range
    assert lib.sculpt_mesh_subdivide(v.ctypes.data, 4, f.ctypes.data, 4, -1, ctypes.byref(out)) != 0
    assert lib.sculpt_mesh_remesh_botsch(v.ctypes.data, 4, f.ctypes.data, 4, 5, float("nan"), 1, ctypes.byref(out)) != 0
    assert lib.sculpt_mesh_remesh_botsch(v.ctypes.data, 4, f.ctypes.data, 4, -3, -1.0, 1, ctypes.byref(out)) != 0
    assert lib.sculpt_mesh_read(None, v.ctypes.data, f.ctypes.data) != 0
    assert lib.sculpt_mesh_num_vertices(None) == 0 and lib.sculpt_mesh_num_faces(None) == 0
    lib.sculpt_mesh_free(None)
    # and the good call still works afterwards: a tetrahedron cannot be decimated
    assert lib.sculpt_mesh_decimate(v.ctypes.data, 4, f.ctypes.data, 4, 0, ctypes.byref(out)) == 0
    assert lib.sculpt_mesh_num_vertices(out) == 4 and lib.sculpt_mesh_num_faces(out) == 4
    v2, f2 = np.empty((4, 3)), np.empty((4, 3), np.int32)
    assert lib.sculpt_mesh_read(out, v2.ctypes.data, f2.ctypes.data) == 0
    lib.sculpt_mesh_free(out)
    assert np.array_equal(v2, v) and np.array_equal(f2, f)


def test_graft_entry_build_passes_on_the_built_tree():
    """The driver's `build()` check: compiles (a no-op on a fresh tree), builds the C oracle and verifies the ABI version the
    header declares -- a hard-coded version here once outlived an ABI bump."""
    import importlib
    import sys

    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    g = importlib.import_module("__graft_entry__")
    g.build()
