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

    assert _lib.lib.sculpt_version() == 1
    assert isinstance(_lib.last_error(), str)
    assert _lib.lib.sculpt_device_count() >= 0


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
