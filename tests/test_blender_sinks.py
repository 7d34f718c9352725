"""The two Blender mesh sinks (the reference's actual output boundary) against golden scene summaries produced by the
REFERENCE's own import_obj_blender / import_mesh_blender on a recording stand-in for bpy
(tests/fake_bpy.py, tests/golden/make_blender_goldens.py).  Runs on the CPU: the sinks are host code."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN

import fake_bpy


@pytest.fixture()
def golden():
    return np.load(os.path.join(GOLDEN, "blender_sink.npz"))


@pytest.fixture()
def bpy():
    keep = sys.modules.get("bpy")
    mod = fake_bpy.install()
    yield mod
    if keep is None:
        sys.modules.pop("bpy", None)
    else:
        sys.modules["bpy"] = keep


def _compare(got, golden, prefix):
    keys = sorted(k[len(prefix):] for k in golden.files if k.startswith(prefix))
    assert sorted(got) == keys, (sorted(got), keys)
    for k in keys:
        want = golden[prefix + k]
        if k == "meta":
            assert json.loads(str(got[k])) == json.loads(str(want))
        elif want.dtype.kind == "f":
            # colours, UVs and pixels are stored by Blender as float32; the reference builds them in float64
            np.testing.assert_allclose(got[k], want, rtol=0, atol=1e-7, err_msg=k)
        else:
            assert np.array_equal(got[k], want), k


def test_tsr_import_obj_blender_matches_reference(golden, bpy):
    from sculptmate_amd.tsr.blender_sink import import_obj_blender

    v, f, c = golden["in.verts"], golden["in.faces"], golden["in.colors"]
    import_obj_blender(v, f, c, name="Chair")
    import_obj_blender(v, f, None, name="Plain")
    _compare(fake_bpy.summary(bpy), golden, "tsr.")


def test_tsr_default_sink_is_the_blender_sink_when_bpy_is_importable(bpy):
    from sculptmate_amd.tsr import system

    sink = system._default_sink()
    assert sink is not None and sink.__name__ == "import_obj_blender"


def test_sf3d_import_mesh_blender_matches_reference(golden, bpy):
    from PIL import Image

    from sculptmate_amd.sf3d.blender_sink import import_mesh_blender

    v, f, uv = golden["in.verts"], golden["in.faces"], golden["in.uvs"]
    tex = dict(vertices=v, faces=f, uvs=uv, basecolor_tex=Image.fromarray(golden["in.base"], "RGBA"),
               bump_tex=Image.fromarray(golden["in.bump"], "RGBA"), roughness=0.625, metallic=0.25)
    import_mesh_blender(tex, "Lamp")
    bare = dict(vertices=v, faces=f, uvs=uv, basecolor_tex=None, bump_tex=None, roughness=None, metallic=None)
    import_mesh_blender(bare, "Bare")
    _compare(fake_bpy.summary(bpy), golden, "sf3d.")
