"""Golden scene summaries for the two Blender mesh sinks, produced by RUNNING THE REFERENCE'S OWN FUNCTIONS against the
recording stand-in for `bpy` (tests/fake_bpy.py):

    /root/reference/TripoSR/tsr/system.py:127-168        TSR.import_obj_blender(verts, faces, vertex_colors, name)
    /root/reference/StableFast/sf3d/system.py:530-598    SF3D.import_mesh_blender(mesh dict, mesh_name)

Run in the build container only (needs /root/reference):  python tests/golden/make_blender_goldens.py
Output: tests/golden/blender_sink.npz = the seeded inputs + fake_bpy.summary() of what each reference call left in the
scene (mesh arrays, per-loop colours / UVs, node graphs, images).  tests/test_blender_sinks.py feeds the same inputs to
sculptmate_amd's sinks and compares.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import fake_bpy  # noqa: E402
import _reference_shims as shims  # noqa: E402

shims.install()            # omegaconf / skimage stand-ins (and an empty bpy, replaced per case below)
import make_reference_goldens as mrg  # noqa: E402

mrg._sf3d_shims()
import transformers.pytorch_utils as _pu  # noqa: E402

for _n in ("find_pruneable_heads_and_indices", "prune_linear_layer"):
    if not hasattr(_pu, _n):
        setattr(_pu, _n, lambda *a, **k: None)


def inputs():
    """A small closed mesh (octahedron subdivided once: 18 vertices, 32 triangles), seeded colours, UVs and textures."""
    v = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float64)
    f = [[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]]
    verts, faces, mid = [tuple(p) for p in v], [], {}

    def m(a, b):
        k = (min(a, b), max(a, b))
        if k not in mid:
            p = (np.array(verts[a]) + np.array(verts[b])) / 2
            verts.append(tuple(p / np.linalg.norm(p)))
            mid[k] = len(verts) - 1
        return mid[k]

    for a, b, c in f:
        ab, bc, ca = m(a, b), m(b, c), m(c, a)
        faces += [[a, ab, ca], [ab, b, bc], [ca, bc, c], [ab, bc, ca]]
    rng = np.random.default_rng(77)
    verts = (np.array(verts) * 0.6 + rng.normal(0, 0.01, (len(verts), 3))).astype(np.float32)
    faces = np.array(faces, np.int64)
    colors = rng.random((len(verts), 3)).astype(np.float32)
    uvs = rng.random((len(verts), 2)).astype(np.float32)
    base = rng.integers(0, 256, (6, 5, 4), dtype=np.uint8)   # height 6, width 5: the vertical flip and the w/h order matter
    bump = rng.integers(0, 256, (6, 5, 4), dtype=np.uint8)
    return dict(verts=verts, faces=faces, colors=colors, uvs=uvs, base=base, bump=bump)


def main():
    from PIL import Image

    out = {"in." + k: v for k, v in inputs().items()}
    I = inputs()

    # ---- TripoSR: TSR.import_obj_blender, with and without vertex colours
    bpy = fake_bpy.install()
    for mod in [k for k in sys.modules if k == "tsr" or k.startswith("tsr.")]:
        del sys.modules[mod]
    from tsr.system import TSR
    import tsr.system as tsr_system

    tsr_system.bpy = bpy
    TSR.import_obj_blender(None, I["verts"], I["faces"], I["colors"], name="Chair")
    TSR.import_obj_blender(None, I["verts"], I["faces"], None, name="Plain")
    out.update({"tsr." + k: v for k, v in fake_bpy.summary(bpy).items()})

    # ---- StableFast: SF3D.import_mesh_blender, textured and untextured
    bpy = fake_bpy.install()
    import types

    # sf3d/system.py imports its estimators at module level; open_clip / torchvision are absent here and the sink never
    # touches them: empty stand-in modules, import only
    oc = types.ModuleType("open_clip")
    oc.constants = types.ModuleType("open_clip.constants")
    oc.constants.OPENAI_DATASET_MEAN, oc.constants.OPENAI_DATASET_STD = (0.0,) * 3, (1.0,) * 3
    tv = types.ModuleType("torchvision")
    tv.transforms = types.ModuleType("torchvision.transforms")
    tv.transforms.Normalize = object
    for name, mod in (("open_clip", oc), ("open_clip.constants", oc.constants), ("torchvision", tv),
                      ("torchvision.transforms", tv.transforms)):
        sys.modules.setdefault(name, mod)
    import sf3d.system as sf3d_system

    sf3d_system.bpy = bpy
    tex = dict(vertices=I["verts"], faces=I["faces"], uvs=I["uvs"], basecolor_tex=Image.fromarray(I["base"], "RGBA"),
               bump_tex=Image.fromarray(I["bump"], "RGBA"), roughness=0.625, metallic=0.25)
    sf3d_system.SF3D.import_mesh_blender(None, tex, "Lamp")
    bare = dict(vertices=I["verts"], faces=I["faces"], uvs=I["uvs"], basecolor_tex=None, bump_tex=None, roughness=None,
                metallic=None)
    sf3d_system.SF3D.import_mesh_blender(None, bare, "Bare")
    out.update({"sf3d." + k: v for k, v in fake_bpy.summary(bpy).items()})

    np.savez_compressed(os.path.join(HERE, "blender_sink.npz"), **out)
    for k, v in out.items():
        print(k, v.shape if v.ndim else str(v)[:300])


if __name__ == "__main__":
    main()
