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


VIT_MAP = (("attention.attention.query", "attention.q_proj"), ("attention.attention.key", "attention.k_proj"),
           ("attention.attention.value", "attention.v_proj"), ("attention.output.dense", "attention.o_proj"),
           ("intermediate.dense", "mlp.fc1"), ("output.dense", "mlp.fc2"), ("encoder.layer.", "layers."))


def to_installed_vit_name(k):
    """HF-4.38 checkpoint names (what the reference's model.ckpt uses) -> names of the installed transformers."""
    for a, b in VIT_MAP:
        k = k.replace(a, b)
    return k


def make_tiny():
    """G2: the reference's whole TSR.forward (system.py:82-115) at a tiny configuration."""
    from transformers.models.vit.modeling_vit import ViTConfig, ViTModel
    from sculptmate_amd.tsr.spec import TINY_CFG as C

    v = C["image_tokenizer"]
    tiny_vit = ViTConfig(hidden_size=v["hidden_size"], num_hidden_layers=v["num_hidden_layers"],
                         num_attention_heads=v["num_attention_heads"], intermediate_size=v["intermediate_size"],
                         image_size=v["image_size"], patch_size=v["patch_size"], layer_norm_eps=v["layer_norm_eps"],
                         hidden_act="gelu", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    ViTModel.config_class.from_pretrained = classmethod(lambda cls, *a, **k: tiny_vit)
    from tsr.system import TSR

    b = C["backbone"]
    cfg = {
        "cond_image_size": C["cond_image_size"],
        "image_tokenizer_cls": "x", "image_tokenizer": {},
        "tokenizer_cls": "x", "tokenizer": dict(C["tokenizer"]),
        "backbone_cls": "x", "backbone": dict(in_channels=b["in_channels"], num_attention_heads=b["num_attention_heads"],
                                              attention_head_dim=b["attention_head_dim"], num_layers=b["num_layers"],
                                              cross_attention_dim=b["cross_attention_dim"]),
        "post_processor_cls": "x", "post_processor": dict(C["post_processor"]),
        "decoder_cls": "x", "decoder": dict(C["decoder"]),
        "renderer_cls": "x", "renderer": dict(C["renderer"]),
    }
    model = TSR(cfg).eval()
    sd = synth.tsr_state(seed=21, cfg=C)
    ref_sd = {}
    for k, val in sd.items():
        rk = to_installed_vit_name(k) if k.startswith("image_tokenizer.") else k
        ref_sd[rk] = T(val)
    missing, unexpected = model.load_state_dict(ref_sd, strict=False)
    assert not unexpected and all("image_mean" in m or "image_std" in m for m in missing), (missing, unexpected)
    img = synth.composite_rgb(synth.image_rgba(seed=22, size=C["cond_image_size"]))
    ctx = model.image_tokenizer(T(img).permute(2, 0, 1)[None, None])[0, 0].t().contiguous()  # [T, H]
    codes = model([img], device="cpu")
    np.savez_compressed(os.path.join(HERE, "tsr_tiny.npz"), scene_codes=codes.numpy(), ctx=ctx.numpy(),
                        meta=np.array("tsr_state(seed=21, TINY_CFG); image composite_rgb(image_rgba(22, 64)); "
                                      "transformers %s ('size' pos-embed mode)" % __import__("transformers").__version__))
    print("tiny:", codes.shape, float(codes.abs().max()), ctx.shape)


def make_block():
    """G3: one full-size BasicTransformerBlock(1024, 16, 64, cross 768) on [3072,1024] x [1025,768]
    (basic_transformer_block.py:149-206)."""
    from sculptmate_amd.tsr.spec import make_cfg
    from tsr.models.transformer.basic_transformer_block import BasicTransformerBlock

    cfg = make_cfg(vit_layers=1, layers=1)
    sd = synth.tsr_state(seed=23, cfg=cfg)
    blk = BasicTransformerBlock(1024, 16, 64, cross_attention_dim=768, activation_fn="geglu", attention_bias=False).eval()
    pre = "backbone.transformer_blocks.0."
    blk.load_state_dict({k[len(pre):]: T(v) for k, v in sd.items() if k.startswith(pre)}, strict=True)
    h = np.random.default_rng(24).standard_normal((3072, 1024), dtype=np.float32)
    ctx = np.random.default_rng(25).standard_normal((1025, 768), dtype=np.float32)
    y = blk(T(h)[None], encoder_hidden_states=T(ctx)[None])[0].numpy()
    idx = np.unique(np.random.default_rng(26).integers(0, y.size, 8192)).astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "tsr_block.npz"), idx=idx, y=y.reshape(-1)[idx],
                        ysum=np.array([y.astype(np.float64).sum(), np.abs(y).astype(np.float64).sum()]),
                        meta=np.array("tsr_state(23, make_cfg(vit_layers=1, layers=1)) block 0; h rng 24, ctx rng 25"))
    print("block:", y.shape, float(np.abs(y).max()))


def make_preproc():
    """G1: ImagePreprocessor 1024^2 -> 512^2 antialiased bilinear (utils.py:62-112)."""
    from tsr.utils import ImagePreprocessor

    img = synth.composite_rgb(synth.image_rgba(seed=27, size=1024))
    y = ImagePreprocessor()(img, 512).numpy()
    idx = np.unique(np.random.default_rng(28).integers(0, y.size, 8192)).astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "preproc.npz"), idx=idx, y=y.reshape(-1)[idx], ysum=np.array(y.astype(np.float64).sum()))
    print("preproc:", y.shape)


def uv_test_mesh(seed=0, n=9):
    """A non-overlapping UV layout: jittered (n x n) vertex grid over [0.06, 0.94]^2, two triangles per quad,
    alternating diagonals; plus per-vertex 3-vectors to interpolate."""
    rng = np.random.default_rng(seed)
    g = np.linspace(0.06, 0.94, n)
    u, v = np.meshgrid(g, g, indexing="ij")
    uv = np.stack([u, v], -1).astype(np.float64)
    uv[1:-1, 1:-1] += (rng.random((n - 2, n - 2, 2)) - 0.5) * 0.04
    uv = uv.reshape(-1, 2).astype(np.float32)
    faces = []
    for i in range(n - 1):
        for j in range(n - 1):
            a, b, c, d = i * n + j, (i + 1) * n + j, (i + 1) * n + j + 1, i * n + j + 1
            faces += [[a, b, c], [a, c, d]] if (i + j) % 2 == 0 else [[a, b, d], [b, c, d]]
    faces = np.array(faces, np.int32)
    attr = rng.standard_normal((n * n, 3)).astype(np.float32)
    return uv, faces, attr


def make_baker():
    """SF3D texture baker semantics from the reference's own Python restatement
    (StableFast/sf3d/texture_baker/common.py:123-142 rasterize_cpu, :214-229 interpolate_cpu)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("tb_common", "/root/reference/StableFast/sf3d/texture_baker/common.py")
    tb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tb)
    uv, faces, attr = uv_test_mesh(seed=0, n=9)
    res = 40
    rast = tb.rasterize_cpu(uv, faces, res)
    inter = tb.interpolate_cpu(attr, faces, rast)
    np.savez_compressed(os.path.join(HERE, "baker.npz"), uv=uv, faces=faces, attr=attr, rast=rast.astype(np.float32),
                        inter=inter.astype(np.float32), meta=np.array("common.py rasterize_cpu/interpolate_cpu, res 40, numpy %s" % np.__version__))
    print("baker:", rast.shape, int((rast[..., 3] >= 0).sum()), "covered pixels")


def _sf3d_shims():
    """jaxtyping / gpytoolbox stand-ins so StableFast/sf3d/models/{utils,mesh}.py import (no arithmetic here)."""
    import types

    class _Sub:
        def __class_getitem__(cls, item):
            return cls

    jt = types.ModuleType("jaxtyping")
    for n in ("Float", "Int", "Num", "Integer", "Bool"):
        setattr(jt, n, _Sub)
    sys.modules["jaxtyping"] = jt
    sys.modules["gpytoolbox"] = types.ModuleType("gpytoolbox")
    import PIL.Image  # noqa: F401  (sf3d/models/utils.py uses PIL.Image after a bare `import PIL`)
    if "/root/reference/StableFast" not in sys.path:
        sys.path.insert(0, "/root/reference/StableFast")


def sf3d_tail_inputs():
    rng = np.random.default_rng(33)
    H = W = 48
    yy, xx = np.mgrid[0:H, 0:W]
    mask = ((xx - 20) ** 2 + (yy - 26) ** 2 < 12 ** 2) | ((xx > 30) & (xx < 36) & (yy > 5) & (yy < 40))
    img = rng.random((1, 3, H, W)).astype(np.float32) * mask[None, None]
    uv, faces, _ = uv_test_mesh(seed=2, n=8)
    v_pos = np.concatenate([uv * 2 - 1, (0.3 * np.sin(uv[:, :1] * 6) * np.cos(uv[:, 1:] * 5))], 1).astype(np.float32)
    return img.astype(np.float32), mask[None, None], v_pos, uv, faces.astype(np.int64)


def make_sf3d_tail():
    """dilate_fill (sf3d/models/utils.py:96-133) and Mesh vertex normals / tangents (sf3d/models/mesh.py:66-139)."""
    _sf3d_shims()
    from sf3d.models.mesh import Mesh
    from sf3d.models.utils import dilate_fill

    img, mask, v_pos, uv, faces = sf3d_tail_inputs()
    out = dilate_fill(T(img), T(mask), iterations=10).numpy()
    m = Mesh(T(v_pos), T(faces))
    m._v_tex = T(uv)
    nrm = m._compute_vertex_normal()
    m._v_nrm = nrm
    tng = m._compute_vertex_tangent()
    np.savez_compressed(os.path.join(HERE, "sf3d_tail.npz"), dilate=out, v_nrm=nrm.numpy(), v_tng=tng.numpy())
    print("sf3d_tail:", out.shape, nrm.shape, tng.shape)


if __name__ == "__main__":
    which = sys.argv[1:] or ["query", "grid", "upsample"]
    for w in which:
        globals()["make_" + w]()
