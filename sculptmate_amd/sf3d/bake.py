"""Texture bake of SF3D.generate_mesh (/root/reference/StableFast/sf3d/system.py:358-486) on the GPU.

  rasterize UVs -> barycentric map          TextureBaker.rasterize   (sculpt_bake_rasterize)
  interpolate position / normal / tangent   TextureBaker.interpolate (sculpt_bake_interpolate)
  MaterialMLP(features, perturb_normal) at the texel positions       (fused sample+MLP kernel)
  albedo + tangent-space bump per texel     system.py:375-440        (sculpt_bake_material)
  uv_padding = dilate_fill(iterations = res // 150)                  (sculpt_dilate_fill)
  float32 -> uint8 with dither, PIL images  models/utils.py:136-151, system.py:455-481 (host, as in the reference)

The reference gathers the covered texels (`pos_bake[bake_mask]`), runs the decoder on those, and scatters back; here
the decoder runs on every texel (uncovered ones sample a dummy position and are masked to zero in the composition
kernel) -- at most res^2 = 1 M points, cheaper than a compaction round trip.
"""
import numpy as np
import torch

from .. import ops


def float32_to_uint8_np(x, dither=True, dither_mask=None, dither_strength=1.0, rng=None):
    """models/utils.py:136-151 (np.random in the reference; a Generator may be passed for reproducible tests)."""
    if dither:
        r = (rng.random(x[..., :1].shape) if rng is not None else np.random.rand(*x[..., :1].shape)).astype(np.float32)
        d = dither_strength * r - 0.5
        if dither_mask is not None:
            d = d * dither_mask
        return np.clip(np.floor(256.0 * x + d), 0, 255).astype(np.uint8)
    return np.clip(np.floor(256.0 * x), 0, 255).astype(np.uint8)


def bake_maps(model, mesh, scene_code, bake_resolution: int, want_bump=True):
    """-> dict(albedo, bump, mask) of padded fp32 [res,res,3] device images (before uint8 conversion)."""
    res = int(bake_resolution)
    faces = mesh.t_pos_idx
    rast = ops.bake_rasterize(mesh.v_tex, faces, res)
    pos = ops.bake_interpolate(mesh.v_pos, rast, faces)
    q = model.query_triplane(pos.reshape(-1, 3), scene_code)
    dec = model.decoder(q, exclude=["density", "vertex_offset"])
    color = dec["features"][0]
    pn = nrm = tng = None
    if want_bump and "perturb_normal" in dec:
        pn = dec["perturb_normal"][0]
        nrm = ops.bake_interpolate(mesh.v_nrm, rast, faces)
        tng = ops.bake_interpolate(mesh.v_tng, rast, faces)  # un-normalised tangents, as the reference notes (:401-403)
    albedo, bump = ops.bake_material(rast, color, pn, nrm, tng)
    mask = (rast[..., -1] >= 0)
    it = res // 150

    def uv_padding(img):
        x = img.permute(2, 0, 1)[None].contiguous()
        return ops.dilate_fill(x, mask[None, None], iterations=it)[0].permute(1, 2, 0).contiguous()

    out = {"albedo": uv_padding(albedo), "bump": uv_padding(bump) if bump is not None else None, "mask": mask}
    return out


def bake_textures(model, mesh, scene_code, bake_resolution: int, global_dict, index: int, rng=None):
    """The texture half of generate_mesh's per-mesh dict (system.py:443-494)."""
    from PIL import Image

    maps = bake_maps(model, mesh, scene_code, bake_resolution)
    basecolor = Image.fromarray(float32_to_uint8_np(maps["albedo"].cpu().numpy(), rng=rng)).convert("RGBA")
    basecolor.format = "JPEG"
    bump_tex = None
    if maps["bump"] is not None:
        bump_np = maps["bump"].cpu().numpy()
        bump_up = np.ones_like(bump_np)
        bump_up[..., :2] = 0.5
        bump_up[..., 2:] = 1
        flat = np.all(bump_np == bump_up, axis=-1, keepdims=True).astype(np.float32)
        bump_tex = Image.fromarray(float32_to_uint8_np(bump_np, dither=True, dither_mask=flat, rng=rng)).convert("RGBA")
        bump_tex.format = "JPEG"

    def scalar(name):
        v = global_dict.get("decoder_" + name)
        return None if v is None else float(torch.as_tensor(v[index]).squeeze().cpu().item())

    return dict(basecolor_tex=basecolor, bump_tex=bump_tex, roughness=scalar("roughness"), metallic=scalar("metallic"))


def cell_atlas_unwrapper(v_pos, v_nrm, faces, island_padding=0.02):
    """Unwrapper-compatible stand-in (see ops.uv_cell_atlas): one cell per triangle."""
    return ops.uv_cell_atlas(v_pos, faces, padding=max(island_padding, 0.05))
