"""SF3D -- the StableFast-3D system behind SculptMate's "fast" generator, MI355X-native (BASELINE config 4).

Mirrors the surface of /root/reference/StableFast/sf3d/system.py::SF3D that StableFast/generate.py uses:
from_pretrained, to(device), eval(), run_image, generate_mesh, get_scene_codes, triplane_to_meshes, query_triplane,
decoder(values, include=/exclude=), import_mesh_blender.  All arithmetic of the networks and of marching tetrahedra
runs in libsculpt_hip.so; torch owns HBM buffers and streams.

Pipeline per image (reference file:line):
  prepare_image (RGBA -> grey composite)                  sf3d/system.py:285-305
  LinearCameraEmbedder on the fixed default camera        sf3d/models/camera.py:21-32, sf3d/utils.py:24-50
  DINOv2-L with adaLN modulation, 1297 tokens             sf3d/models/tokenizers/image.py:64-96, dinov2.py:468-546
  TriplaneLearnablePositionalEmbedding [1024, 27648]      sf3d/models/tokenizers/triplane.py:29-49
  TwoStreamInterleaveTransformer, 4 x (fuse, 3 basic, fuse)  sf3d/models/transformers/backbone.py:398-515
  PixelShuffleUpsampleNetwork -> scene code [3,40,384,384]   sf3d/models/network.py:29-75
  triplane_to_meshes: MaterialMLP(density, vertex_offset) at the tet-grid vertices -> marching tetrahedra
                                                          sf3d/system.py:140-168, models/isosurface.py:108-229
  texture bake (rasterize, interpolate, MaterialMLP(features, perturb_normal), bump, dilate)  sf3d/system.py:358-486

Input-independent work is folded once at load: the camera is the constant default camera, so every adaLN
modulation (scale, shift) is a constant and folds into its LayerNorm's affine; LayerScale folds into the preceding
Linear; GroupNorm+proj of the learned triplane tokens and LayerNorm+proj of the learned latents are constants.
"""
import math
import os
from typing import List, Optional

import numpy as np
import torch

from .. import _lib, ops
from ..engine import KernelEngine, prepare_ln_linear
from ..tsr.posemb import interpolate_pos_embedding
from .spec import DEFAULT_CFG, IGNORED_PREFIXES, param_spec
from .tets import load_tets

BF16 = torch.bfloat16
IMAGE_MEAN = (0.485, 0.456, 0.406)  # sf3d/models/tokenizers/image.py:52-61
IMAGE_STD = (0.229, 0.224, 0.225)


def _bf(x, dev):
    return torch.as_tensor(x).to(device=dev, dtype=BF16).contiguous()


def _f32(x, dev):
    return torch.as_tensor(x).to(device=dev, dtype=torch.float32).contiguous()


def default_cond_c2w(distance: float) -> np.ndarray:
    """sf3d/utils.py:41-50"""
    return np.array([[0, 0, 1, distance], [1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1]], np.float32)


def create_intrinsic_from_fov_deg(fov_deg: float, cond_height: int, cond_width: int):
    """sf3d/utils.py:24-38 + models/utils.py:223-236"""
    focal = 0.5 * cond_height / np.tan(0.5 * np.deg2rad(fov_deg))
    K = np.identity(3, dtype=np.float32)
    K[0, 0] = focal
    K[1, 1] = focal
    K[0, 2] = cond_width / 2.0
    K[1, 2] = cond_height / 2.0
    Kn = K.copy()
    Kn[0, 2] /= cond_width
    Kn[1, 2] /= cond_height
    Kn[0, 0] /= cond_width
    Kn[1, 1] /= cond_height
    return K, Kn


class TriplaneQuery:
    """What query_triplane returns here: the (positions, triplane) pair.  The reference materialises the
    [N, 120] sampled features (system.py:186-199) and hands them to the decoder; the HIP kernel samples inside the
    decoder launch, so the features never exist in HBM."""

    def __init__(self, positions, triplane, radius):
        self.positions, self.triplane, self.radius = positions, triplane, radius


class MaterialMLP:
    """sf3d/models/network.py:148-210 -- one fused sample+MLP launch per requested head."""

    def __init__(self, cfg, sd, dev):
        self.cfg = cfg
        self.heads = {}
        for h in cfg["heads"]:
            key = "decoder.heads.%s." % h["name"]
            n = h["n_hidden_layers"]
            Ws = [np.asarray(sd[key + "%d.weight" % (2 * i)], np.float32) for i in range(n + 1)]
            bs = [np.asarray(sd[key + "%d.bias" % (2 * i)], np.float32) for i in range(n + 1)]
            # the kernel's output layer has 4 rows (density | 3 features): a 1-channel head lives in row 0,
            # a 3-channel head in rows 1..3
            co = h["out_channels"]
            if co not in (1, 3):
                raise _lib.SculptError("MaterialMLP head %s: %d output channels unsupported" % (h["name"], co))
            W4 = np.zeros((4, Ws[-1].shape[1]), np.float32)
            b4 = np.zeros(4, np.float32)
            r0 = 0 if co == 1 else 1
            W4[r0:r0 + co] = Ws[-1]
            b4[r0:r0 + co] = bs[-1]
            self.heads[h["name"]] = (h, ops.PackedMLP(Ws[:-1] + [W4], bs[:-1] + [b4], dev))

    def keys(self):
        return self.heads.keys()

    def __call__(self, x: TriplaneQuery, include: Optional[List] = None, exclude: Optional[List] = None):
        if include is not None and exclude is not None:
            raise ValueError("Cannot specify both include and exclude.")
        out = {}
        for name, (h, mlp) in self.heads.items():
            if include is not None and name not in include:
                continue
            if exclude is not None and name in exclude:
                continue
            act = h["output_activation"]
            one = h["out_channels"] == 1
            if act == "trunc_exp":
                want = "density_act" if one else None
            elif act == "sigmoid":
                want = None if one else "color"
            else:
                want = "density" if one else "features"
            if want is None:
                raise _lib.SculptError("MaterialMLP head %s: activation %s on %d channels unsupported"
                                       % (name, act, h["out_channels"]))
            bias = float(h["out_bias"])
            if want in ("features", "color", "density") and bias != 0.0:
                raise _lib.SculptError("MaterialMLP head %s: out_bias only supported with trunc_exp" % name)
            r = ops.triplane_query(x.triplane, mlp, x.positions, radius=x.radius, density_bias=bias, want=(want,),
                                   align_corners=True)[want]
            if act == "normalize_channel_last":
                r = ops.normalize_rows3(r.reshape(-1, 3), 1e-7).view(r.shape)
            elif act not in (None, "trunc_exp", "sigmoid"):
                raise _lib.SculptError("MaterialMLP head %s: activation %s unsupported" % (name, act))
            out[name] = r[None]  # the reference keeps a leading batch dimension of 1
        return out

    def lattice_heads(self, planes, axis, radius, density_out_add=0.0, density_out=None):
        """The "density" and "vertex_offset" heads on the full lattice axis x axis x axis (flat order (ix*R + iy)*R + iz) --
        what triplane_to_meshes asks of query_triplane + this decoder (system.py:141-168) when the marching-tetrahedra grid is
        a regular lattice: the first layer is linear in the three plane samples, so it is evaluated once per lattice PAIR and
        plane (3 R^2 rows) instead of per point (ops.lattice_decode).  -> (density_act + density_out_add [R^3], offsets [R^3, 3]),
        or None when a head is not of the shape the lattice kernel writes (then the caller takes the per-point path)."""
        d, v = self.heads.get("density"), self.heads.get("vertex_offset")
        if d is None or v is None:
            return None
        (dh, dmlp), (vh, vmlp) = d, v
        if (dh["out_channels"], dh["output_activation"]) != (1, "trunc_exp") or \
                (vh["out_channels"], vh["output_activation"], float(vh["out_bias"])) != (3, None, 0.0):
            return None
        dens = ops.lattice_decode(planes, dmlp, axis, radius, density_bias=float(dh["out_bias"]), out_add=density_out_add,
                                  want=("density_act",), out=density_out)["density_act"]
        offs = ops.lattice_decode(planes, vmlp, axis, radius, want=("features",))["features"]
        return dens, offs


class Mesh:
    """sf3d/models/mesh.py:18-139, 236-262: positions, faces, lazily computed normals / tangents / UVs."""

    def __init__(self, v_pos, t_pos_idx, unwrapper=None, **extras):
        self.v_pos = v_pos
        self.t_pos_idx = t_pos_idx
        self._v_nrm = None
        self._v_tng = None
        self._v_tex = None
        self.extras = dict(extras)
        self.unwrapper = unwrapper

    @property
    def v_nrm(self):
        if self._v_nrm is None:
            self._v_nrm = ops.vertex_normals(self.v_pos, self.t_pos_idx)
        return self._v_nrm

    @property
    def v_tng(self):
        if self._v_tng is None:
            self._v_tng = ops.vertex_tangents(self.v_pos, self.v_tex, self.v_nrm, self.t_pos_idx)
        return self._v_tng

    @property
    def v_tex(self):
        if self._v_tex is None:
            self.unwrap_uv()
        return self._v_tex

    def unwrap_uv(self, island_padding: float = 0.02):
        """mesh.py:236-262: (uv, indices) from the unwrapper, then one vertex per face corner."""
        if self.unwrapper is None:
            raise _lib.SculptError(
                "Mesh.unwrap_uv: no UV unwrapper configured; set Mesh.unwrapper / SF3D.unwrapper = callable(v_pos, v_nrm, "
                "faces, island_padding) -> (uv [Nuv,2], indices [Nf,3]), e.g. sf3d.unwrap.BoxProjectionUnwrapper()")
        uv, indices = self.unwrapper(self.v_pos, self.v_nrm, self.t_pos_idx, island_padding)
        self.v_pos = self.v_pos[self.t_pos_idx].reshape(-1, 3).contiguous()
        self.t_pos_idx = torch.arange(self.v_pos.shape[0], device=self.v_pos.device,
                                      dtype=self.t_pos_idx.dtype).reshape(-1, 3)
        self._v_tex = uv[indices].reshape(-1, 2).contiguous()
        self._v_nrm = ops.vertex_normals(self.v_pos, self.t_pos_idx)
        self._v_tng = ops.vertex_tangents(self.v_pos, self._v_tex, self._v_nrm, self.t_pos_idx)


class MarchingTetrahedraHelper:
    """sf3d/models/isosurface.py:22-229 on the GPU (sculpt_mtet_*), static tables from sf3d/tets.py."""

    points_range = (0, 1)

    def __init__(self, resolution: int, tets_path: Optional[str], device):
        self.resolution = resolution
        v, idx, self.source = load_tets(resolution, tets_path)
        self.grid = ops.TetGrid(v, idx, device)
        self._indices_np = idx

    @property
    def grid_vertices(self):
        return self.grid.vertices

    @property
    def all_edges(self):
        return self.grid.edges.to(torch.int64)

    def normalize_grid_deformation(self, offsets):
        return ops.mtet_deform(self.grid, offsets, self.resolution) - self.grid.vertices

    def __call__(self, level, deformation=None, vert_mul=1.0, vert_add=0.0, unwrapper=None) -> Mesh:
        grid_vertices = self.grid.vertices if deformation is None else ops.mtet_deform(self.grid, deformation, self.resolution)
        v, f = ops.marching_tets(self.grid, grid_vertices, level, vert_mul, vert_add)
        return Mesh(v, f, unwrapper=unwrapper, grid_vertices=grid_vertices, tet_edges=self.grid.edges, grid_level=level,
                    grid_deformation=deformation)


class SF3D(KernelEngine):
    def __init__(self, cfg=None, precision="bf16", tets_path=None):
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision must be 'bf16' or 'fp32'")
        self.cfg = cfg or DEFAULT_CFG
        self.precision = precision
        self.adt = BF16 if precision == "bf16" else torch.float32
        self._spec = param_spec(self.cfg)
        self._sd = None
        self.device = None
        self._w = None
        self._buf = {}
        self.tets_path = tets_path
        self.isosurface_helper = None
        self.decoder = None
        # callable(v_pos, v_nrm, faces, island_padding) -> (uv, indices); default: the reference's box projection
        # (mesh.py:33 `self.unwrapper = Unwrapper()`) on the GPU.  None: no UVs (and no texture bake).
        from .unwrap import BoxProjectionUnwrapper

        self.unwrapper = BoxProjectionUnwrapper()
        # callable(mesh, mode, vertex_count) -> Mesh: Mesh.triangle_remesh (mesh.py:176-234) on the native decimate /
        # Botsch-Kobbelt code (sf3d/remesh.py); None: run_image refuses remesh != "none"
        from .remesh import default_remesher

        self.remesher = default_remesher()
        # triplane_to_meshes: evaluate the density / vertex_offset heads through the separable lattice kernels when the
        # tetrahedral grid is a regular lattice (False: always the per-point query, as the reference does it)
        self.lattice_decode = True
        # built by load_state_dict when the checkpoint carries their weights (estimators.py); None otherwise, and
        # then roughness / metallic stay None in run_image's dict
        self.image_estimator = None
        self.global_estimator = None

    # ------------------------------------------------------------------ loading
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, config_name: str, weight_name: str, device=None):
        """system.py:77-93 (config.yaml via PyYAML, model.safetensors via safetensors)."""
        if not os.path.isdir(pretrained_model_name_or_path):
            raise FileNotFoundError("Checkpoint directory given doesnt exist")
        cfg = load_config(os.path.join(pretrained_model_name_or_path, config_name))
        tets = os.path.join(pretrained_model_name_or_path, "..", "load", "tets",
                            "%d_tets.npz" % cfg["isosurface_resolution"])
        model = cls(cfg, tets_path=tets)
        from safetensors.torch import load_file

        model.load_state_dict(load_file(os.path.join(pretrained_model_name_or_path, weight_name)))
        return model

    def eval(self):
        return self

    def state_dict(self):
        return dict(self._sd or {})

    def load_state_dict(self, sd, strict=True):
        sd = {k: (torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v) for k, v in sd.items()}
        # the tokenizer registers every Modulation twice (image.py:38-50): accept the ModuleList alias
        for k in list(sd):
            if k.startswith("image_tokenizer.modulations."):
                n, rest = k[len("image_tokenizer.modulations."):].split(".", 1)
                alias = "image_tokenizer.model.encoder.layer.%d.norm%d_modulation.%s" % (int(n) // 2, int(n) % 2 + 1, rest)
                sd.setdefault(alias, sd[k])
        missing = [k for k in self._spec if k not in sd]
        unexpected = [k for k in sd if k not in self._spec and not k.startswith(IGNORED_PREFIXES)]
        if strict and (missing or unexpected):
            raise RuntimeError("Error(s) in loading state_dict for SF3D: missing %s unexpected %s"
                               % (missing[:5], unexpected[:5]))
        for k, shp in self._spec.items():
            if k in sd and tuple(sd[k].shape) != tuple(shp):
                raise RuntimeError("size mismatch for %s: %s vs %s" % (k, tuple(sd[k].shape), shp))
        self._sd = {k: sd[k].detach().to(torch.float32) for k in self._spec if k in sd}
        self._load_estimators(sd)
        if self.device is not None:
            self._prepare(self.device)
        return self

    def _load_estimators(self, sd):
        """system.py:109-114: the checkpoint's `image_estimator.*` (CLIP visual tower + heads) and `global_estimator.*`."""
        from .estimators import ClipBasedHeadEstimator, MultiHeadEstimator

        self.image_estimator = self.global_estimator = None
        if any(k.startswith("image_estimator.heads.") for k in sd):
            self.image_estimator = ClipBasedHeadEstimator(self.cfg.get("image_estimator"), self.precision).load_state_dict(sd)
        if any(k.startswith("global_estimator.layers.") for k in sd):
            self.global_estimator = MultiHeadEstimator(self.cfg.get("global_estimator"), self.precision).load_state_dict(sd)

    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise _lib.SculptError("SF3D runs on an MI355X only (device %s requested; there is no CPU fallback)" % device)
        self.device = device
        if self._sd is not None:
            self._prepare(device)
        return self

    def state_dict_estimators(self):
        out = {}
        for e in (self.image_estimator, self.global_estimator):
            if e is not None and e._sd is not None:
                out.update(e._sd)
        return out

    # ------------------------------------------------------------------ weight preparation
    def camera_embedding(self) -> np.ndarray:
        """LinearCameraEmbedder on the default camera (run_image: system.py:257-276) -- a constant of the config."""
        cfg = self.cfg
        c2w = default_cond_c2w(cfg["default_distance"])
        _, Kn = create_intrinsic_from_fov_deg(cfg["default_fovy_deg"], cfg["cond_image_size"], cfg["cond_image_size"])
        cond = np.concatenate([c2w.reshape(-1), Kn.reshape(-1)]).astype(np.float32)
        W = self._sd["camera_embedder.linear.weight"].numpy().astype(np.float64)
        b = self._sd["camera_embedder.linear.bias"].numpy().astype(np.float64)
        return (W @ cond.astype(np.float64) + b).astype(np.float32)

    def _attn_scale(self, scale):
        """bf16 mode: the query projections carry scale * log2(e) (ln_linear in _prepare) -> sculpt_attention_bf16_prescaled (scale None)."""
        return None if self.precision == "bf16" else scale

    def _prepare(self, dev):
        sd, cfg = self._sd, self.cfg
        for e in (self.image_estimator, self.global_estimator):
            if e is not None:
                e.to(dev)
        wt = _bf if self.precision == "bf16" else _f32
        pre = self.precision == "bf16"   # query projections carry softmax_scale * log2(e) (sculpt_attention_bf16_prescaled)

        def ln_linear(L, key, W, bias, gamma, beta, q_rows=0, head_dim=1):
            """A Linear fed by a LayerNorm: folded into the GEMM in bf16 mode (engine.prepare_ln_linear), with the
            first q_rows output rows (an attention's queries) carrying softmax_scale * log2(e)."""
            prepare_ln_linear(L, key, W, bias, torch.as_tensor(gamma), torch.as_tensor(beta), pre, lambda x: wt(x, dev),
                              lambda x: _f32(x, dev), q_rows if pre else 0, 1.4426950408889634 / math.sqrt(head_dim))

        v, b, t, pp = cfg["image_tokenizer"], cfg["backbone"], cfg["tokenizer"], cfg["post_processor"]
        H, P = v["hidden_size"], v["patch_size"]
        w = {}
        cam = self.camera_embedding().astype(np.float64)
        silu_cam = cam / (1.0 + np.exp(-cam))
        p = "image_tokenizer.model."
        pw = sd[p + "embeddings.patch_embeddings.projection.weight"].reshape(H, -1)
        kpad = ((pw.shape[1] + 63) // 64) * 64
        pwp = torch.zeros(H, kpad)
        pwp[:, : pw.shape[1]] = pw
        w["patch_k"] = kpad
        w["patch_w"], w["patch_b"] = wt(pwp, dev), _f32(sd[p + "embeddings.patch_embeddings.projection.bias"], dev)
        w["cls"] = _f32(sd[p + "embeddings.cls_token"].reshape(H), dev)
        w["dino"] = []
        for i in range(v["num_hidden_layers"]):
            q = p + "encoder.layer.%d." % i
            L, mod = {}, {}
            for j, ln in ((1, "norm1"), (2, "norm2")):
                # adaLN (attention.py:27-31): LN(x)*(1+scale)+shift with (scale|shift) = linear2(silu(cam)) constant
                Wm = sd[q + ln + "_modulation.linear2.weight"].numpy().astype(np.float64)
                bm = sd[q + ln + "_modulation.linear2.bias"].numpy().astype(np.float64)
                emb = Wm @ silu_cam + bm
                scale, shift = emb[:H], emb[H:]
                g = sd[q + ln + ".weight"].numpy().astype(np.float64)
                be = sd[q + ln + ".bias"].numpy().astype(np.float64)
                mod[j] = (torch.from_numpy((g * (1 + scale)).astype(np.float32)), torch.from_numpy((be * (1 + scale) + shift).astype(np.float32)))
            ln_linear(L, "qkv_w", torch.cat([torch.as_tensor(sd[q + "attention.attention.%s.weight" % n]) for n in ("query", "key", "value")], 0),
                      torch.cat([torch.as_tensor(sd[q + "attention.attention.%s.bias" % n]) for n in ("query", "key", "value")], 0),
                      mod[1][0], mod[1][1], q_rows=H, head_dim=H // v["num_attention_heads"])
            l1, l2 = sd[q + "layer_scale1.lambda1"], sd[q + "layer_scale2.lambda1"]
            # LayerScale (dinov2.py:380-396) folded into the preceding Linear
            L["o_w"] = wt(sd[q + "attention.output.dense.weight"] * l1[:, None], dev)
            L["o_b"] = _f32(sd[q + "attention.output.dense.bias"] * l1, dev)
            ln_linear(L, "f1_w", sd[q + "mlp.fc1.weight"], sd[q + "mlp.fc1.bias"], mod[2][0], mod[2][1])
            L["f2_w"] = wt(sd[q + "mlp.fc2.weight"] * l2[:, None], dev)
            L["f2_b"] = _f32(sd[q + "mlp.fc2.bias"] * l2, dev)
            w["dino"].append(L)
        w["dino_ln_w"], w["dino_ln_b"] = _f32(sd[p + "layernorm.weight"], dev), _f32(sd[p + "layernorm.bias"], dev)

        C, S = t["num_channels"], t["plane_size"]
        emb = sd["tokenizer.embeddings"]
        emb_ct = emb.permute(1, 0, 2, 3).reshape(C, 3 * S * S).contiguous()  # "Np Ct Hp Wp -> Ct (Np Hp Wp)"
        w["emb_ct"] = _f32(emb_ct, dev)
        w["emb_tc"] = _f32(emb_ct.t().contiguous(), dev)
        q = "backbone."

        def lin(name, bias=True):
            return wt(sd[q + name + ".weight"], dev), (_f32(sd[q + name + ".bias"], dev) if bias else None)

        def ln(name):
            return _f32(sd[q + name + ".weight"], dev), _f32(sd[q + name + ".bias"], dev)

        w["gn"] = ln("norm_triplane")
        w["proj_tri"] = lin("proj_triplane")
        w["norm_image"] = ln("norm_image")
        w["proj_image"] = lin("proj_image")
        w["norm_latent"] = ln("norm_latent")
        w["proj_latent"] = lin("proj_latent")
        w["latent_init"] = _f32(sd[q + "latent_init"][0], dev)
        w["proj_out"] = lin("proj_out")

        def fuse(key):
            F = {}
            Dq = sd[key + "attn.wq.weight"].shape[0]
            ln_linear(F, "q", sd[key + "attn.wq.weight"], None, sd[key + "norm_z1.weight"], sd[key + "norm_z1.bias"], q_rows=Dq,
                      head_dim=b["attention_head_dim"])
            kv = torch.cat([torch.as_tensor(sd[key + "attn.wk.weight"]), torch.as_tensor(sd[key + "attn.wv.weight"])], 0)
            if b.get("norm_x_input", False):
                ln_linear(F, "kv", kv, None, sd[key + "norm_x.weight"], sd[key + "norm_x.bias"])
            else:
                F["kv"] = wt(kv, dev)
            F["o"], F["ob"] = wt(sd[key + "attn.proj.weight"], dev), _f32(sd[key + "attn.proj.bias"], dev)
            ln_linear(F, "ff1", sd[key + "ff.net.0.proj.weight"], sd[key + "ff.net.0.proj.bias"], sd[key + "norm_z2.weight"],
                      sd[key + "norm_z2.bias"])
            F["ff2"], F["ff2_b"] = wt(sd[key + "ff.net.2.weight"], dev), _f32(sd[key + "ff.net.2.bias"], dev)
            return F

        w["blocks"] = []
        ca_k, ca_v = [], []
        for i in range(b["num_blocks"]):
            k = q + "main_blocks.%d." % i
            B = {"fuse_in": fuse(k + "fuse_block_in."), "fuse_out": fuse(k + "fuse_block_out."), "basic": []}
            for j in range(b["num_basic_blocks"]):
                kk = k + "transformer_block.%d." % j
                L = {}
                Db = b["num_attention_heads"] * b["attention_head_dim"]
                ln_linear(L, "sa_qkv", torch.cat([torch.as_tensor(sd[kk + "attn1.wq.weight"]), torch.as_tensor(sd[kk + "attn1.wk.weight"]),
                                                  torch.as_tensor(sd[kk + "attn1.wv.weight"])], 0), None,
                          sd[kk + "norm1.weight"], sd[kk + "norm1.bias"], q_rows=Db, head_dim=b["attention_head_dim"])
                L["sa_o"], L["sa_ob"] = wt(sd[kk + "attn1.proj.weight"], dev), _f32(sd[kk + "attn1.proj.bias"], dev)
                ln_linear(L, "ca_q", sd[kk + "attn2.wq.weight"], None, sd[kk + "norm2.weight"], sd[kk + "norm2.bias"], q_rows=Db,
                          head_dim=b["attention_head_dim"])
                ca_k.append(sd[kk + "attn2.wk.weight"])
                ca_v.append(sd[kk + "attn2.wv.weight"])
                L["ca_o"], L["ca_ob"] = wt(sd[kk + "attn2.proj.weight"], dev), _f32(sd[kk + "attn2.proj.bias"], dev)
                ln_linear(L, "ff1", sd[kk + "ff.net.0.proj.weight"], sd[kk + "ff.net.0.proj.bias"], sd[kk + "norm3.weight"],
                          sd[kk + "norm3.bias"])
                L["ff2"], L["ff2_b"] = wt(sd[kk + "ff.net.2.weight"], dev), _f32(sd[kk + "ff.net.2.bias"], dev)
                B["basic"].append(L)
            w["blocks"].append(B)
        # every BasicBlock's cross-attention K/V projection reads the same raw image tokens: one GEMM per image,
        # rows [K of block 0..n-1 | V of block 0..n-1]
        w["ca_kv_all"] = wt(torch.cat(ca_k + ca_v, 0), dev)
        # 3x3 convs as GEMMs over im2col rows: weight [Cout][ky][kx][Cin], Cout padded to a multiple of 128
        w["convs"] = []
        for i in range(pp["conv_layers"]):
            cw = sd["post_processor.upsample.%d.weight" % (2 * i)]  # [Cout, Cin, 3, 3]
            cb = sd["post_processor.upsample.%d.bias" % (2 * i)]
            co = cw.shape[0]
            cop = ((co + 127) // 128) * 128
            W2 = torch.zeros(cop, 9 * cw.shape[1])
            W2[:co] = cw.permute(0, 2, 3, 1).reshape(co, -1)
            b2 = torch.zeros(cop)
            b2[:co] = cb
            w["convs"].append((wt(W2, dev), _f32(b2, dev), co))
        self._w = w
        self._buf = {}
        self._const = None
        self.decoder = MaterialMLP(cfg["decoder"], {k: v.numpy() for k, v in sd.items() if k.startswith("decoder.")}, dev)
        self.isosurface_helper = MarchingTetrahedraHelper(cfg["isosurface_resolution"], self.tets_path, dev)
        r = np.float32(cfg["radius"])
        self._bbox_mul = float(np.float32(r) - np.float32(-r))  # bbox[1] - bbox[0] in fp32 (scale_tensor, utils.py:84-93)
        self._bbox_add = float(np.float32(-r))
        gv = self.isosurface_helper.grid.vertices.cpu().numpy()
        # scale_tensor(grid_vertices, (0,1), bbox): ((g - 0) / (1 - 0)) * (hi - lo) + lo, fp32 op by op
        world = ((gv - np.float32(0)) / np.float32(1)) * np.float32(self._bbox_mul) + np.float32(self._bbox_add)
        self._grid_world = _f32(world.astype(np.float32), dev)
        # Is the grid the plain (n x n x n) lattice in ij-order (the Kuhn stand-in is; the reference's npz file may not be)?
        # Then the two heads of triplane_to_meshes run through the separable lattice kernels (MaterialMLP.lattice_heads).
        self._lattice_axis = None
        n = int(round(gv.shape[0] ** (1.0 / 3.0)))
        if n >= 2 and n ** 3 == gv.shape[0]:
            w4 = world.astype(np.float32).reshape(n, n, n, 3)
            ax = w4[0, 0, :, 2].copy()
            if (np.array_equal(w4[..., 2], np.broadcast_to(ax[None, None, :], (n, n, n))) and
                    np.array_equal(w4[..., 1], np.broadcast_to(ax[None, :, None], (n, n, n))) and
                    np.array_equal(w4[..., 0], np.broadcast_to(ax[:, None, None], (n, n, n)))):
                self._lattice_axis = _f32(ax, dev)
        self._pos_cache = {}

    def _pos(self, n_side, dev):
        if n_side not in self._pos_cache:
            pe = self._sd["image_tokenizer.model.embeddings.position_embeddings"].numpy()
            self._pos_cache[n_side] = _f32(interpolate_pos_embedding(pe, n_side, "scale_factor"), dev)
        return self._pos_cache[n_side]

    # ------------------------------------------------------------------ forward
    def image_tokens(self, image_hwc: torch.Tensor):
        """DINOV2SingleImageTokenizer.forward for one [S,S,3] fp32 device image -> fp32 [Nt, H] (CLS kept)."""
        v, w = self.cfg["image_tokenizer"], self._w
        H, P, nh = v["hidden_size"], v["patch_size"], v["num_attention_heads"]
        M = int(H * v["mlp_ratio"])
        S = image_hwc.shape[0]
        n_side = S // P
        npatch = n_side * n_side
        T = npatch + 1
        Tp = ((T + 63) // 64) * 64
        patches = self._b("d_patches", (npatch, w["patch_k"]), self.adt)
        ops.vit_patchify(image_hwc, P, IMAGE_MEAN, IMAGE_STD, patches)
        pout = self._b("d_patch_out", (npatch, H), torch.float32)
        self._gemm(patches, w["patch_w"], bias=w["patch_b"], out_f32=pout)
        h = self._b("d_h", (T, H), torch.float32)
        ops.vit_assemble(pout, w["cls"], self._pos(n_side, image_hwc.device), h)
        st = self._stream_state("d", h)  # the residual stream + its bf16 copy + slice statistics (LayerNorm fold, engine.py)
        self._stats_of(st)
        qk = self._b("d_qk", (T, 2 * H), self.adt)
        vt = self._b("d_vt", (H, Tp), self.adt, zero=True)
        att = self._b("d_att", (T, H), self.adt)
        ff = self._b("d_ff", (T, M), self.adt)
        eps = v["layer_norm_eps"]
        for L in w["dino"]:
            self._ln_gemm(st, L, "qkv_w", eps, out_bf16=qk, out_t=vt, n_split=2 * H)  # adaLN folded into the projection
            self._attn(qk[:, :H], qk[:, H:], vt, att, T, T, nh, self._attn_scale(1.0 / math.sqrt(H // nh)))
            self._res_gemm(st, att, L["o_w"], L["o_b"])
            self._ln_gemm(st, L, "f1_w", eps, out_bf16=ff, epilogue=_lib.EPI_GELU)
            self._res_gemm(st, ff, L["f2_w"], L["f2_b"])
        tok = self._b("d_tok", (T, H), torch.float32)
        ops.layernorm(h, w["dino_ln_w"], w["dino_ln_b"], eps, y_f32=tok)
        return tok

    def _constants(self):
        """Input-independent activations: proj_triplane(GroupNorm(tokens)) and proj_latent(LN(latent_init))."""
        if self._const is None:
            w, b, t = self._w, self.cfg["backbone"], self.cfg["tokenizer"]
            C, T = t["num_channels"], 3 * t["plane_size"] ** 2
            D = b["num_attention_heads"] * b["attention_head_dim"]
            xn = torch.empty((T, C), dtype=self.adt, device=self.device)
            stats = torch.empty(2 * b["norm_num_groups"], dtype=torch.float32, device=self.device)
            ops.groupnorm_tokens(w["emb_ct"], b["norm_num_groups"], w["gn"][0], w["gn"][1], 1e-6, xn, stats)
            tri0 = torch.empty((T, D), dtype=torch.float32, device=self.device)
            self._gemm(xn, w["proj_tri"][0], bias=w["proj_tri"][1], out_f32=tri0)
            nl = b["num_latents"]
            ln = torch.empty((nl, D), dtype=self.adt, device=self.device)
            self._ln(w["latent_init"], w["norm_latent"][0], w["norm_latent"][1], 1e-5, ln)
            lat0 = torch.empty((nl, D), dtype=torch.float32, device=self.device)
            self._gemm(ln, w["proj_latent"][0], bias=w["proj_latent"][1], out_f32=lat0)
            self._const = (tri0, lat0)
        return self._const

    def _cast(self, x, name):
        """fp32 residual stream -> the activation storage type the K/V (or proj_out) GEMMs read."""
        if self.precision != "bf16":
            return x
        y = self._b(name, tuple(x.shape), BF16)
        ops.cast_bf16(x, y)
        return y

    def _ff(self, zst, L, rows, D, tag):
        """z += FF(LN(z)): GEGLU feed-forward with its LayerNorm folded into the first projection."""
        ffb = self._b(tag + "_ff", (rows, 4 * D), self.adt)
        self._ln_gemm(zst, L, "ff1", 1e-5, out_bf16=ffb, epilogue=_lib.EPI_GEGLU)
        self._res_gemm(zst, ffb, L["ff2"], L["ff2_b"])

    def _act(self, st):
        """The stream in the storage type the K/V (or proj_out) GEMMs read: its bf16 copy, kept current by every GEMM that
        writes the stream (fp32 mode: the stream itself)."""
        return st["hb"] if self.precision == "bf16" else st["h"]

    def _fuse(self, F, zst, xst, D, nh, tag):
        """FuseBlock.forward (backbone.py:249-257): z += attn(LN(z), x); z += FF(LN(z)), on the stream states of z and x
        (norm_x_input, False in the shipped config: K / V read LN(x), folded into their projection like the others)."""
        rows_z, rows_x = zst["h"].shape[0], xst["h"].shape[0]
        scale = self._attn_scale(1.0 / math.sqrt(D // nh))
        qb = self._b(tag + "_q", (rows_z, D), self.adt)
        self._ln_gemm(zst, F, "q", 1e-5, out_bf16=qb)
        kb = self._b(tag + "_k", (rows_x, D), self.adt)
        xp = ((rows_x + 63) // 64) * 64
        vt = self._b(tag + "_vt", (D, xp), self.adt, zero=True)
        if self.cfg["backbone"].get("norm_x_input", False):
            self._ln_gemm(xst, F, "kv", 1e-5, out_bf16=kb, out_t=vt, n_split=D, M=rows_x)
        else:
            self._gemm(self._act(xst), F["kv"], out_bf16=kb, out_t=vt, n_split=D, M=rows_x)
        att = self._b(tag + "_att", (rows_z, D), self.adt)
        self._attn(qb, kb, vt, att, rows_z, rows_x, nh, scale)
        self._res_gemm(zst, att, F["o"], F["ob"])
        self._ff(zst, F, rows_z, D, tag)

    def backbone_tokens(self, img_tok: torch.Tensor):
        """TwoStreamInterleaveTransformer.forward -> direct codes, token-major fp32 [3*S*S, C]
        (== the reference's [C, Nt] output transposed; token index = plane*S*S + h*S + w)."""
        w, b, t = self._w, self.cfg["backbone"], self.cfg["tokenizer"]
        C, T = t["num_channels"], 3 * t["plane_size"] ** 2
        nh = b["num_attention_heads"]
        D = nh * b["attention_head_dim"]
        Ni = img_tok.shape[0]
        nl = b["num_latents"]
        Lr = Ni + nl
        tri0, lat0 = self._constants()
        tri = self._b("bb_tri", (T, D), torch.float32)
        tri.copy_(tri0)
        tst = self._stream_state("bb_tri", tri)
        self._stats_of(tst)
        latent = self._b("bb_latent", (Lr, D), torch.float32)
        xn = self._b("bb_img_n", (Ni, img_tok.shape[1]), self.adt)
        self._ln(img_tok, w["norm_image"][0], w["norm_image"][1], 1e-5, xn)
        self._gemm(xn, w["proj_image"][0], bias=w["proj_image"][1], out_f32=latent[:Ni])
        latent[Ni:].copy_(lat0)
        lst = self._stream_state("bb_lat", latent)
        self._stats_of(lst)
        # cross-attention K / V^T of every BasicBlock from the raw image tokens, one GEMM
        img_act = self._cast(img_tok, "bb_img_act")
        nb = b["num_blocks"] * b["num_basic_blocks"]
        Nip = ((Ni + 63) // 64) * 64
        cak = self._b("bb_cak", (Ni, nb * D), self.adt)
        cavt = self._b("bb_cavt", (nb * D, Nip), self.adt, zero=True)
        self._gemm(img_act, w["ca_kv_all"], out_bf16=cak, out_t=cavt, n_split=nb * D)
        scale = self._attn_scale(1.0 / math.sqrt(D // nh))
        Lp = ((Lr + 63) // 64) * 64
        ib = 0
        for B in w["blocks"]:
            self._fuse(B["fuse_in"], lst, tst, D, nh, "fi")
            for L in B["basic"]:
                qk = self._b("bs_qk", (Lr, 2 * D), self.adt)
                vt = self._b("bs_vt", (D, Lp), self.adt, zero=True)
                self._ln_gemm(lst, L, "sa_qkv", 1e-5, out_bf16=qk, out_t=vt, n_split=2 * D)
                att = self._b("bs_att", (Lr, D), self.adt)
                self._attn(qk[:, :D], qk[:, D:], vt, att, Lr, Lr, nh, scale)
                self._res_gemm(lst, att, L["sa_o"], L["sa_ob"])
                qb = self._b("bs_q", (Lr, D), self.adt)
                self._ln_gemm(lst, L, "ca_q", 1e-5, out_bf16=qb)
                self._attn(qb, cak[:, ib * D:(ib + 1) * D], cavt[ib * D:(ib + 1) * D], att, Lr, Ni, nh, scale)
                self._res_gemm(lst, att, L["ca_o"], L["ca_ob"])
                self._ff(lst, L, Lr, D, "bs")
                ib += 1
            self._fuse(B["fuse_out"], tst, lst, D, nh, "fo")
        direct = self._b("bb_direct", (T, C), torch.float32)
        self._gemm(self._act(tst), w["proj_out"][0], bias=w["proj_out"][1], residual=w["emb_tc"], out_f32=direct)
        return direct

    def post_process(self, direct_tc: torch.Tensor):
        """PixelShuffleUpsampleNetwork.forward on token-major (= channel-last) codes -> [3, Co, S*r, S*r] fp32."""
        w, t, pp = self._w, self.cfg["tokenizer"], self.cfg["post_processor"]
        S, T = t["plane_size"], 3 * t["plane_size"] ** 2
        act = self._cast(direct_tc, "pp_act0")
        n = len(w["convs"])
        for i, (W2, b2, co) in enumerate(w["convs"]):
            cin = act.shape[1]
            implicit = self.precision == "bf16" and cin % 64 == 0  # the GEMM fetches the shifted pixels itself
            if not implicit:
                col = self._b("pp_col", (T, 9 * cin), self.adt)
                ops.im2col3x3(act, 3, S, col)
            if i != n - 1:
                nxt = self._b("pp_act%d" % (1 + i % 2), (T, W2.shape[0]), self.adt)
                if implicit:
                    ops.conv3x3_planes(act, 3, S, W2, b2, out_bf16=nxt, relu=True)
                else:
                    self._gemm(col, W2, bias=b2, out_bf16=nxt, epilogue=_lib.EPI_RELU)
                act = nxt[:, :co] if co != W2.shape[0] else nxt
                if not act.is_contiguous():
                    act = act.contiguous()
            else:
                g = self._b("pp_out", (T, W2.shape[0]), torch.float32)
                if implicit:
                    ops.conv3x3_planes(act, 3, S, W2, b2, out_f32=g)
                else:
                    self._gemm(col, W2, bias=b2, out_f32=g)
        r, Co = pp["scale_factor"], pp["out_channels"]
        planes = torch.empty((3, Co, S * r, S * r), dtype=torch.float32, device=self.device)
        ops.pixel_shuffle(g, planes, 3, S, Co, r)
        return planes

    def scene_code(self, image_hwc: torch.Tensor, want_direct=False):
        """get_scene_codes for one device image [S,S,3] fp32 -> scene code (and the direct codes [3,C,S,S])."""
        tok = self.image_tokens(image_hwc)
        direct = self.backbone_tokens(tok)
        planes = self.post_process(direct)
        if want_direct:
            t = self.cfg["tokenizer"]
            S, C = t["plane_size"], t["num_channels"]
            return planes, direct.view(3, S, S, C).permute(0, 3, 1, 2).contiguous()
        return planes

    def get_scene_codes(self, batch):
        """system.py:201-236.  batch["rgb_cond"]: [B,(Nv=1,)H,W,3] fp32; the camera entries are the default camera
        (folded at load)."""
        rgb = batch["rgb_cond"]
        if rgb.ndim == 5:
            if rgb.shape[1] != 1:
                raise _lib.SculptError("SF3D: single-view conditioning only (as run_image provides)")
            rgb = rgb[:, 0]
        codes, directs = [], []
        for i in range(rgb.shape[0]):
            img = rgb[i].to(self.device, torch.float32).contiguous()
            c, d = self.scene_code(img, want_direct=True)
            codes.append(c)
            directs.append(d)
        return torch.stack(codes, 0), torch.stack(directs, 0)

    # ------------------------------------------------------------------ geometry
    def query_triplane(self, positions, triplanes) -> TriplaneQuery:
        """system.py:170-199 -- returns the deferred query the decoder launches fused (see TriplaneQuery)."""
        if isinstance(triplanes, torch.Tensor) and positions.numel() >= 3 * 4096:
            triplanes = self.channel_last(triplanes)
        elif isinstance(triplanes, torch.Tensor):
            triplanes = triplanes.contiguous()
        return TriplaneQuery(positions, triplanes, self.cfg["radius"])

    def channel_last(self, triplanes: torch.Tensor) -> ops.ChannelLastPlanes:
        """[3,C,H,W] -> the [3,H,W,C] copy the point-query kernel reads fastest; the last conversion is kept."""
        key = (triplanes.data_ptr(), triplanes._version, tuple(triplanes.shape))
        if getattr(self, "_cl_key", None) != key:
            self._cl = ops.ChannelLastPlanes(triplanes)
            self._cl_key = key
        return self._cl

    def triplane_to_meshes(self, triplanes) -> List[Mesh]:
        """system.py:140-168"""
        meshes = []
        h = self.isosurface_helper
        for i in range(triplanes.shape[0]):
            sdf = self._b("sdf", (h.grid.n_vertices,), torch.float32)
            fast = None
            if self._lattice_axis is not None and self.lattice_decode:
                fast = self.decoder.lattice_heads(triplanes[i].contiguous(), self._lattice_axis, self.cfg["radius"],
                                                  density_out_add=-float(self.cfg["isosurface_threshold"]), density_out=sdf)
            if fast is not None:
                deform = fast[1]
            else:
                values = self.query_triplane(self._grid_world, triplanes[i])
                decoded = self.decoder(values, include=["vertex_offset", "density"])
                torch.sub(decoded["density"].reshape(-1), self.cfg["isosurface_threshold"], out=sdf)
                deform = decoded["vertex_offset"].reshape(-1, 3)
            meshes.append(h(sdf, deform, vert_mul=self._bbox_mul, vert_add=self._bbox_add, unwrapper=self.unwrapper))
        return meshes

    # ------------------------------------------------------------------ entry points
    def prepare_image(self, image):
        """system.py:285-305: RGBA PIL -> (mask [S,S,1], rgb composited on the background colour) fp32 on the device."""
        if image.mode != "RGBA":
            raise ValueError("Image must be in RGBA mode")
        S = self.cfg["cond_image_size"]
        img = np.asarray(image.resize((S, S))).astype(np.float32) / 255.0
        img = torch.from_numpy(img).float().clip(0, 1).to(self.device)
        mask = img[:, :, -1:]
        bg = torch.tensor(self.cfg["background_color"], device=self.device)[None, None, :]
        rgb = torch.lerp(bg, img[:, :, :3], mask)
        return mask, rgb

    def run_image(self, image, bake_resolution: int, remesh="none", vertex_simplification_factor="high",
                  estimate_illumination: bool = False, enable_texture: bool = True):
        """system.py:238-283"""
        if isinstance(image, list):
            pairs = [self.prepare_image(im) for im in image]
            mask_cond = torch.stack([p[0] for p in pairs], 0)
            rgb_cond = torch.stack([p[1] for p in pairs], 0)
            batch_size = rgb_cond.shape[0]
        else:
            mask_cond, rgb_cond = self.prepare_image(image)
            mask_cond, rgb_cond = mask_cond[None], rgb_cond[None]
            batch_size = 1
        batch = {"rgb_cond": rgb_cond, "mask_cond": mask_cond}
        meshes, global_dict = self.generate_mesh(batch, bake_resolution, remesh, vertex_simplification_factor,
                                                 estimate_illumination, enable_texture)
        if batch_size == 1:
            return meshes[0], global_dict
        return meshes, global_dict

    def generate_mesh(self, batch, bake_resolution: int, remesh="none", vertex_simplification_factor="high",
                      estimate_illumination: bool = False, enable_texture: bool = True):
        """system.py:307-526"""
        S = self.cfg["cond_image_size"]
        rgb = batch["rgb_cond"]
        if rgb.shape[-3] != S or rgb.shape[-2] != S:
            rgb = torch.stack([ops.resize_aa_bilinear(im.contiguous(), S) for im in rgb], 0)
            batch = dict(batch, rgb_cond=rgb)
        scene_codes, direct_codes = self.get_scene_codes(batch)
        global_dict = {}
        if self.image_estimator is not None:
            # system.py:326-329: image_estimator(rgb_cond * mask_cond); the product is formed inside the resize kernel
            rgb_e = batch["rgb_cond"].reshape(-1, *batch["rgb_cond"].shape[-3:]).to(self.device, torch.float32)
            mask_e = batch["mask_cond"].to(self.device, torch.float32)
            mask_e = mask_e.reshape(-1, *mask_e.shape[-3:])
            if mask_e.shape[1] != S or mask_e.shape[2] != S:   # image_processor on mask_cond (system.py:321-323)
                mask_e = torch.stack([ops.resize_aa_bilinear(m.contiguous(), S) for m in mask_e], 0)
            global_dict.update(self.image_estimator(rgb_e, mask=mask_e[..., 0].contiguous()))
        if self.global_estimator is not None and estimate_illumination:
            # system.py:330-331 on non_postprocessed_codes [B,3,C,S,S], handed over channel-last
            Sp = direct_codes.shape[-1]
            toks = [d.permute(0, 2, 3, 1).reshape(3 * Sp * Sp, -1).contiguous() for d in direct_codes]
            global_dict.update(self.global_estimator(toks, Sp))
        meshes = self.triplane_to_meshes(scene_codes)
        rets = []
        for i, mesh in enumerate(meshes):
            if mesh.v_pos.shape[0] == 0:
                rets.append(None)
                continue
            if vertex_simplification_factor == "high":
                vertex_count = round(0.75 * mesh.v_pos.shape[0])
            elif vertex_simplification_factor == "med":  # the GUI passes 'medium' -> falls to low, as in the reference
                vertex_count = round(0.4 * mesh.v_pos.shape[0])
            else:
                vertex_count = round(0.1 * mesh.v_pos.shape[0])
            if remesh in ("triangle", "quad"):
                if self.remesher is None:
                    raise _lib.SculptError("remesh=%r needs SF3D.remesher (sf3d/remesh.py; mesh.py:176-234)" % remesh)
                mesh = self.remesher(mesh, remesh, vertex_count)
            uvs = None
            tex = dict(basecolor_tex=None, bump_tex=None, roughness=None, metallic=None)
            if self.unwrapper is not None:
                mesh.unwrapper = self.unwrapper
                mesh.unwrap_uv()
                uvs = mesh.v_tex
                if enable_texture:
                    from .bake import bake_textures

                    tex = bake_textures(self, mesh, scene_codes[i], bake_resolution, global_dict, i)
            elif enable_texture:
                raise _lib.SculptError("enable_texture needs a UV unwrapper (SF3D.unwrapper); see Mesh.unwrap_uv")
            rets.append({"vertices": mesh.v_pos.cpu().numpy(), "faces": mesh.t_pos_idx.cpu().numpy(),
                         "uvs": None if uvs is None else uvs.cpu().numpy(), **tex})
        return rets, global_dict

    def import_mesh_blender(self, mesh, mesh_name="GeneratedMesh"):
        from .blender_sink import import_mesh_blender

        return import_mesh_blender(mesh, mesh_name)


def load_config(config_yaml: str):
    """checkpoints/config.yaml -> the dict layout of spec.DEFAULT_CFG (class paths dropped, Config defaults filled)."""
    import copy

    import yaml

    with open(config_yaml) as f:
        y = yaml.safe_load(f)
    cfg = copy.deepcopy(DEFAULT_CFG)
    for k in ("cond_image_size", "isosurface_resolution", "isosurface_threshold", "radius", "default_fovy_deg",
              "default_distance"):
        if k in y:
            cfg[k] = y[k]
    cfg["camera_embedder"].update({k: v for k, v in y.get("camera_embedder", {}).items() if k != "conditions"})
    it = y.get("image_tokenizer", {})
    if "modulation_cond_dim" in it:
        cfg["image_tokenizer"]["modulation_cond_dim"] = it["modulation_cond_dim"]
    cfg["tokenizer"].update(y.get("tokenizer", {}))
    cfg["backbone"].update(y.get("backbone", {}))
    cfg["post_processor"].update(y.get("post_processor", {}))
    d = y.get("decoder", {})
    if d:
        heads = []
        for h in d.get("heads", []):
            heads.append(dict(name=h["name"], out_channels=h["out_channels"], n_hidden_layers=h["n_hidden_layers"],
                              output_activation=h.get("output_activation"), out_bias=float(h.get("out_bias", 0.0))))
        cfg["decoder"] = dict(in_channels=d.get("in_channels", 120), n_neurons=d.get("n_neurons", 64),
                              activation=d.get("activation", "silu"), heads=tuple(heads))
    from .estimators import GLOBAL_ESTIMATOR_CFG, IMAGE_ESTIMATOR_CFG

    for key, base in (("image_estimator", IMAGE_ESTIMATOR_CFG), ("global_estimator", GLOBAL_ESTIMATOR_CFG)):
        if y.get(key):
            e = dict(base)
            e.update({k: v for k, v in y[key].items() if k != "heads"})
            if y[key].get("heads"):
                e["heads"] = tuple(dict(h) for h in y[key]["heads"])
            cfg[key] = e
    return cfg
