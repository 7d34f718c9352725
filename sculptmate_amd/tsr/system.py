"""TSR -- the TripoSR system behind SculptMate's "lean" generator, MI355X-native.

Mirrors the surface of /root/reference/TripoSR/tsr/system.py::TSR that the add-on uses
(SURVEY.md section 8b): from_pretrained, renderer.set_chunk_size, to(device), __call__/forward,
extract_mesh, plus a headless run() that returns meshes.  All arithmetic of the hot path runs in
libsculpt_hip.so (hand-written HIP, gfx950); torch only owns HBM buffers and streams.

Pipeline per image (reference file:line):
  ImagePreprocessor (host)                          tsr/utils.py:62-112
  DINOSingleImageTokenizer = ViT-B/16, 1025 tokens  tsr/models/tokenizers/image.py:41-60
  Triplane1DTokenizer tokens [1024, 3072]           tsr/models/tokenizers/triplane.py:29-45
  Transformer1D, 16 blocks                          tsr/models/transformer/transformer_1d.py:179-219
  TriplaneUpsampleNetwork -> scene code [3,40,64,64] tsr/models/network_utils.py:24-32
  extract_mesh: density grid -> marching cubes      tsr/system.py:171-200
"""
import math
import os
from typing import List, Optional

import numpy as np
import torch

from .. import _lib, ops
from ..engine import KernelEngine, prepare_ln_linear
from .posemb import interpolate_pos_embedding
from .utils import ImagePreprocessor

BF16 = torch.bfloat16
LOG2E = 1.4426950408889634

from .spec import DEFAULT_CFG, IMAGE_MEAN, IMAGE_STD, param_spec  # noqa: E402,F401


class Mesh:
    """What run() returns per image: trimesh-constructible arrays."""

    def __init__(self, vertices, faces, vertex_colors=None):
        self.vertices = vertices
        self.faces = faces
        self.vertex_colors = vertex_colors

    def to_trimesh(self):  # pragma: no cover (trimesh is optional)
        import trimesh

        return trimesh.Trimesh(vertices=self.vertices, faces=self.faces, vertex_colors=self.vertex_colors)

    def export(self, path):
        """Write .obj / .ply / .glb by extension (sculptmate_amd/meshio.py), the call upstream users make on the
        trimesh object their extract_mesh returns."""
        from .. import meshio

        ext = str(path).rsplit(".", 1)[-1].lower()
        writer = {"obj": meshio.write_obj, "ply": meshio.write_ply, "glb": meshio.write_glb}.get(ext)
        if writer is None:
            raise ValueError(f"unsupported mesh format .{ext} (obj, ply, glb)")
        writer(str(path), self.vertices, self.faces, vertex_colors=self.vertex_colors)


class _PinnedPool:
    """Pinned host buffers for the mesh hand-off, recycled by size class (next power of two of the byte count).  A pinned
    allocation is a driver call of ~0.5 ms that can stall the device queue; mesh sizes differ from image to image, so an
    exact-size cache never hits."""

    def __init__(self):
        self.free = {}

    def take(self, shape, dtype):
        n = 1
        for d in shape:
            n *= int(d)
        esize = torch.empty((), dtype=dtype).element_size()
        nbytes = n * esize
        cap = 1 << (max(nbytes, 16) - 1).bit_length()   # at least one element of any dtype: an empty mesh array gets a real buffer
        lst = self.free.get(cap)
        buf = lst.pop() if lst else torch.empty(cap, dtype=torch.uint8, pin_memory=True)
        view = buf[:max(nbytes, esize)].view(dtype)[:n].view(shape)
        return _PinnedLease(self, cap, buf), view

    def give(self, cap, buf):
        self.free.setdefault(cap, []).append(buf)


class _PinnedLease:
    """One pinned buffer on loan.  It goes back to the pool when the lease dies -- or, once `hand_over(array)` has tied it to
    the NumPy array that views it, when THAT array (and every view of it) is gone."""

    def __init__(self, pool, cap, buf):
        self.pool, self.cap, self.buf = pool, cap, buf

    def hand_over(self, array):
        import weakref

        pool, cap, buf = self.pool, self.cap, self.buf
        self.buf = None
        weakref.finalize(array, pool.give, cap, buf)

    def __del__(self):
        try:
            if self.buf is not None:
                self.pool.give(self.cap, self.buf)
        except Exception:  # interpreter shutdown
            pass


class PendingMesh:
    """A mesh whose device -> pinned-host copy is in flight (TSR.run_async)."""

    def __init__(self, host_tensors, done_event, leases=()):
        self._host = host_tensors
        self._done = done_event
        self._leases = leases
        self._mesh = None

    def done(self) -> bool:
        return self._done.query()

    def result(self) -> Mesh:
        """The arrays are views of pinned host buffers; a buffer returns to the pool when its array -- and every view or
        slice taken from it -- has been released (it does not matter whether the Mesh object itself is kept)."""
        if self._mesh is None:
            self._done.synchronize()
            arrays = [None if t is None else t.numpy() for t in self._host]
            for lease, a in zip(self._leases, [a for a in arrays if a is not None]):
                lease.hand_over(a)
            # later calls return the SAME Mesh: a second set of views would carry no lease, and the buffers could go back to
            # the pool (and be overwritten by a later run_async) while it is still alive
            self._leases, self._host = (), None
            self._mesh = Mesh(*arrays)
        return self._mesh


_MC_SIGN_PLANES = not _lib.form_has("SCULPT_MC_FORM", "noplanes")   # A/B: the plain count phase over the whole volume


class PendingTokens:
    """The image tokens of one image, being computed on the tokenizer stream (TSR.tokens_async)."""

    __slots__ = ("ctx", "ready", "_slot", "image")

    def __init__(self, ctx, ready, slot, image=None):
        self.ctx, self.ready, self._slot, self.image = ctx, ready, slot, image


class MarchingCubeHelper:
    """tsr/models/isosurface.py:17-54 on the GPU (sculpt_mc_*)."""

    points_range = (0, 1)

    def __init__(self, resolution: int):
        self.resolution = resolution
        self._grid_vertices = None

    @property
    def grid_vertices(self) -> torch.Tensor:
        """[R^3, 3] lattice in [0,1] (isosurface.py:25-39).  Only built if somebody asks for it: the
        dense query generates positions from the separable axis table instead (no 201 MB tensor)."""
        if self._grid_vertices is None:
            R = self.resolution
            x = torch.linspace(*self.points_range, R)
            x, y, z = torch.meshgrid(x, x, x, indexing="ij")
            self._grid_vertices = torch.cat([x.reshape(-1, 1), y.reshape(-1, 1), z.reshape(-1, 1)], dim=-1)
        return self._grid_vertices

    def __call__(self, level: torch.Tensor):
        """level = -(density - threshold) as the reference passes it; returns (v_pos in [0,1], faces int64)."""
        R = self.resolution
        vol = (-level).reshape(R, R, R).contiguous()
        return ops.marching_cubes(vol, 0.0, reference_order=True, vert_div=R - 1.0, vert_mul=1.0, vert_add=0.0)


class TriplaneNeRFRenderer:
    """tsr/models/nerf_renderer.py:17-91 (mesh path only; the volume renderer is never called)."""

    def __init__(self, cfg):
        self.cfg = type("Cfg", (), dict(cfg))()
        assert cfg.get("feature_reduction", "concat") == "concat"
        assert cfg.get("density_activation", "exp") == "exp"
        self.chunk_size = 0

    def set_chunk_size(self, chunk_size: int):
        assert chunk_size >= 0, "chunk_size must be a non-negative integer (0 for no chunking)."
        self.chunk_size = chunk_size  # kept for API parity; the fused kernel needs no chunking

    def query_triplane(self, decoder, positions, triplane):
        triplane = triplane.contiguous()
        if positions.numel() >= 3 * 4096:  # many points (vertex colours): channel-last taps, ~20x fewer cache lines
            triplane = ops.ChannelLastPlanes(triplane)
        return ops.triplane_query(triplane, decoder, positions, radius=self.cfg.radius,
                                  density_bias=self.cfg.density_bias)


def _bf(x, dev):
    return torch.as_tensor(x).to(device=dev, dtype=BF16).contiguous()


def _f32(x, dev):
    return torch.as_tensor(x).to(device=dev, dtype=torch.float32).contiguous()


class TSR(KernelEngine):
    def __init__(self, cfg=None, pos_embed_mode="scale_factor", precision="bf16", decoder_precision="bf16l3", decoder_filter=True):
        """precision: "bf16" (BASELINE config 2: bf16 storage, fp32 accumulate -- what bench.py times),
        "fp32" (parity mode: the whole transformer on the exact-fp32 matrix pipe, like the fp32 reference), or
        "bf16l3" (the fast parity mode: fp32 storage, norms and softmax as in "fp32", every matrix product on the bf16 matrix pipe
        with both operands split exactly into three bf16 limbs and fp32 accumulation -- fp32-equivalent, csrc/gemm_l3.hip), or
        "fp16l2" ("bf16l3" with the Linears of the two transformers on TWO fp16 limbs per operand, 22 significant bits, three
        products per multiply: fp32-equivalent on this model -- scene code 1e-6 from the fp32 reference like "fp32" itself -- at
        two thirds of the time; an activation beyond 65504 in magnitude makes the scene code non-finite: forward() then repeats the
        call on a "bf16l3" twin of the model and counts it in range_fallbacks).
        decoder_precision: how the 64x64 hidden layers of the dense density query (extract_mesh's 256^3 grid) are evaluated.
          "bf16l3" (default): fp32-equivalent -- both operands split EXACTLY into three bf16 limbs (24 significant bits, fp32
                   exponent range), six exact products per weight on the bf16 matrix pipe, fp32 accumulation; no range
                   limit and no fallback; measured error against the CPU oracle = the fp32 kernel's own.
          "fp32":  the exact-fp32 MFMA kernel (a k-ordered fmaf chain, 1.5x slower): the parity mode.
        decoder_filter (default True; applies to decoder_precision "bf16l3"): extract_mesh evaluates the dense grid in two passes
          (csrc/density_filter.hip) -- every lattice point with one 16-bit product per hidden layer, then the three-limb arithmetic at
          all corners of every cell that can be active -- which gives the mesh of the full evaluation bit for bit as long as no coarse
          error reaches the calibrated margin (8 x the largest error measured on a 64^3 probe of the first scene code); every call is
          guarded by the largest coarse error seen at the re-evaluated points and redone in full when that exceeds a third of the
          margin (filter_info counts both).  False: every lattice point with the three-limb arithmetic."""
        if precision not in ("bf16", "fp32", "bf16l3", "fp16l2"):
            raise ValueError("precision must be 'bf16', 'fp32', 'bf16l3' or 'fp16l2'")
        if decoder_precision not in ("fp32", "bf16l3"):
            raise ValueError("decoder_precision must be 'bf16l3' or 'fp32'")
        self.decoder_precision = decoder_precision
        self.decoder_filter = bool(decoder_filter)
        # state of the two-pass density grid: margin (None = not calibrated yet), coarse operand type, counters
        self.filter_info = {"margin": None, "coarse": "fp16", "usable": True, "calibrations": 0, "filtered": 0, "fallbacks": 0,
                            "last": None}
        self.cfg = cfg or DEFAULT_CFG
        self.pos_embed_mode = pos_embed_mode
        self.precision = precision
        self.l3p = False   # set by _prepare: the three-limb mode with operands split once (engine.py)
        self.range_fallbacks = 0   # fp16l2: forward() calls redone on three bf16 limbs because an activation left the fp16 range
        self._range_twin = None    # the "bf16l3" model those calls run on (built on first need from the current weights)
        self.adt = BF16 if precision == "bf16" else torch.float32  # activation / weight storage type
        self._spec = param_spec(self.cfg)
        self._sd = None
        self.device = None
        self.renderer = TriplaneNeRFRenderer(self.cfg["renderer"])
        self.image_processor = ImagePreprocessor()
        self.isosurface_helper = None
        self.decoder = None  # ops.PackedMLP after to(device)
        self.mesh_sink = None  # callable(verts, faces, colors, name); default: bpy if importable
        # forward(): images per transformer pass.  1 (default): image by image.  > 1: the reference's batched pass
        # (system.py:82-115) over stacked token rows; in the bf16 mode it gives each image the bits of its single-image pass
        # (every GEMM keeps the single-image tile form: encode_images), which is why run() batches by default there
        self.max_batch = 1
        # A stacked pass keeps the tile forms of the single-image pass (True: each image gets the scene code of its own pass bit for
        # bit; bf16 mode) or takes the forms that are fastest for the stacked rows (False: round 4-5 behaviour -- 0.7 ms per image
        # faster at four images, scene codes 2.7e-3 from the single-image ones: bf16 roundings flip in the LayerNorm statistics)
        self.batch_exact = True
        self._w = None
        self._pos_cache = {}
        self._buf = {}

    # ------------------------------------------------------------------ loading
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, config_name: str, weight_name: str, **model_kwargs):
        """system.py:51-66.  config.yaml is read with PyYAML (the ${tokenizer.num_channels}
        interpolation is resolved by hand); ViT hyper-parameters come from checkpoints/config.json.
        model_kwargs (beyond the reference's signature): TSR's own keyword arguments -- precision="bf16" | "bf16l3" | "fp32",
        decoder_precision, decoder_filter, pos_embed_mode."""
        if not os.path.isdir(pretrained_model_name_or_path):
            raise FileNotFoundError("Checkpoint directory given doesnt exist")
        cfg = load_config(os.path.join(pretrained_model_name_or_path, config_name),
                          os.path.join(pretrained_model_name_or_path, "config.json"))
        model = cls(cfg, **model_kwargs)
        ckpt = torch.load(os.path.join(pretrained_model_name_or_path, weight_name), map_location="cpu")
        model.load_state_dict(ckpt)
        return model

    def state_dict(self):
        return dict(self._sd or {})

    def load_state_dict(self, sd, strict=True):
        sd = {k: (torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v) for k, v in sd.items()}
        missing = [k for k in self._spec if k not in sd]
        unexpected = [k for k in sd if k not in self._spec]
        if strict and (missing or unexpected):
            raise RuntimeError("Error(s) in loading state_dict for TSR: missing %s unexpected %s"
                               % (missing[:5], unexpected[:5]))
        for k, shp in self._spec.items():
            if k in sd and tuple(sd[k].shape) != tuple(shp):
                raise RuntimeError("size mismatch for %s: %s vs %s" % (k, tuple(sd[k].shape), shp))
        self._sd = {k: sd[k].detach().to(torch.float32) for k in self._spec if k in sd}
        # state derived from the previous weights: the fp16l2 range twin, the two-pass grid's calibration
        self._range_twin = None
        self.filter_info.update(margin=None, usable=True, last=None)
        if self.device is not None:
            self._prepare(self.device)
        return self

    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise _lib.SculptError("TSR runs on an MI355X only (device %s requested; there is no CPU fallback)" % device)
        self.device = device
        if self._sd is not None:
            self._prepare(device)
        return self

    # ------------------------------------------------------------------ weight preparation
    def _prepare(self, dev):
        sd, cfg = self._sd, self.cfg
        wt = _bf if self.precision == "bf16" else _f32  # GEMM weight storage
        # "limbs once" (engine.py): in the three-limb mode the Linears of the two transformers keep their weights split
        # (the attention must be the fused kernel: the three-launch composition has no limb output)
        # A/B forms (tests): SCULPT_ATTN_FORM=l3unfused = the three-launch composition of the limb attention; SCULPT_L3_TILE=split =
        # every GEMM splits its operands while staging (the form before limbs-once)
        fused_attn = not _lib.form_has("SCULPT_ATTN_FORM", "l3unfused")
        self.l3p = self.precision == "fp16l2" or (self.precision == "bf16l3" and not _lib.form_has("SCULPT_L3_TILE", "split") and fused_attn)
        if self.l3p:
            # sculpt_gemm_l3p's tiles: N % 128 == 0 (N % 64 per GEGLU half), K % 32 == 0 for every Linear kept as limbs.  A model
            # with other widths runs "bf16l3" on the splitting GEMM (csrc/gemm_l3.hip: N % 4), as it did before limbs-once.
            vv, bb = cfg["image_tokenizer"], cfg["backbone"]
            Hh, Ii = vv["hidden_size"], vv["intermediate_size"]
            Dd = bb["num_attention_heads"] * bb["attention_head_dim"]
            bad = [n for n, ok in (("image_tokenizer.hidden_size", Hh % 128 == 0), ("image_tokenizer.intermediate_size", Ii % 128 == 0),
                                   ("backbone width", Dd % 128 == 0), ("backbone.cross_attention_dim", bb["cross_attention_dim"] % 32 == 0))
                   if not ok]
            if self.precision == "fp16l2":
                if bad or not fused_attn:
                    raise ValueError("precision='fp16l2' needs Linear widths the limb GEMM tiles (multiples of 128; cross-attention "
                                     "width a multiple of 32) and the fused limb attention (SCULPT_L3_ATTN_FUSED unset): "
                                     + (", ".join(bad) or "SCULPT_ATTN_FORM=l3unfused"))
            elif bad:
                self.l3p = False
        self.limb_format = "f16x2" if self.precision == "fp16l2" else "bf16x3"
        fmt = self.limb_format
        wh = (lambda x, d, geglu=False: ops.Limbs.of(_f32(ops.geglu_row_blocks(torch.as_tensor(x)) if geglu else x, d), fmt=fmt,
                                                     weight=True)) if self.l3p else (lambda x, d, geglu=False: wt(x, d))
        v, b = cfg["image_tokenizer"], cfg["backbone"]
        H = v["hidden_size"]
        w = {}
        p = "image_tokenizer.model."
        w["patch_w"] = wt(sd[p + "embeddings.patch_embeddings.projection.weight"].reshape(H, -1), dev)
        w["patch_b"] = _f32(sd[p + "embeddings.patch_embeddings.projection.bias"], dev)
        w["cls"] = _f32(sd[p + "embeddings.cls_token"].reshape(H), dev)
        fold = self.precision == "bf16"

        def ln_linear(L, key, W, bias, gamma, beta, q_rows=0, q_scale=1.0, geglu=False):
            """A Linear fed by a LayerNorm (engine.prepare_ln_linear): folded into the GEMM in bf16 mode."""
            prepare_ln_linear(L, key, W, bias, gamma, beta, fold, lambda x: wh(x, dev, geglu), lambda x: _f32(x, dev), q_rows, q_scale)

        w["vit"] = []
        for i in range(v["num_hidden_layers"]):
            q = p + "encoder.layer.%d." % i
            L = {}
            ln_linear(L, "qkv_w", torch.cat([sd[q + "attention.attention.%s.weight" % n] for n in ("query", "key", "value")], 0),
                      torch.cat([sd[q + "attention.attention.%s.bias" % n] for n in ("query", "key", "value")], 0),
                      sd[q + "layernorm_before.weight"], sd[q + "layernorm_before.bias"], q_rows=H,
                      q_scale=LOG2E / math.sqrt(H // v["num_attention_heads"]))
            L["o_w"], L["o_b"] = wh(sd[q + "attention.output.dense.weight"], dev), _f32(sd[q + "attention.output.dense.bias"], dev)
            ln_linear(L, "f1_w", sd[q + "intermediate.dense.weight"], sd[q + "intermediate.dense.bias"],
                      sd[q + "layernorm_after.weight"], sd[q + "layernorm_after.bias"])
            L["f2_w"], L["f2_b"] = wh(sd[q + "output.dense.weight"], dev), _f32(sd[q + "output.dense.bias"], dev)
            w["vit"].append(L)
        w["vit_ln_w"], w["vit_ln_b"] = _f32(sd[p + "layernorm.weight"], dev), _f32(sd[p + "layernorm.bias"], dev)

        t = cfg["tokenizer"]
        C, S = t["num_channels"], t["plane_size"]
        emb = sd["tokenizer.embeddings"]  # [3, C, S, S]
        emb_ct = emb.permute(1, 0, 2, 3).reshape(C, 3 * S * S).contiguous()  # "Np Ct Hp Wp -> Ct (Np Hp Wp)"
        w["emb_ct"] = _f32(emb_ct, dev)
        w["emb_tc"] = _f32(emb_ct.t().contiguous(), dev)  # residual in token-major layout
        w["gn_w"], w["gn_b"] = _f32(sd["backbone.norm.weight"], dev), _f32(sd["backbone.norm.bias"], dev)
        w["pin_w"], w["pin_b"] = wt(sd["backbone.proj_in.weight"], dev), _f32(sd["backbone.proj_in.bias"], dev)
        w["pout_w"], w["pout_b"] = wt(sd["backbone.proj_out.weight"], dev), _f32(sd["backbone.proj_out.bias"], dev)
        w["blocks"] = []
        for i in range(b["num_layers"]):
            q = "backbone.transformer_blocks.%d." % i
            L = {}
            Db, qs = b["num_attention_heads"] * b["attention_head_dim"], LOG2E / math.sqrt(b["attention_head_dim"])
            ln_linear(L, "sa_qkv", torch.cat([sd[q + "attn1.to_q.weight"], sd[q + "attn1.to_k.weight"], sd[q + "attn1.to_v.weight"]], 0),
                      None, sd[q + "norm1.weight"], sd[q + "norm1.bias"], q_rows=Db, q_scale=qs)
            L["sa_o"], L["sa_ob"] = wh(sd[q + "attn1.to_out.0.weight"], dev), _f32(sd[q + "attn1.to_out.0.bias"], dev)
            ln_linear(L, "ca_q", sd[q + "attn2.to_q.weight"], None, sd[q + "norm2.weight"], sd[q + "norm2.bias"], q_rows=Db, q_scale=qs)
            L["_ca_k"], L["_ca_v"] = sd[q + "attn2.to_k.weight"], sd[q + "attn2.to_v.weight"]
            L["ca_o"], L["ca_ob"] = wh(sd[q + "attn2.to_out.0.weight"], dev), _f32(sd[q + "attn2.to_out.0.bias"], dev)
            ln_linear(L, "ff1", sd[q + "ff.net.0.proj.weight"], sd[q + "ff.net.0.proj.bias"], sd[q + "norm3.weight"], sd[q + "norm3.bias"],
                      geglu=True)
            L["ff2"], L["ff2_b"] = wh(sd[q + "ff.net.2.weight"], dev), _f32(sd[q + "ff.net.2.bias"], dev)
            w["blocks"].append(L)
        # the cross-attention K/V projections of ALL layers depend only on the image tokens: one GEMM
        # [Tc, 768] x [L*2*D, 768]^T per image, rows ordered [K of layer 0..L-1 | V of layer 0..L-1]
        w["ca_kv_all"] = wh(torch.cat([L.pop("_ca_k") for L in w["blocks"]] + [L.pop("_ca_v") for L in w["blocks"]], 0), dev)
        # ConvTranspose2d(k2,s2) as a GEMM: rows (co,dy,dx), K = Cin; rows padded to a multiple of 128
        up = sd["post_processor.upsample.weight"]  # [Cin, Co, 2, 2]
        Co = up.shape[1]
        rows = up.permute(1, 2, 3, 0).reshape(4 * Co, up.shape[0])
        npad = ((4 * Co + 127) // 128) * 128
        upw = torch.zeros(npad, up.shape[0])
        upw[: 4 * Co] = rows
        w["up_w"], w["up_b"] = wt(upw, dev), _f32(sd["post_processor.upsample.bias"], dev)
        self._w = w
        d = cfg["decoder"]
        n = d["n_hidden_layers"] + 1
        self.decoder = ops.PackedMLP([sd["decoder.layers.%d.weight" % (2 * i)] for i in range(n)],
                                     [sd["decoder.layers.%d.bias" % (2 * i)] for i in range(n)], dev)
        self._pos_cache = {}
        self._buf = {}

    def _pos(self, n_side, dev):
        if n_side not in self._pos_cache:
            pe = self._sd["image_tokenizer.model.embeddings.position_embeddings"].numpy()
            self._pos_cache[n_side] = _f32(interpolate_pos_embedding(pe, n_side, self.pos_embed_mode), dev)
        return self._pos_cache[n_side]

    # ------------------------------------------------------------------ forward
    @staticmethod
    def _token_stride(T, B):
        """Rows from one image's tokens to the next in a stacked batch: T itself for one image, else T rounded up to 8 so that
        every image's rows (and its V^T columns) start 16-byte aligned; the pad rows carry finite values nobody reads."""
        return T if B == 1 else ((T + 7) // 8) * 8

    def image_tokens(self, image_hwc):
        """DINOSingleImageTokenizer.forward (tokenizers/image.py:41-60) for one [S,S,3] fp32 device image, or for a LIST of B
        such images in one pass (the reference's batch dimension, system.py:82-99): every Linear is one launch over the stacked
        token rows, the attention one launch over B x heads -> (ctx bf16 [rows, H] for cross attention, ctx fp32 [rows, H]);
        image b's T tokens are rows b*Ts .. b*Ts+T-1, Ts = _token_stride(T, B)."""
        imgs = list(image_hwc) if isinstance(image_hwc, (list, tuple)) else [image_hwc]
        B = len(imgs)
        v, w = self.cfg["image_tokenizer"], self._w
        H, P, nh = v["hidden_size"], v["patch_size"], v["num_attention_heads"]
        S = imgs[0].shape[0]
        if any(tuple(im.shape) != tuple(imgs[0].shape) for im in imgs):
            raise ValueError("TSR.image_tokens: the images of a batch must have one size")
        n_side = S // P
        npatch = n_side * n_side
        T = npatch + 1
        Ts = self._token_stride(T, B)
        M = B * Ts
        ldt = ((((B - 1) * Ts + ((T + 63) // 64) * 64) + 63) // 64) * 64   # V^T columns: image b at column b*Ts
        patches = self._b("patches", (B * npatch, 3 * P * P), self.adt)
        for b, im in enumerate(imgs):
            ops.vit_patchify(im, P, IMAGE_MEAN, IMAGE_STD, patches[b * npatch:(b + 1) * npatch])
        pout = self._b("patch_out", (B * npatch, H), torch.float32)
        self._gemm(patches, w["patch_w"], bias=w["patch_b"], out_f32=pout)
        h = self._b("vit_h", (M, H), torch.float32, zero=True)
        pos = self._pos(n_side, imgs[0].device)
        for b in range(B):
            ops.vit_assemble(pout[b * npatch:(b + 1) * npatch], w["cls"], pos, h[b * Ts:b * Ts + T])
            if Ts > T:   # the pad rows of the stacked stream take part in every Linear: start them from zero on every call, or
                h[b * Ts + T:(b + 1) * Ts].zero_()   # they drift without bound in a long-running process
        st = self._stream_state("vit", h)     # the residual stream + its bf16 copy + slice statistics (LayerNorm fold)
        self._stats_of(st)                    # rows that do not come out of a GEMM: one small kernel
        qk = self._b("vit_qk", (M, 2 * H), self.adt)
        vt = self._b("vit_vt", (H, ldt), self.adt, zero=True)
        if self.l3p:   # read only by the next Linear: written as limbs by their producers
            att, ff = self._lt("vit_att", M, H), self._lt("vit_ff", M, v["intermediate_size"])
        else:
            att = self._b("vit_att", (M, H), self.adt, zero=True)
            ff = self._b("vit_ff", (M, v["intermediate_size"]), self.adt)
        eps = v["layer_norm_eps"]
        for L in w["vit"]:
            self._ln_gemm(st, L, "qkv_w", eps, out_bf16=qk, out_t=vt, n_split=2 * H)  # Q|K token-major, V^T
            self._attn(qk[:, :H], qk[:, H:], vt, att, T, T, nh, self._attn_scale(1.0 / math.sqrt(H // nh)),
                       B, Ts * 2 * H, Ts * 2 * H, Ts, Ts * H)
            self._res_gemm(st, att, L["o_w"], L["o_b"])
            self._ln_gemm(st, L, "f1_w", eps, out_bf16=ff, epilogue=_lib.EPI_GELU)
            self._res_gemm(st, ff, L["f2_w"], L["f2_b"])
        ctx32 = self._b("ctx32", (M, H), torch.float32)
        if self.precision == "bf16":
            ctx = self._b("ctx", (M, H), BF16)
            ops.layernorm(h, w["vit_ln_w"], w["vit_ln_b"], eps, y=ctx, y_f32=ctx32)
        elif self.l3p:   # the cross-attention K / V projection reads the tokens as limbs
            ctx = self._lt("ctx", M, H)
            ops.layernorm(h, w["vit_ln_w"], w["vit_ln_b"], eps, y_lt=ctx, y_f32=ctx32)
        else:
            ops.layernorm(h, w["vit_ln_w"], w["vit_ln_b"], eps, y_f32=ctx32)
            ctx = ctx32
        return ctx, ctx32

    def _attn_scale(self, scale):
        """bf16 mode: the query projections carry scale * log2(e) (ln_linear q_scale) -> sculpt_attention_bf16_prescaled (scale None)."""
        return None if self.precision == "bf16" else scale

    def _self_attention(self, st, L, batch=1):
        """h += attn1(LN1(h)) of one BasicTransformerBlock (basic_transformer_block.py:149-167); h holds `batch` images' tokens
        stacked (attention.py:629-631 with its leading batch dimension)."""
        b = self.cfg["backbone"]
        nh, hd = b["num_attention_heads"], b["attention_head_dim"]
        D, M = nh * hd, st["h"].shape[0]
        T = M // batch
        # V^T columns: entry b is read at column b * T for round64(T) columns (the same formula as ldc / ldt below: ADVICE r4)
        Mp = ((((batch - 1) * T + ((T + 63) // 64) * 64) + 63) // 64) * 64
        qk = self._b("bb_qk", (M, 2 * D), self.adt)
        vt = self._b("bb_vt", (D, Mp), self.adt, zero=True)
        att = self._lt("bb_att", M, D) if self.l3p else self._b("bb_att", (M, D), self.adt)
        self._ln_gemm(st, L, "sa_qkv", 1e-5, out_bf16=qk, out_t=vt, n_split=2 * D)  # one launch: Q|K token-major, V^T
        self._attn(qk[:, :D], qk[:, D:], vt, att, T, T, nh, self._attn_scale(1.0 / math.sqrt(hd)),
                   batch, T * 2 * D, T * 2 * D, T, T * D)
        self._res_gemm(st, att, L["sa_o"], L["sa_ob"])

    def _run_blocks(self, st, ctx: torch.Tensor, first_self_attention_done: bool = False, batch: int = 1, ctx_tokens=None):
        """All BasicTransformerBlocks on the fp32 residual stream h [batch*T, D] (updated in place).  ctx [rows, cross_dim]: the
        image tokens, image b's ctx_tokens rows starting at row b * (rows // batch) (image_tokens' stacking)."""
        b, w = self.cfg["backbone"], self._w
        nh, hd = b["num_attention_heads"], b["attention_head_dim"]
        D = nh * hd
        if self.l3p and not isinstance(ctx, ops.Limbs):   # tokens handed over as a plain fp32 matrix (backbone_tokens): split here
            ctx = ops.Limbs.of(ctx.to(torch.float32).contiguous(), fmt=self.limb_format)
        M, Mc = st["h"].shape[0], (ctx.rows if isinstance(ctx, ops.Limbs) else ctx.shape[0])
        T, Ts = M // batch, Mc // batch
        Tc = Ts if ctx_tokens is None else ctx_tokens
        ldc = ((((batch - 1) * Ts + ((Tc + 63) // 64) * 64) + 63) // 64) * 64
        q = self._b("bb_q", (M, D), self.adt)
        nL = len(w["blocks"])
        ck_all = self._b("bb_ck", (Mc, nL * D), self.adt)
        cvt_all = self._b("bb_cvt", (nL * D, ldc), self.adt, zero=True)
        self._gemm(ctx, w["ca_kv_all"], out_bf16=ck_all, out_t=cvt_all, n_split=nL * D, M=Mc)
        if self.l3p:
            att, ff = self._lt("bb_att", M, D), self._lt("bb_ff", M, 4 * D)
        else:
            att = self._b("bb_att", (M, D), self.adt)
            ff = self._b("bb_ff", (M, 4 * D), self.adt)
        scale = self._attn_scale(1.0 / math.sqrt(hd))
        for li, L in enumerate(w["blocks"]):
            ck, cvt = ck_all[:, li * D:(li + 1) * D], cvt_all[li * D:(li + 1) * D]
            if li > 0 or not first_self_attention_done:
                self._self_attention(st, L, batch)
            self._ln_gemm(st, L, "ca_q", 1e-5, out_bf16=q)
            self._attn(q, ck, cvt, att, T, Tc, nh, scale, batch, T * D, Ts * nL * D, Ts, T * D)
            self._res_gemm(st, att, L["ca_o"], L["ca_ob"])
            self._ln_gemm(st, L, "ff1", 1e-5, out_bf16=ff, epilogue=_lib.EPI_GEGLU)
            self._res_gemm(st, ff, L["ff2"], L["ff2_b"])
        return st

    def _backbone_head(self):
        """The part of Transformer1D.forward that does not depend on the image: GroupNorm of the learned triplane tokens,
        proj_in (transformer_1d.py:181-189) and the first block's self-attention."""
        b, w = self.cfg["backbone"], self._w
        D = b["num_attention_heads"] * b["attention_head_dim"]
        T = w["emb_ct"].shape[1]
        xn = self._b("bb_xn", (T, w["emb_ct"].shape[0]), self.adt)
        stats = self._b("gn_stats", (2 * b["norm_num_groups"],), torch.float32)
        ops.groupnorm_tokens(w["emb_ct"], b["norm_num_groups"], w["gn_w"], w["gn_b"], 1e-6, xn, stats)
        h = self._b("bb_h", (T, D), torch.float32)
        st = self._stream_state("bb", h)
        if self.precision == "bf16":
            ops.gemm(xn, w["pin_w"], bias=w["pin_b"], out_f32=h, out_bf16=st["hb"], stats_out=st["stats"])
        else:
            self._gemm(xn, w["pin_w"], bias=w["pin_b"], out_f32=h)
        return st

    def _backbone_tail(self, st, batch=1):
        w = self._w
        C = self.cfg["tokenizer"]["num_channels"]
        h = st["h"]
        M, D = h.shape
        res = w["emb_tc"]                       # the learned tokens: the same residual rows for every image of a batch
        if batch > 1:
            key = "emb_tc_x%d" % batch
            if key not in w:
                w[key] = res.repeat(batch, 1).contiguous()
            res = w[key]
        out = self._b("bb_out", (M, C), torch.float32)
        if self.precision == "bf16":
            outb = self._b("bb_outb", (M, C), BF16)
            ops.gemm(st["hb"], w["pout_w"], bias=w["pout_b"], residual=res, out_f32=out, out_bf16=outb)  # hb = bf16(h)
        else:
            ops.gemm_f32(h, w["pout_w"], bias=w["pout_b"], residual=res, out=out, l3=self.l3)
            outb = out
        return out, outb

    def _broadcast_state(self, st1, batch):
        """The residual stream of ONE image (the image-independent head of the backbone) copied into a stacked state of `batch`."""
        T, D = st1["h"].shape
        st = self._stream_state("bb", self._b("bb_h", (batch * T, D), torch.float32))
        st["h"].view(batch, T, D).copy_(st1["h"][None])
        if self.precision == "bf16":
            st["hb"].view(batch, T, D).copy_(st1["hb"][None])
            st["stats"].view(-1, batch, T, 2).copy_(st1["stats"][:, None])
        return st

    def encode_images(self, images):
        """encode_image for a batch (system.py:82-115 with batch_size = len(images)): ONE pass of the tokenizer and the backbone
        over the stacked token rows of all images -- weight panels are read once per batch, every Linear is one launch of
        B x 3072 (B x 1025) rows, every attention one launch over B x heads.  The image-independent head of the backbone
        (GroupNorm, proj_in, first self-attention) is evaluated once, on a second stream under the tokenizer, and copied into
        the B streams.  Returns the output tokens of all images stacked: (fp32 [B*3*S*S, C], bf16 copy)."""
        B = len(images)
        if B == 1:
            return self.encode_image(images[0])
        main = torch.cuda.current_stream(self.device)
        last_tok = getattr(self, "_tok_last", None)
        if last_tok is not None:
            main.wait_event(last_tok)
        side = getattr(self, "_side_stream", None)
        if side is None:
            side = self._side_stream = torch.cuda.Stream(self.device)
        fork, join = torch.cuda.Event(), torch.cuda.Event()
        fork.record(main)
        with torch.cuda.stream(side):
            side.wait_event(fork)
            st1 = self._backbone_head()
            self._self_attention(st1, self._w["blocks"][0])
            st = self._broadcast_state(st1, B)
            join.record(side)
        # Every GEMM of the pass picks the tile form ONE image takes (ops.single_image_tiles): the stacked pass then accumulates every
        # output and every LayerNorm slice statistic in the single-image order, and each image gets the scene code of its own pass,
        # bit for bit (test_full_size_batched_forward_equals_single_image_passes) -- which is what lets TSR.run batch by default.
        T = self.cfg["image_tokenizer"]
        n_side = images[0].shape[0] // T["patch_size"]
        with ops.single_image_tiles(n_side * n_side + 1 if self.batch_exact else 0):
            ctx, _ = self.image_tokens(list(images))
        main.wait_event(join)
        with ops.single_image_tiles(3 * self.cfg["tokenizer"]["plane_size"] ** 2 if self.batch_exact else 0):
            st = self._run_blocks(st, ctx, first_self_attention_done=True, batch=B, ctx_tokens=n_side * n_side + 1)
            return self._backbone_tail(st, B)

    def backbone_tokens(self, ctx: torch.Tensor):
        """Triplane1DTokenizer + Transformer1D for one image; ctx bf16 [Tc, cross_dim].
        Returns the output tokens token-major: fp32 [3*S*S, C] (+ bf16 copy)."""
        st = self._run_blocks(self._backbone_head(), ctx)
        return self._backbone_tail(st)

    def encode_image(self, image_hwc: torch.Tensor):
        """image_tokens + backbone_tokens for one [S,S,3] fp32 device image, with the image-independent head of the backbone
        (GroupNorm, proj_in, first self-attention: 6 launches) issued on a second HIP stream so that it runs under the ViT,
        whose 1025-token launches leave most CUs idle.  Same kernels on the same operands as the sequential calls:
        bit-identical tokens."""
        main = torch.cuda.current_stream(self.device)
        last_tok = getattr(self, "_tok_last", None)
        if last_tok is not None:  # tokens_async shares the tokenizer's work buffers: finish it first
            main.wait_event(last_tok)
        side = getattr(self, "_side_stream", None)
        if side is None:
            side = self._side_stream = torch.cuda.Stream(self.device)
        fork, join = torch.cuda.Event(), torch.cuda.Event()
        fork.record(main)               # everything queued so far (the previous image's readers of these buffers) comes first
        with torch.cuda.stream(side):
            side.wait_event(fork)
            st = self._backbone_head()
            self._self_attention(st, self._w["blocks"][0])
            join.record(side)
        ctx, _ = self.image_tokens(image_hwc)
        main.wait_event(join)
        st = self._run_blocks(st, ctx, first_self_attention_done=True)
        return self._backbone_tail(st)

    def _preprocess(self, im) -> torch.Tensor:
        """ImagePreprocessor (tsr/utils.py:62-112): uint8 / PIL -> float / 255 on the host, then the antialiased bilinear resize
        to cond_image_size on the GPU (sculpt_resize_aa_bilinear)."""
        size = self.cfg["cond_image_size"]
        img = self._upload(_to_float_hwc(im)).contiguous()
        if img.shape[-1] != 3:
            raise ValueError("TSR.forward expects RGB images (composite RGBA on grey first, preprocessing.py:122)")
        if img.shape[0] != size or img.shape[1] != size:
            img = ops.resize_aa_bilinear(img, size)
        return img

    def tokens_async(self, image) -> PendingTokens:
        """Upload + ImagePreprocessor + DINO image tokenizer of ONE image, queued on a second HIP stream.  The tokenizer is
        ~1.0 ms of launches of 72 - 200 workgroups that leave most CUs idle; queued here for image i + 1 before the backbone,
        density grid and marching cubes of image i are queued on the current stream, it runs beside them (11.40 -> 11.08 ms per
        image, tools/try_vit_prefetch.py) -- same kernels on the same operands, bit-identical tokens.  The tokens land in one
        of two slots; a slot is rewritten only after the backbone that read it has been through (forward_tokens)."""
        if self._w is None:
            raise _lib.SculptError("TSR: call load_state_dict() and to(device) before tokens_async()")
        tok = getattr(self, "_tok_stream", None)
        if tok is None:
            tok = self._tok_stream = torch.cuda.Stream(self.device)
            self._tok_ring = {"i": 0, "slots": [{"ctx": None, "consumed": None} for _ in range(2)]}
        slot = self._tok_ring["slots"][self._tok_ring["i"] % 2]
        self._tok_ring["i"] += 1
        main = torch.cuda.current_stream(self.device)
        fork = torch.cuda.Event()
        fork.record(main)  # a serial forward() queued earlier uses the same tokenizer buffers on the current stream
        with torch.cuda.stream(tok), torch.no_grad():
            tok.wait_event(fork)
            if slot["consumed"] is not None:
                tok.wait_event(slot["consumed"])
            ctx, _ = self.image_tokens(self._preprocess(image))
            if isinstance(ctx, ops.Limbs):   # three-limb mode, limbs once: the tokens travel as limbs
                if not isinstance(slot["ctx"], ops.Limbs) or (slot["ctx"].rows, slot["ctx"].cols, slot["ctx"].fmt) != (ctx.rows, ctx.cols, ctx.fmt):
                    slot["ctx"] = ops.Limbs(ctx.rows, ctx.cols, data=torch.empty_like(ctx.data), fmt=ctx.fmt)
                slot["ctx"].data.copy_(ctx.data)
            else:
                if not isinstance(slot["ctx"], torch.Tensor) or slot["ctx"].shape != ctx.shape or slot["ctx"].dtype != ctx.dtype:
                    slot["ctx"] = torch.empty_like(ctx)
                slot["ctx"].copy_(ctx)
            ready = torch.cuda.Event()
            ready.record(tok)
        self._tok_last = ready
        return PendingTokens(slot["ctx"], ready, slot, image)

    def forward_tokens(self, tokens: PendingTokens) -> torch.Tensor:
        """The rest of forward() for one image whose tokens come from tokens_async: backbone + upsampler on the current stream
        -> scene code fp32 [1, 3, 40, 64, 64]."""
        main = torch.cuda.current_stream(self.device)
        main.wait_event(tokens.ready)
        st = self._run_blocks(self._backbone_head(), tokens.ctx)
        consumed = torch.cuda.Event()
        consumed.record(main)
        tokens._slot["consumed"] = consumed
        _, outb = self._backbone_tail(st)
        out = self.scene_code(outb)[None]
        if self.precision == "fp16l2" and tokens.image is not None and not bool(torch.isfinite(out).all()):
            return self._range_fallback([tokens.image])   # forward()'s range fallback: straight to the three-limb twin
        return out

    def scene_code(self, tokens_bf16: torch.Tensor, batch: int = 1):
        """detokenize + TriplaneUpsampleNetwork: tokens [batch*3*S*S, C] -> planes fp32 [3, Co, 2S, 2S] ([batch, 3, ...] for batch > 1)."""
        w = self._w
        S = self.cfg["tokenizer"]["plane_size"]
        Co = self.cfg["post_processor"]["out_channels"]
        g = self._b("up_g", (tokens_bf16.shape[0], w["up_w"].shape[0]), torch.float32)
        self._gemm(tokens_bf16, w["up_w"], out_f32=g)
        planes = torch.empty((batch, 3, Co, 2 * S, 2 * S), dtype=torch.float32, device=tokens_bf16.device)
        T = tokens_bf16.shape[0] // batch
        for b in range(batch):
            ops.upsample_scatter(g[b * T:(b + 1) * T], w["up_b"], planes[b], S, Co)
        return planes if batch > 1 else planes[0]

    def _upload(self, t: torch.Tensor) -> torch.Tensor:
        """Host image -> HBM without blocking the host: a pageable source makes hipMemcpyAsync synchronous (the host waits for
        the stream to drain and for the copy), so it is first copied into one of three pinned staging buffers (a 3 MB host
        memcpy) and sent from there; a slot is reused once the copy that read it has completed."""
        if t.device.type != "cpu":
            return t.to(self.device, non_blocking=True)
        if t.is_pinned():
            return t.to(self.device, non_blocking=True)
        ring = getattr(self, "_stage", None)
        if ring is None:
            ring = self._stage = {"i": 0, "slots": [[None, None] for _ in range(3)]}
        slot = ring["slots"][ring["i"] % 3]
        ring["i"] += 1
        n = t.numel()
        if slot[1] is not None:
            slot[1].synchronize()
        if slot[0] is None or slot[0].numel() < n:
            slot[0] = torch.empty(max(n, 3 * 512 * 512), dtype=torch.float32, pin_memory=True)
        stage = slot[0][:n].view(t.shape)
        # a plain single-threaded memcpy: torch's CPU copy_ is an OpenMP parallel_for, and waking the (sleeping) worker threads of
        # a 256-core host once per image costs ~20 ms -- measured: 11.4 -> 31 ms per step
        np.copyto(stage.numpy(), t.numpy() if t.is_contiguous() else t.contiguous().numpy())
        d = stage.to(self.device, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record(torch.cuda.current_stream(self.device))
        return d

    def forward(self, image, device=None) -> torch.Tensor:
        """system.py:82-115: image(s) -> scene_codes fp32 [B, 3, 40, 64, 64] on the device."""
        if self._w is None:
            if device is not None and self._sd is not None:
                self.to(device)
            else:
                raise _lib.SculptError("TSR: call load_state_dict() and to(device) before forward()")
        images = [self._preprocess(im) for im in _as_image_list(image)]
        codes = []
        step = max(1, int(self.max_batch))
        for i in range(0, len(images), step):
            group = images[i:i + step]
            n_tok = 3 * self.cfg["tokenizer"]["plane_size"] ** 2   # a batch entry's rows / V^T columns must start 16-byte aligned
            if len(group) == 1 or n_tok % 8 or any(tuple(g.shape) != tuple(group[0].shape) for g in group):
                for im in group:
                    _, outb = self.encode_image(im)
                    codes.append(self.scene_code(outb)[None])
            else:   # the reference's batched pass (system.py:82-115)
                _, outb = self.encode_images(group)
                with ops.single_image_tiles(n_tok if self.batch_exact else 0):   # the upsampler's GEMM too (encode_images)
                    codes.append(self.scene_code(outb, len(group)))
        out = torch.cat(codes, 0) if len(codes) > 1 else codes[0]
        if self.precision == "fp16l2" and not bool(torch.isfinite(out).all()):
            # an fp16 limb overflowed (an activation of 65504 or more in magnitude) or the input was not finite: never hand on a
            # silently wrong scene code -- the same images go through a three-limb twin of this model ("bf16l3": the fp32 exponent
            # range), built on first need; range_fallbacks counts the calls that needed it
            return self._range_fallback(image)
        return out

    def _range_fallback(self, image):
        """fp16l2: the images of a call whose scene code came out non-finite, through the "bf16l3" twin of this model (built from the
        CURRENT weights on first need; load_state_dict drops it)."""
        self.range_fallbacks += 1
        twin = self._range_twin
        if twin is None:
            twin = self._range_twin = TSR(self.cfg, self.pos_embed_mode, precision="bf16l3", decoder_precision=self.decoder_precision,
                                          decoder_filter=self.decoder_filter)
            twin.load_state_dict(self._sd)
            twin.to(self.device)
        twin.max_batch = self.max_batch
        return twin.forward(image)

    __call__ = forward

    # ------------------------------------------------------------------ mesh extraction
    def set_marching_cubes_resolution(self, resolution: int):
        if self.isosurface_helper is not None and self.isosurface_helper.resolution == resolution:
            return
        self.isosurface_helper = MarchingCubeHelper(resolution)

    def extract_meshes(self, scene_codes, enable_texture=False, resolution: int = 256, threshold: float = 25.0,
                       x_range=None, density_events=None) -> List[Mesh]:
        """The arithmetic of system.py:171-200 without the Blender sink: returns device tensors.
        density_events: optional (start, stop) torch events recorded around the dense-grid launch (bench.py's live
        per-launch timing of the dominant kernel, on the stream it is launched on)."""
        self.set_marching_cubes_resolution(resolution)
        r = self.renderer.cfg.radius
        R = resolution
        out = []
        def mc(v, sign_planes=None):
            return ops.marching_cubes(v.view(R, R, R), 0.0, reference_order=True, vert_div=R - 1.0, vert_mul=r - (-r), vert_add=-r,
                                      sign_planes=sign_planes)

        mc.takes_sign_planes = True
        for scene_code in scene_codes:
            planes = scene_code.contiguous()
            dkw = dict(radius=r, density_bias=self.renderer.cfg.density_bias, out_add=-threshold)
            # density_act - threshold == -(-(density_act - threshold))  (system.py:184, isosurface.py:45)
            if self._filter_applies(planes, R, threshold):
                v_pos, t_pos_idx = self._extract_filtered(planes, R, mc, dkw, density_events)
                color = None
                if enable_texture:
                    color = self.renderer.query_triplane(self.decoder, v_pos, planes)["color"]
                out.append(Mesh(v_pos, t_pos_idx, color))
                continue
            vol = ops.density_grid(planes, self.decoder, R, precision=self.decoder_precision, events=density_events, **dkw)
            v_pos, t_pos_idx = mc(vol)   # (both decoder modes have the fp32 range: a NaN here is a NaN of the model)
            color = None
            if enable_texture:
                color = self.renderer.query_triplane(self.decoder, v_pos, planes)["color"]
            out.append(Mesh(v_pos, t_pos_idx, color))
        return out

    # -- the two-pass ("filtered") density grid: calibration, guard, fallback (csrc/density_filter.hip)
    FILTER_SAFETY = 8.0        # margin = FILTER_SAFETY x the largest coarse error of the calibration probe
    FILTER_GUARD = 1.0 / 3.0   # a call whose re-evaluated points show a coarse error above FILTER_GUARD x margin is redone in full
    FILTER_PROBE = 64          # lattice of the calibration probe
    FILTER_MAX_MARGIN = 2.0    # a model whose coarse pass is this far off (log units of density) gains nothing: filter off

    def _filter_applies(self, planes, R, threshold):
        return (self.decoder_filter and self.decoder_precision == "bf16l3" and self.filter_info["usable"] and threshold > 0
                and self.decoder.n_hidden >= 1 and 32 <= R <= 1024)

    def calibrate_decoder_filter(self, planes):
        """Margin of the two-pass density grid for this model, measured on `planes` (one scene code [3, C, H, W]): every point of a
        64^3 lattice through both the one-product pass and the three-limb arithmetic (SCULPT_FILTER_MARK_ALL); margin =
        FILTER_SAFETY x the largest |log d~ - log d|.  IEEE-half operands when the probe stays finite, bf16 otherwise; a model
        for which neither gives a usable margin runs unfiltered."""
        info, r = self.filter_info, self.renderer.cfg.radius
        info["calibrations"] += 1
        for coarse in ("fp16", "bf16"):
            _, st = ops.density_grid_filtered(planes, self.decoder, self.FILTER_PROBE, 0.0, radius=r,
                                              density_bias=self.renderer.cfg.density_bias, out_add=0.0, coarse=coarse, mark_all=True)
            s = ops.filter_stats(st)
            if s["n_nonfinite"] == 0 and np.isfinite(s["max_err"]):
                margin = max(self.FILTER_SAFETY * s["max_err"], 1e-3)
                if margin <= self.FILTER_MAX_MARGIN:
                    info.update(margin=margin, coarse=coarse, probe_max_err=s["max_err"])
                    return info
        info.update(usable=False, margin=None)
        return info

    def _extract_filtered(self, planes, R, mc, dkw, density_events=None, x_begin=0, x_end=None):
        """The two-pass grid of lattice planes [x_begin, x_end) + mc(volume) (marching cubes: its count read-back waits for the
        stream, so the statistics of the grid call have landed when it returns), under the run-time guard."""
        info = self.filter_info
        dkw = dict(dkw, x_begin=x_begin, x_end=x_end)
        if info["margin"] is None:
            self.calibrate_decoder_filter(planes)
            if not info["usable"]:
                return mc(ops.density_grid(planes, self.decoder, R, precision="bf16l3", events=density_events, **dkw))
        host = getattr(self, "_filter_stats_host", None)
        if host is None:
            host = self._filter_stats_host = torch.zeros(ops.FILTER_STATS_WORDS, dtype=torch.int32).pin_memory()
        vol, _ = ops.density_grid_filtered(planes, self.decoder, R, info["margin"], coarse=info["coarse"], events=density_events,
                                           stats_host=host, **dkw)
        err = None
        # a whole grid (not a slab): marching cubes takes its cell signs from the grid's sign planes and reads the volume only
        # where a cell is active; SCULPT_MC_SIGN_PLANES=0: the plain count phase over the whole volume (A/B)
        signs = None
        if x_begin == 0 and (x_end is None or x_end == R) and getattr(mc, "takes_sign_planes", False) and _MC_SIGN_PLANES:
            signs = ops.filter_sign_planes(R, planes.device)
        try:
            # waits for the stream (sculpt_mc_count reads its counts back): the statistics have landed with it
            mesh = mc(vol) if signs is None else mc(vol, signs)
        except Exception as e:  # an empty / out-of-range surface raises in both evaluations; checked below before it is believed
            mesh, err = None, e
        st = ops.filter_stats(host)
        info["last"] = st
        seen = ops.filter_guard_error(st)
        if seen <= self.FILTER_GUARD * info["margin"]:
            info["filtered"] += 1
            if err is not None:
                raise err
            return mesh
        # the coarse pass was further off than the calibration allows for: this grid in full, and a margin that covers what was seen
        info["fallbacks"] += 1
        info["margin"] = None
        vol = ops.density_grid(planes, self.decoder, R, precision="bf16l3", out=vol, **dkw)
        mesh = mc(vol)
        self.calibrate_decoder_filter(planes)
        if info["usable"] and info["margin"] is not None and np.isfinite(seen):
            info["margin"] = min(max(info["margin"], self.FILTER_SAFETY * seen), self.FILTER_MAX_MARGIN)
        return mesh

    def extract_mesh_sharded(self, scene_code, resolution: int = 512, threshold: float = 25.0, enable_texture=False):
        """BASELINE config 5: the voxel grid of ONE image split into slabs along the slowest lattice axis over
        the ranks of the default process group (RCCL over xGMI), one padded all-gather of the per-slab
        triangles, identical result on every rank and identical to the single-GPU mesh
        (sculptmate_amd/slab.py).  Without a process group the slabs run one after the other here."""
        import torch.distributed as dist

        from .. import slab

        r = self.renderer.cfg.radius
        planes = scene_code.contiguous()
        kw = dict(radius=r, density_bias=self.renderer.cfg.density_bias, threshold=threshold,
                  precision=self.decoder_precision)
        if self._filter_applies(planes, resolution, threshold):
            # every slab through the two-pass grid (cells inside the slab's planes only: the halo plane is part of the slab)
            dkw = dict(radius=r, density_bias=self.renderer.cfg.density_bias, out_add=-threshold)
            kw["run"] = lambda x0, x1, mc: self._extract_filtered(planes, resolution, mc, dkw, None, x0, x1)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            part = slab.extract_slab(planes, self.decoder, resolution, dist.get_rank(), dist.get_world_size(), **kw)
            v_pos, t_pos_idx = slab.gather_and_assemble(part, planes.device)
        else:
            v_pos, t_pos_idx = slab.extract_mesh_slabs_local(planes, self.decoder, resolution, 1, **kw)
        color = None
        if enable_texture:
            color = self.renderer.query_triplane(self.decoder, v_pos, planes)["color"]
        return Mesh(v_pos, t_pos_idx, color)

    def extract_mesh(self, scene_codes, enable_texture=False, mesh_name="NewMesh", resolution: int = 256,
                     threshold: float = 25.0):
        """system.py:171-200: same signature; pushes each mesh into the sink (Blender when `bpy` is
        importable, exactly like the reference's import_obj_blender) and also returns the meshes."""
        meshes = self.extract_meshes(scene_codes, enable_texture, resolution, threshold)
        sink = self.mesh_sink or _default_sink()
        for m in meshes:
            if sink is not None:
                sink(m.vertices.cpu().numpy(), m.faces.cpu().numpy(),
                     None if m.vertex_colors is None else m.vertex_colors.cpu().numpy(), mesh_name)
        return meshes

    def run_async(self, image, mc_resolution: int = 256, threshold: float = 25.0, enable_texture: bool = False, tokens=None):
        """One host image -> PendingMesh.  The image goes host -> HBM, the forward and the mesh extraction are queued on
        the current stream, and the mesh (the reference's `.cpu().numpy()` at system.py:200) is copied device -> pinned
        host memory on a separate copy stream, so the copy of mesh i runs under the kernels of image i + 1.
        PendingMesh.result() waits for that copy only.  tokens: the PendingTokens of THIS image from tokens_async (then
        `image` is not touched again): a caller with several images queues the tokens of image i + 1 before this call (run)."""
        with torch.no_grad():
            codes = self.forward([image], self.device) if tokens is None else self.forward_tokens(tokens)
            m = self.extract_meshes(codes, enable_texture, mc_resolution, threshold)[0]
        return self._mesh_to_host_async(m)

    def _mesh_to_host_async(self, m) -> PendingMesh:
        """Device mesh -> PendingMesh: vertices / faces / colours copied into pinned host buffers on the copy stream."""
        main = torch.cuda.current_stream(self.device)
        copy = getattr(self, "_copy_stream", None)
        if copy is None:
            copy = self._copy_stream = torch.cuda.Stream(self.device)
        ready = torch.cuda.Event()
        ready.record(main)
        pool = getattr(self, "_pin_pool", None)
        if pool is None:
            pool = self._pin_pool = _PinnedPool()
        host, leases = [], []
        with torch.cuda.stream(copy):
            copy.wait_event(ready)
            for t in (m.vertices, m.faces, m.vertex_colors):
                if t is None:
                    host.append(None)
                    continue
                lease, h = pool.take(t.shape, t.dtype)
                h.copy_(t, non_blocking=True)
                t.record_stream(copy)  # the device block must not be reused before the copy has read it
                host.append(h)
                leases.append(lease)
            done = torch.cuda.Event()
            done.record(copy)
        return PendingMesh(host, done, tuple(leases))

    def run(self, images, mc_resolution: int = 256, threshold: float = 25.0, enable_texture: bool = False, batch=None) -> List[Mesh]:
        """Headless entry point: images -> list of Mesh with host (NumPy) arrays; the device -> host copy of mesh i overlaps the
        kernels that follow it.
        batch (images per transformer pass; None = the default): in the bf16 mode a stacked pass of several images gives each
        image the scene code of its own single-image pass BIT FOR BIT (every GEMM keeps the single-image tile form,
        ops.single_image_tiles), so several images run RUN_BATCH = 8 per pass by default -- the reference's batched forward
        (system.py:82-115), 3.5-3.9 instead of 5.2 ms of transformer per image -- and the meshes are those of one-at-a-time calls
        (test_run_batches_by_default_and_returns_the_serial_meshes).  The limb modes default to one image per pass with the
        tokenizer look-ahead (run_pipelined).  batch=1 forces that everywhere."""
        images = _as_image_list(images)
        if batch is None:
            batch = self.RUN_BATCH if (self.precision == "bf16" and len(images) >= 2) else 1
        if batch <= 1 or len(images) < 2:
            return [p.result() for p in self.run_pipelined(images, mc_resolution, threshold, enable_texture)]
        return [p.result() for p in self.run_batched(images, batch, mc_resolution, threshold, enable_texture)]

    RUN_BATCH = 8   # images per transformer pass of TSR.run in the bf16 mode (4: 139.7, 8: 143.5 meshes/s device to device)

    def run_batched(self, images, batch: int = 4, mc_resolution: int = 256, threshold: float = 25.0, enable_texture: bool = False):
        """images (host or device) -> list of PendingMesh through batched forward passes of `batch` images each."""
        images = list(images)
        pending = []
        keep = self.max_batch
        try:
            self.max_batch = max(1, int(batch))
            for i in range(0, len(images), self.max_batch):
                with torch.no_grad():
                    codes = self.forward(images[i:i + self.max_batch], self.device)
                    for m in self.extract_meshes(codes, enable_texture, mc_resolution, threshold):
                        pending.append(self._mesh_to_host_async(m))
        finally:
            self.max_batch = keep
        return pending

    def run_pipelined(self, images, mc_resolution: int = 256, threshold: float = 25.0, enable_texture: bool = False):
        """images (host or device) -> list of PendingMesh, with the tokenizer of image i + 1 queued beside the backbone /
        density grid / marching cubes of image i (tokens_async) and the device -> host copy of mesh i under image i + 1."""
        images = list(images)
        if len(images) < 2:
            return [self.run_async(im, mc_resolution, threshold, enable_texture) for im in images]
        pending, nxt = [], self.tokens_async(images[0])
        for i, im in enumerate(images):
            cur = nxt
            nxt = self.tokens_async(images[i + 1]) if i + 1 < len(images) else None
            pending.append(self.run_async(im, mc_resolution, threshold, enable_texture, tokens=cur))
        return pending


def _run_sharded(self, images, mc_resolution: int = 256, threshold: float = 25.0, enable_texture: bool = False, **kw):
    """A batch over the ranks of the default process group, one image per GPU at a time (sculptmate_amd/batch.py): returns
    ({index: Mesh} for this rank's images, [(index, rank, n_vertices, n_faces)] for all of them)."""
    from .. import batch

    with torch.no_grad():
        return batch.run_sharded(self, images, mc_resolution, threshold, enable_texture, **kw)


TSR.run_sharded = _run_sharded


def _as_image_list(image):
    if isinstance(image, (np.ndarray, torch.Tensor)) and image.ndim == 4:
        return [image[i] for i in range(image.shape[0])]
    return image if isinstance(image, (list, tuple)) else [image]


def _to_float_hwc(image):
    """ImagePreprocessor.convert_and_resize's conversion step (tsr/utils.py:68-77), without the resize."""
    if not isinstance(image, (np.ndarray, torch.Tensor)):  # PIL.Image
        image = np.array(image)
    if isinstance(image, np.ndarray):
        image = torch.from_numpy(image.astype(np.float32) / 255.0) if image.dtype == np.uint8 else torch.from_numpy(image)
    return image.to(torch.float32)


def _default_sink():
    try:
        import bpy  # noqa: F401
    except Exception:
        return None
    from .blender_sink import import_obj_blender

    return import_obj_blender


def load_config(yaml_path: str, vit_json_path: Optional[str] = None):
    """checkpoints/config.yaml (+ config.json for the ViT) -> the dict TSR takes."""
    import json

    import yaml

    with open(yaml_path) as f:
        y = yaml.safe_load(f)
    cfg = {k: (dict(v) if isinstance(v, dict) else v) for k, v in DEFAULT_CFG.items()}
    cfg["cond_image_size"] = y.get("cond_image_size", cfg["cond_image_size"])
    tok = y.get("tokenizer", {})
    cfg["tokenizer"].update({k: tok[k] for k in ("plane_size", "num_channels") if k in tok})
    bb = dict(y.get("backbone", {}))
    if isinstance(bb.get("in_channels"), str):  # "${tokenizer.num_channels}"
        bb["in_channels"] = cfg["tokenizer"]["num_channels"]
    cfg["backbone"].update({k: bb[k] for k in cfg["backbone"] if k in bb})
    cfg["post_processor"].update(y.get("post_processor", {}))
    cfg["decoder"].update(y.get("decoder", {}))
    cfg["renderer"].update(y.get("renderer", {}))
    if vit_json_path and os.path.exists(vit_json_path):
        with open(vit_json_path) as f:
            j = json.load(f)
        for k in cfg["image_tokenizer"]:
            if k in j:
                cfg["image_tokenizer"][k] = j[k]
    return cfg
