"""StableFast-3D's two estimators on the MI355X, with the reference's class names, config keys and output dicts:

  ClipBasedHeadEstimator   StableFast/sf3d/models/image_estimator/clip_based_estimator.py:26-168 -- roughness / metallic
                           of the whole object from the conditioning image: rgb_cond * mask_cond resized to 224^2, the
                           CLIP ViT-B/32 visual tower (open_clip "ViT-B-32" / laion2b_s34b_b79k, :43-46), two small MLP
                           heads per quantity -> Beta(softplus(d1 + bias), softplus(d2 + bias)) read out at its mode.
  MultiHeadEstimator       StableFast/sf3d/models/global_estimator/multi_head_estimator.py:23-118 -- illumination
                           (spherical-Gaussian amplitudes) from the un-upsampled triplane: two stride-2 3x3 convolutions,
                           max pool, MLP heads.  Only run when run_image(estimate_illumination=True).

The tower is the ViT kernel pipeline of the TripoSR tokenizer with CLIP's wiring (bias-free 32x32 patch GEMM, ln_pre,
packed in_proj, GELU MLP, ln_post on the class token, projection); 50 tokens, so it is a chain of ~90 small launches
(latency-bound, ~1 ms).  The heads are [B, 512] x [512, 512] products: exact-fp32 kernel.  The last step -- softplus and the
Beta mode of two numbers per quantity -- is host arithmetic on the four scalars that have to reach the host anyway
(the reference calls .item() on them, sf3d/system.py:474-475).

Weights keep the reference's state_dict names under `image_estimator.` / `global_estimator.`; the text tower that
open_clip's checkpoint carries along (`image_estimator.model.transformer.*`, token_embedding, ...) is ignored.
"""
import math

import numpy as np
import torch

from .. import _lib, ops
from ..engine import KernelEngine

BF16 = torch.bfloat16
OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)   # open_clip.constants (clip_based_estimator.py:101-104)
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)

CLIP_VIT_B_32 = dict(image_size=224, patch_size=32, width=768, layers=12, heads=12, mlp=3072, embed_dim=512, eps=1e-5)

_HEAD = dict(out_channels=1, n_hidden_layers=3, output_activation="linear", add_to_decoder_features=True, output_bias=1.0,
             shape=(-1, 1, 1))
IMAGE_ESTIMATOR_CFG = dict(model="ViT-B-32", pretrain="laion2b_s34b_b79k", distribution="beta", distribution_eval="mode",
                           activation="relu", hidden_features=512, clip=CLIP_VIT_B_32,
                           heads=(dict(name="roughness", **_HEAD), dict(name="metallic", **_HEAD)))
GLOBAL_ESTIMATOR_CFG = dict(triplane_features=1024, n_layers=2, hidden_features=512, activation="relu", pool="max",
                            heads=(dict(name="sg_amplitudes", out_channels=24, n_hidden_layers=3, output_activation="softplus",
                                        output_bias=1.0, add_to_decoder_features=False, shape=(-1, 24, 1)),))


def _head_cfg(h):
    """HeadSpec defaults (clip_based_estimator.py:15-23, multi_head_estimator.py:12-20)."""
    return dict(name=h["name"], out_channels=h["out_channels"], n_hidden_layers=h["n_hidden_layers"],
                output_activation=h.get("output_activation"), output_bias=float(h.get("output_bias", 0.0)),
                add_to_decoder_features=bool(h.get("add_to_decoder_features", False)),
                shape=None if h.get("shape") is None else tuple(h["shape"]))


def clip_param_spec(clip, prefix="image_estimator.model.visual."):
    W, P, M, E = clip["width"], clip["patch_size"], clip["mlp"], clip["embed_dim"]
    n_tok = (clip["image_size"] // P) ** 2 + 1
    spec = {prefix + "class_embedding": (W,), prefix + "positional_embedding": (n_tok, W), prefix + "proj": (W, E),
            prefix + "conv1.weight": (W, 3, P, P)}
    for ln in ("ln_pre", "ln_post"):
        spec[prefix + ln + ".weight"] = (W,)
        spec[prefix + ln + ".bias"] = (W,)
    for i in range(clip["layers"]):
        q = prefix + "transformer.resblocks.%d." % i
        spec[q + "attn.in_proj_weight"], spec[q + "attn.in_proj_bias"] = (3 * W, W), (3 * W,)
        spec[q + "attn.out_proj.weight"], spec[q + "attn.out_proj.bias"] = (W, W), (W,)
        spec[q + "mlp.c_fc.weight"], spec[q + "mlp.c_fc.bias"] = (M, W), (M,)
        spec[q + "mlp.c_proj.weight"], spec[q + "mlp.c_proj.bias"] = (W, M), (W,)
        for ln in ("ln_1", "ln_2"):
            spec[q + ln + ".weight"] = (W,)
            spec[q + ln + ".bias"] = (W,)
    return spec


def image_estimator_param_spec(cfg, prefix="image_estimator."):
    spec = clip_param_spec(cfg["clip"], prefix + "model.visual.")
    HF = cfg["hidden_features"]
    for h in cfg["heads"]:
        q = prefix + "heads.%s." % h["name"]
        for j in range(h["n_hidden_layers"]):                     # shared trunk: Sequential(Linear, act, ...) -> 0, 2, 4
            spec[q + "0.%d.weight" % (2 * j)], spec[q + "0.%d.bias" % (2 * j)] = (HF, HF), (HF,)
        for br in (1, 2):                                         # two branches: Sequential(Linear, act, Linear(HF, 1))
            spec[q + "%d.0.weight" % br], spec[q + "%d.0.bias" % br] = (HF, HF), (HF,)
            spec[q + "%d.2.weight" % br], spec[q + "%d.2.bias" % br] = (1, HF), (1,)
    return spec


def global_estimator_param_spec(cfg, prefix="global_estimator."):
    spec = {}
    cur, HF = 3 * cfg["triplane_features"], cfg["hidden_features"]
    for i in range(cfg["n_layers"]):
        spec[prefix + "layers.%d.weight" % (2 * i)], spec[prefix + "layers.%d.bias" % (2 * i)] = (HF, cur, 3, 3), (HF,)
        cur = HF
    for h in cfg["heads"]:
        q = prefix + "heads.%s." % h["name"]
        n = h["n_hidden_layers"]
        for j in range(n):
            spec[q + "%d.weight" % (2 * j)], spec[q + "%d.bias" % (2 * j)] = (HF, HF), (HF,)
        spec[q + "%d.weight" % (2 * n)], spec[q + "%d.bias" % (2 * n)] = (h["out_channels"], HF), (h["out_channels"],)
    return spec


def _np(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def _softplus(x):
    """F.softplus (beta 1, threshold 20) in float32."""
    x = np.asarray(x, np.float32)
    return np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, np.float32(20.0))))).astype(np.float32)


def beta_mode(alpha, beta):
    """torch.distributions.Beta(alpha, beta).mode: Dirichlet mode of (alpha, beta), first component."""
    conc = np.stack([np.asarray(alpha, np.float32), np.asarray(beta, np.float32)], -1)
    cm1 = np.maximum(conc - np.float32(1.0), np.float32(0.0))
    with np.errstate(invalid="ignore", divide="ignore"):
        mode = cm1 / cm1.sum(-1, keepdims=True)
    small = (conc < 1).all(-1)
    if small.any():
        onehot = np.zeros_like(conc)
        onehot[np.arange(conc.shape[0]), mode.argmax(-1)] = 1.0   # (0/0 rows: argmax of NaNs is 0, like torch)
        mode[small] = onehot[small]
    return mode[..., 0].astype(np.float32)


def _output_activation(name, x):
    """sf3d/models/network.py:98-136, the activations the estimator configs use (host, a handful of numbers)."""
    if name is None or str(name).lower() in ("none", "linear", "identity"):
        return x
    name = str(name).lower()
    if name == "softplus":
        return _softplus(x)
    if name == "sigmoid":
        return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(np.float32)
    if name == "exp":
        return np.exp(x).astype(np.float32)
    if name == "tanh":
        return np.tanh(x).astype(np.float32)
    raise _lib.SculptError("estimator head: output activation %r unsupported" % name)


class _Estimator(KernelEngine):
    prefix = ""

    def __init__(self, cfg, precision="bf16"):
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision must be 'bf16' or 'fp32'")
        self.cfg = dict(cfg)
        self.cfg["heads"] = tuple(_head_cfg(h) for h in cfg["heads"])
        if self.cfg.get("activation", "relu") != "relu":
            raise _lib.SculptError("%s: activation %r unsupported (relu only)" % (type(self).__name__, self.cfg["activation"]))
        self.precision = precision
        self.adt = BF16 if precision == "bf16" else torch.float32
        self._buf = {}
        self._sd = None
        self._w = None
        self.device = None

    def _wt(self, x, dev):
        return torch.as_tensor(np.ascontiguousarray(x)).to(device=dev, dtype=self.adt).contiguous()

    @staticmethod
    def _f32(x, dev):
        return torch.as_tensor(np.ascontiguousarray(x)).to(device=dev, dtype=torch.float32).contiguous()

    def load_state_dict(self, sd, strict=True):
        """sd: the SF3D state dict (keys under self.prefix are read; everything else is left alone)."""
        missing = [k for k in self._spec if k not in sd]
        if missing:
            raise RuntimeError("Error(s) in loading state_dict for %s: missing %s" % (type(self).__name__, missing[:5]))
        for k, shp in self._spec.items():
            if tuple(sd[k].shape) != tuple(shp):
                raise RuntimeError("size mismatch for %s: %s vs %s" % (k, tuple(sd[k].shape), shp))
        self._sd = {k: _np(sd[k]).astype(np.float32) for k in self._spec}
        if self.device is not None:
            self._prepare(self.device)
        return self

    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise _lib.SculptError("%s runs on an MI355X only (device %s requested; there is no CPU fallback)"
                                   % (type(self).__name__, device))
        self.device = device
        if self._sd is not None:
            self._prepare(device)
        return self

    def _head_linear(self, x, W, b, rows, relu, name):
        """fp32 Linear on [rows, K] (exact-fp32 kernel); W is stored with its row count padded to a multiple of 4."""
        out = self._b(name, (max(rows, 1), W.shape[0]), torch.float32)
        ops.gemm_f32(x, W, bias=b, out=out, M=rows, epilogue=_lib.EPI_RELU if relu else _lib.EPI_NONE)
        return out

    def _pad_rows(self, W, b, dev):
        n = W.shape[0]
        npad = ((n + 3) // 4) * 4
        Wp = np.zeros((npad, W.shape[1]), np.float32)
        bp = np.zeros((npad,), np.float32)
        Wp[:n], bp[:n] = W, b
        return self._f32(Wp, dev), self._f32(bp, dev)


class ClipBasedHeadEstimator(_Estimator):
    prefix = "image_estimator."

    def __init__(self, cfg=None, precision="bf16"):
        cfg = dict(IMAGE_ESTIMATOR_CFG if cfg is None else cfg)
        cfg.setdefault("clip", CLIP_VIT_B_32)
        cfg.setdefault("hidden_features", 512)
        if cfg.get("model", "ViT-B-32") != "ViT-B-32":
            raise _lib.SculptError("ClipBasedHeadEstimator: only open_clip's ViT-B-32 visual tower is built")
        if cfg.get("distribution", "beta") != "beta" or cfg.get("distribution_eval", "mode") != "mode":
            raise _lib.SculptError("ClipBasedHeadEstimator: only distribution=beta / distribution_eval=mode (the shipped "
                                   "configuration) is built")
        super().__init__(cfg, precision)
        if self.cfg["clip"]["width"] // self.cfg["clip"]["heads"] != 64:
            raise _lib.SculptError("ClipBasedHeadEstimator: the attention kernel needs a head width of 64")
        if self.cfg["clip"]["embed_dim"] != self.cfg["hidden_features"]:
            raise _lib.SculptError("ClipBasedHeadEstimator: hidden_features must equal CLIP's embedding width")
        self._spec = image_estimator_param_spec(self.cfg, self.prefix)

    def _prepare(self, dev):
        sd, c = self._sd, self.cfg["clip"]
        p = self.prefix + "model.visual."
        W = c["width"]
        w = {"conv": self._wt(sd[p + "conv1.weight"].reshape(W, -1), dev), "cls": self._f32(sd[p + "class_embedding"], dev),
             "pos": self._f32(sd[p + "positional_embedding"], dev),
             "ln_pre": (self._f32(sd[p + "ln_pre.weight"], dev), self._f32(sd[p + "ln_pre.bias"], dev)),
             "ln_post": (self._f32(sd[p + "ln_post.weight"], dev), self._f32(sd[p + "ln_post.bias"], dev)),
             "proj": self._f32(np.ascontiguousarray(sd[p + "proj"].T), dev), "blocks": [], "heads": {}}
        for i in range(c["layers"]):
            q = p + "transformer.resblocks.%d." % i
            w["blocks"].append(dict(
                ln1=(self._f32(sd[q + "ln_1.weight"], dev), self._f32(sd[q + "ln_1.bias"], dev)),
                ln2=(self._f32(sd[q + "ln_2.weight"], dev), self._f32(sd[q + "ln_2.bias"], dev)),
                qkv_w=self._wt(sd[q + "attn.in_proj_weight"], dev), qkv_b=self._f32(sd[q + "attn.in_proj_bias"], dev),
                o_w=self._wt(sd[q + "attn.out_proj.weight"], dev), o_b=self._f32(sd[q + "attn.out_proj.bias"], dev),
                f1_w=self._wt(sd[q + "mlp.c_fc.weight"], dev), f1_b=self._f32(sd[q + "mlp.c_fc.bias"], dev),
                f2_w=self._wt(sd[q + "mlp.c_proj.weight"], dev), f2_b=self._f32(sd[q + "mlp.c_proj.bias"], dev)))
        for h in self.cfg["heads"]:
            q = self.prefix + "heads.%s." % h["name"]
            trunk = [(self._f32(sd[q + "0.%d.weight" % (2 * j)], dev), self._f32(sd[q + "0.%d.bias" % (2 * j)], dev))
                     for j in range(h["n_hidden_layers"])]
            hid = [(self._f32(sd[q + "%d.0.weight" % br], dev), self._f32(sd[q + "%d.0.bias" % br], dev)) for br in (1, 2)]
            last = [self._pad_rows(sd[q + "%d.2.weight" % br], sd[q + "%d.2.bias" % br], dev) for br in (1, 2)]
            w["heads"][h["name"]] = (trunk, hid, last)
        self._w = w

    # ------------------------------------------------------------------ CLIP visual tower
    def encode_image(self, cond_hwc: torch.Tensor, mask_hw=None) -> torch.Tensor:
        """cond_hwc fp32 [H, W, 3] on the device (times mask_hw [H, W] if given) -> image features fp32 [embed_dim]."""
        c, w = self.cfg["clip"], self._w
        W, P, nh, S = c["width"], c["patch_size"], c["heads"], c["image_size"]
        small = self._b("clip_in", (S, S, 3), torch.float32)
        ops.resize_bilinear_hwc(cond_hwc.contiguous(), S, mul_hw=mask_hw, out=small)
        n_side = S // P
        npatch, T = n_side * n_side, n_side * n_side + 1
        Tp = ((T + 63) // 64) * 64
        patches = self._b("clip_patches", (npatch, 3 * P * P), self.adt)
        ops.vit_patchify(small, P, OPENAI_DATASET_MEAN, OPENAI_DATASET_STD, patches)
        pout = self._b("clip_patch_out", (npatch, W), torch.float32)
        self._gemm(patches, w["conv"], out_f32=pout)
        h0 = self._b("clip_h0", (T, W), torch.float32)
        ops.vit_assemble(pout, w["cls"], w["pos"], h0)
        h = self._b("clip_h", (T, W), torch.float32)
        ops.layernorm(h0, w["ln_pre"][0], w["ln_pre"][1], c["eps"], y_f32=h)
        xn = self._b("clip_xn", (T, W), self.adt)
        qk = self._b("clip_qk", (T, 2 * W), self.adt)
        vt = self._b("clip_vt", (W, Tp), self.adt, zero=True)
        att = self._b("clip_att", (T, W), self.adt)
        ff = self._b("clip_ff", (T, c["mlp"]), self.adt)
        scale = 1.0 / math.sqrt(W // nh)
        for L in w["blocks"]:
            self._ln(h, L["ln1"][0], L["ln1"][1], c["eps"], xn)
            self._gemm(xn, L["qkv_w"], bias=L["qkv_b"], out_bf16=qk, out_t=vt, n_split=2 * W)
            self._attn(qk[:, :W], qk[:, W:], vt, att, T, T, nh, scale)
            self._gemm(att, L["o_w"], bias=L["o_b"], residual=h, out_f32=h)
            self._ln(h, L["ln2"][0], L["ln2"][1], c["eps"], xn)
            self._gemm(xn, L["f1_w"], bias=L["f1_b"], out_bf16=ff, epilogue=_lib.EPI_GELU)
            self._gemm(ff, L["f2_w"], bias=L["f2_b"], residual=h, out_f32=h)
        pooled = self._b("clip_pooled", (1, W), torch.float32)
        ops.layernorm(h[:1], w["ln_post"][0], w["ln_post"][1], c["eps"], y_f32=pooled)
        feats = self._b("clip_feats", (1, c["embed_dim"]), torch.float32)
        ops.gemm_f32(pooled, w["proj"], out=feats, M=1)
        return feats[0]

    # ------------------------------------------------------------------ heads
    def heads_forward(self, features: torch.Tensor):
        """features fp32 [B, hidden] on the device -> (outputs dict of numpy arrays, {name: (alpha, beta)})."""
        B = features.shape[0]
        out, dists = {}, {}
        for h in self.cfg["heads"]:
            trunk, hid, last = self._w["heads"][h["name"]]
            y = features
            for j, (Wj, bj) in enumerate(trunk):
                y = self._head_linear(y, Wj, bj, B, True, "ie_t%d_%d" % (j % 2, B))
            d = []
            for br in range(2):
                z = self._head_linear(y, hid[br][0], hid[br][1], B, True, "ie_h%d_%d" % (br, B))
                d.append(self._head_linear(z, last[br][0], last[br][1], B, False, "ie_o%d_%d" % (br, B))[:B, 0].cpu().numpy())
            alpha = _softplus(d[0] + np.float32(h["output_bias"]))
            beta = _softplus(d[1] + np.float32(h["output_bias"]))
            v = _output_activation(h["output_activation"], beta_mode(alpha, beta))
            if h["shape"]:
                v = v.reshape(h["shape"])
            out[("decoder_" if h["add_to_decoder_features"] else "") + h["name"]] = v
            dists[h["name"]] = (alpha, beta)
        return out, dists

    def __call__(self, cond_image, mask=None):
        """cond_image fp32 [B, 1, H, W, 3] or [B, H, W, 3] on the device (clip_based_estimator.py:88-92; pass rgb_cond with
        mask = mask_cond to have the product formed inside the resize kernel) -> {"decoder_roughness": [B,1,1], ...}."""
        if self._w is None:
            raise _lib.SculptError("ClipBasedHeadEstimator: weights not loaded / not on a device")
        x = cond_image.reshape(-1, *cond_image.shape[-3:])
        m = None if mask is None else mask.reshape(x.shape[0], x.shape[1], x.shape[2])
        feats = self._b("ie_feats_%d" % x.shape[0], (x.shape[0], self.cfg["clip"]["embed_dim"]), torch.float32)
        for b in range(x.shape[0]):
            feats[b].copy_(self.encode_image(x[b], None if m is None else m[b].contiguous()))
        return self.heads_forward(feats)[0]


class MultiHeadEstimator(_Estimator):
    prefix = "global_estimator."

    def __init__(self, cfg=None, precision="bf16"):
        cfg = dict(GLOBAL_ESTIMATOR_CFG if cfg is None else cfg)
        for k, v in GLOBAL_ESTIMATOR_CFG.items():
            cfg.setdefault(k, v)
        super().__init__(cfg, precision)
        if self.cfg["pool"] not in ("max", "mean"):
            raise NotImplementedError(self.cfg["pool"])
        if self.cfg["hidden_features"] % 128 or (3 * self.cfg["triplane_features"]) % 64:
            raise _lib.SculptError("MultiHeadEstimator: hidden_features % 128 == 0 and 3*triplane_features % 64 == 0 needed")
        self._spec = global_estimator_param_spec(self.cfg, self.prefix)

    def _prepare(self, dev):
        sd = self._sd
        w = {"convs": [], "heads": {}}
        for i in range(self.cfg["n_layers"]):
            Wc = sd[self.prefix + "layers.%d.weight" % (2 * i)]               # [HF, Cin, 3, 3] -> [HF][(ky*3+kx)*Cin + c]
            w["convs"].append((self._wt(Wc.transpose(0, 2, 3, 1).reshape(Wc.shape[0], -1), dev),
                               self._f32(sd[self.prefix + "layers.%d.bias" % (2 * i)], dev)))
        for h in self.cfg["heads"]:
            q = self.prefix + "heads.%s." % h["name"]
            n = h["n_hidden_layers"]
            layers = [(self._f32(sd[q + "%d.weight" % (2 * j)], dev), self._f32(sd[q + "%d.bias" % (2 * j)], dev)) for j in range(n)]
            layers.append(self._pad_rows(sd[q + "%d.weight" % (2 * n)], sd[q + "%d.bias" % (2 * n)], dev))
            w["heads"][h["name"]] = layers
        self._w = w

    def pooled_features(self, tokens: torch.Tensor, S: int) -> torch.Tensor:
        """tokens [3*S*S][F] (plane-major, channel-last: what the backbone leaves) -> fp32 [hidden] after the convolutions
        and the pool."""
        F_ = self.cfg["triplane_features"]
        assert tokens.shape == (3 * S * S, F_) and tokens.is_contiguous()
        act = tokens if tokens.dtype == self.adt else tokens.to(self.adt)
        groups, size, C = 3, S, F_
        HF = self.cfg["hidden_features"]
        y = None
        for i, (Wc, bc) in enumerate(self._w["convs"]):
            So = (size - 3) // 2 + 1
            if So < 1:
                raise _lib.SculptError("MultiHeadEstimator: the triplane is too small for %d stride-2 convolutions" % len(self._w["convs"]))
            rows = self._b("ge_rows%d" % i, (So * So, 9 * groups * C), self.adt)
            ops.im2col3x3_strided(act, groups, size, 2, rows)
            last = i + 1 == len(self._w["convs"])
            if last:
                y = self._b("ge_y%d" % i, (So * So, HF), torch.float32)
                self._gemm(rows, Wc, bias=bc, out_f32=y, epilogue=_lib.EPI_RELU)
            else:
                y = self._b("ge_y%d" % i, (So * So, HF), self.adt)
                self._gemm(rows, Wc, bias=bc, out_bf16=y, epilogue=_lib.EPI_RELU)
            act, groups, size, C = y, 1, So, HF
        pooled = self._b("ge_pooled", (1, HF), torch.float32)
        ops.col_reduce(y, size * size, pooled[0], mean=self.cfg["pool"] == "mean")
        return pooled

    def heads_forward(self, pooled: torch.Tensor):
        B = pooled.shape[0]
        out = {}
        for h in self.cfg["heads"]:
            layers = self._w["heads"][h["name"]]
            y = pooled
            for j, (Wj, bj) in enumerate(layers[:-1]):
                y = self._head_linear(y, Wj, bj, B, True, "ge_t%d_%d" % (j % 2, B))
            o = self._head_linear(y, layers[-1][0], layers[-1][1], B, False, "ge_o_%s_%d" % (h["name"], B))
            v = o[:B, :h["out_channels"]].cpu().numpy() + np.float32(h["output_bias"])
            v = _output_activation(h["output_activation"], v)
            if h["shape"]:
                v = v.reshape(h["shape"])
            out[("decoder_" if h["add_to_decoder_features"] else "") + h["name"]] = v
        return out

    def __call__(self, tokens_list, S):
        """tokens_list: per image, the backbone's output tokens [3*S*S][F] on the device (== non_postprocessed_codes
        [B, 3, F, S, S] of sf3d/system.py:330-331 in its channel-last form) -> dict like MultiHeadEstimator.forward."""
        if self._w is None:
            raise _lib.SculptError("MultiHeadEstimator: weights not loaded / not on a device")
        feats = self._b("ge_feats_%d" % len(tokens_list), (len(tokens_list), self.cfg["hidden_features"]), torch.float32)
        for b, t in enumerate(tokens_list):
            feats[b].copy_(self.pooled_features(t, S)[0])
        return self.heads_forward(feats)
