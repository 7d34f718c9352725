"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by sculptmate_amd/).

Plain PyTorch fp32 restatement (CPU) of TSR.forward, the floating-point part of the hot path:
  TSR.forward                         /root/reference/TripoSR/tsr/system.py:82-115
  DINOSingleImageTokenizer.forward    /root/reference/TripoSR/tsr/models/tokenizers/image.py:41-60
     + HF ViTModel (third-party `transformers`, pinned 4.38.0 by the reference, __init__.py:37):
       patch conv16/16, CLS, bicubic-interpolated position embeddings, 12 pre-LN blocks
       (MHA with qkv bias, GELU(erf) MLP), final LayerNorm (eps 1e-12)
  Triplane1DTokenizer                 /root/reference/TripoSR/tsr/models/tokenizers/triplane.py:29-45
  Transformer1D.forward               /root/reference/TripoSR/tsr/models/transformer/transformer_1d.py:179-219
  BasicTransformerBlock.forward       /root/reference/TripoSR/tsr/models/transformer/basic_transformer_block.py:149-206
  Attention / AttnProcessor2_0        /root/reference/TripoSR/tsr/models/transformer/attention.py:569-653
  FeedForward / GEGLU                 basic_transformer_block.py:209-259, 291-315
  TriplaneUpsampleNetwork.forward     /root/reference/TripoSR/tsr/models/network_utils.py:24-32

Weights: a dict name -> tensor with the reference checkpoint's key names (HF-4.38 ViT names).
PARITY PIN: tests/golden/tsr_tiny.npz and tsr_block.npz, produced by running the reference's own
modules (and the installed `transformers` ViT, "size" position-embedding mode) in the build
container -- tests/golden/make_reference_goldens.py tiny / block.

`bf16=True` rounds tensors to bfloat16 at the points where the HIP pipeline stores bf16 (weights,
GEMM operands, q/k/v, softmax probabilities, attention output, GEGLU output): the GPU tests
compare against that variant tightly and against the fp32 variant with the documented bf16 tolerance.
Since round 2 the HIP pipeline folds every LayerNorm into the GEMM that consumes it: the GEMM's
operand is bf16(x) -- the UN-normalised residual row -- against bf16(W * gamma), and mean / rstd (from the fp32 x)
are applied to the fp32 accumulator; `_ln_linear` restates exactly that for bf16=True and is the plain
F.linear(F.layer_norm(x)) of the reference for bf16=False.
"""
import math

import torch
import torch.nn.functional as F

IMAGE_MEAN = (0.485, 0.456, 0.406)
IMAGE_STD = (0.229, 0.224, 0.225)


def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(x)


class _Q:
    def __init__(self, bf16):
        self.on = bf16

    def __call__(self, x):
        return x.to(torch.bfloat16).to(torch.float32) if self.on else x


LOG2E = 1.4426950408889634


def _ln_linear(h, gamma, beta, eps, W, bias, Q, q_scale=None):
    """Linear(LayerNorm(h)).  fp32 (Q off): the reference's two calls.  bf16 emulation: the folded form the HIP
    GEMM computes -- rstd * (bf16(h) . bf16(W*gamma)^T - mean * colsum) + (bias + W . beta).
    q_scale (bf16 emulation only): this is an attention's query projection and the HIP pipeline stores
    bf16(softmax_scale * log2(e) * q) -- the scale goes into the weights before their bf16 rounding; _attn then takes the
    scores as exponents of 2."""
    D = h.shape[-1]
    if not Q.on:
        return F.linear(F.layer_norm(h, (D,), gamma, beta, eps), W, bias)
    if q_scale is not None:
        W = W * q_scale
        bias = None if bias is None else bias * q_scale
    Wp = Q(W * gamma[None, :])
    bp = (W.double() @ beta.double()).float()
    if bias is not None:
        bp = bp + bias
    mean = h.mean(-1, keepdim=True)
    rstd = torch.rsqrt(h.var(-1, unbiased=False, keepdim=True) + eps)
    acc = F.linear(Q(h), Wp)
    return rstd * (acc - mean * Wp.sum(1)[None, :]) + bp


def interpolate_pos(pos, n_side, mode):
    """HF ViTEmbeddings.interpolate_pos_encoding: 'scale_factor' = transformers 4.38 (+0.1 hack),
    'size' = transformers >= 4.4x."""
    pos = _t(pos).float()
    n_pos = pos.shape[1] - 1
    g = int(math.sqrt(n_pos))
    if n_side == g:
        return pos[0]
    dim = pos.shape[-1]
    patch = pos[:, 1:].reshape(1, g, g, dim).permute(0, 3, 1, 2)
    if mode == "scale_factor":
        s = (n_side + 0.1) / g
        patch = F.interpolate(patch, scale_factor=(s, s), mode="bicubic", align_corners=False)
    else:
        patch = F.interpolate(patch, size=(n_side, n_side), mode="bicubic", align_corners=False)
    assert patch.shape[-1] == n_side
    patch = patch.permute(0, 2, 3, 1).reshape(-1, dim)
    return torch.cat([pos[0, :1], patch], 0)


def _attn(q, k, v, heads, Q, prescaled=None):
    """prescaled (default: in the bf16 emulation): q already carries softmax_scale * log2(e)."""
    if prescaled is None:
        prescaled = Q.on
    Tq, D = q.shape
    hd = D // heads
    qh = q.view(Tq, heads, hd).transpose(0, 1)
    kh = k.view(-1, heads, hd).transpose(0, 1)
    vh = v.view(-1, heads, hd).transpose(0, 1)
    if Q.on:
        # prescaled: q carries softmax_scale * log2(e) already (_ln_linear q_scale): the scores are exponents of 2.  The
        # kernel keeps fp32 scores, rounds 2^(s - max) to bf16 for the PV product, and divides by the fp32 row sum at the end
        # (its running maximum may lag the true one by a power-of-two-ish factor: bf16 rounding noise only)
        s = qh @ kh.transpose(1, 2)
        if not prescaled:
            s = s * (LOG2E / math.sqrt(hd))
        m = s.amax(-1, keepdim=True)
        p = torch.exp2(s - m)
        o = (Q(p) @ vh) / p.sum(-1, keepdim=True)
    else:
        s = (qh @ kh.transpose(1, 2)) * (1.0 / math.sqrt(hd))
        o = torch.softmax(s, -1) @ vh
    return o.transpose(0, 1).reshape(Tq, D)


def vit_forward(sd, image_hwc, cfg, pos_mode="scale_factor", bf16=False, collect=None):
    """image_hwc float32 [S,S,3] in [0,1] -> last_hidden_state [T, H] (CLS kept at index 0)."""
    Q = _Q(bf16)
    v = cfg["image_tokenizer"]
    H, P, nh, eps = v["hidden_size"], v["patch_size"], v["num_attention_heads"], v["layer_norm_eps"]
    p = "image_tokenizer.model."
    g = lambda k: _t(sd[p + k]).float()  # noqa: E731
    x = _t(image_hwc).float().permute(2, 0, 1)[None]
    x = (x - torch.tensor(IMAGE_MEAN).view(1, 3, 1, 1)) / torch.tensor(IMAGE_STD).view(1, 3, 1, 1)
    x = F.conv2d(Q(x), Q(g("embeddings.patch_embeddings.projection.weight")),
                 g("embeddings.patch_embeddings.projection.bias"), stride=P)
    n_side = x.shape[-1]
    x = x.flatten(2).transpose(1, 2)[0]
    h = torch.cat([g("embeddings.cls_token").view(1, H), x], 0)
    h = h + interpolate_pos(sd[p + "embeddings.position_embeddings"], n_side, pos_mode)
    for i in range(v["num_hidden_layers"]):
        q = "encoder.layer.%d." % i
        l1 = (g(q + "layernorm_before.weight"), g(q + "layernorm_before.bias"), eps)
        qq = Q(_ln_linear(h, *l1, g(q + "attention.attention.query.weight"), g(q + "attention.attention.query.bias"), Q,
                          q_scale=LOG2E / math.sqrt(H // nh)))
        kk = Q(_ln_linear(h, *l1, g(q + "attention.attention.key.weight"), g(q + "attention.attention.key.bias"), Q))
        vv = Q(_ln_linear(h, *l1, g(q + "attention.attention.value.weight"), g(q + "attention.attention.value.bias"), Q))
        a = Q(_attn(qq, kk, vv, nh, Q))
        h = F.linear(a, Q(g(q + "attention.output.dense.weight")), g(q + "attention.output.dense.bias")) + h
        f = Q(F.gelu(_ln_linear(h, g(q + "layernorm_after.weight"), g(q + "layernorm_after.bias"), eps,
                                g(q + "intermediate.dense.weight"), g(q + "intermediate.dense.bias"), Q)))
        h = F.linear(f, Q(g(q + "output.dense.weight")), g(q + "output.dense.bias")) + h
        if collect is not None:
            collect["vit_layer%d" % i] = h.clone()
    return F.layer_norm(h, (H,), g("layernorm.weight"), g("layernorm.bias"), eps)


def block_forward(sd, prefix, h, ctx, heads, bf16=False):
    """One BasicTransformerBlock: h [T, D] fp32, ctx [Tc, cross_dim]."""
    Q = _Q(bf16)
    g = lambda k: _t(sd[prefix + k]).float()  # noqa: E731
    D = h.shape[1]
    n1 = (g("norm1.weight"), g("norm1.bias"), 1e-5)
    qs = LOG2E / math.sqrt(D // heads)
    a = Q(_attn(Q(_ln_linear(h, *n1, g("attn1.to_q.weight"), None, Q, q_scale=qs)), Q(_ln_linear(h, *n1, g("attn1.to_k.weight"), None, Q)),
                Q(_ln_linear(h, *n1, g("attn1.to_v.weight"), None, Q)), heads, Q))
    h = F.linear(a, Q(g("attn1.to_out.0.weight")), g("attn1.to_out.0.bias")) + h
    c = Q(ctx)
    a = Q(_attn(Q(_ln_linear(h, g("norm2.weight"), g("norm2.bias"), 1e-5, g("attn2.to_q.weight"), None, Q, q_scale=qs)),
                Q(F.linear(c, Q(g("attn2.to_k.weight")))), Q(F.linear(c, Q(g("attn2.to_v.weight")))), heads, Q))
    h = F.linear(a, Q(g("attn2.to_out.0.weight")), g("attn2.to_out.0.bias")) + h
    pr = _ln_linear(h, g("norm3.weight"), g("norm3.bias"), 1e-5, g("ff.net.0.proj.weight"), g("ff.net.0.proj.bias"), Q)
    val, gate = pr.chunk(2, dim=-1)
    f = Q(val * F.gelu(gate))
    h = F.linear(f, Q(g("ff.net.2.weight")), g("ff.net.2.bias")) + h
    return h


def backbone_forward(sd, ctx, cfg, bf16=False, collect=None):
    """Triplane1DTokenizer + Transformer1D: ctx [Tc, cross_dim] -> tokens [C, 3*S*S] (reference layout)."""
    Q = _Q(bf16)
    b, t = cfg["backbone"], cfg["tokenizer"]
    C, S = t["num_channels"], t["plane_size"]
    emb = _t(sd["tokenizer.embeddings"]).float()
    tokens = emb.permute(1, 0, 2, 3).reshape(1, C, 3 * S * S)  # "B Np Ct Hp Wp -> B Ct (Np Hp Wp)"
    residual = tokens
    x = F.group_norm(tokens, b["norm_num_groups"], _t(sd["backbone.norm.weight"]).float(),
                     _t(sd["backbone.norm.bias"]).float(), 1e-6)
    x = Q(x[0].t())  # [T, C]
    h = F.linear(x, Q(_t(sd["backbone.proj_in.weight"]).float()), _t(sd["backbone.proj_in.bias"]).float())
    for i in range(b["num_layers"]):
        h = block_forward(sd, "backbone.transformer_blocks.%d." % i, h, ctx, b["num_attention_heads"], bf16)
        if collect is not None:
            collect["block%d" % i] = h.clone()
    o = F.linear(Q(h), Q(_t(sd["backbone.proj_out.weight"]).float()), _t(sd["backbone.proj_out.bias"]).float())
    return o.t() + residual[0]  # [C, T]


def upsample_forward(sd, tokens_ct, cfg, bf16=False):
    """detokenize (triplane.py:35-45) + ConvTranspose2d(k2,s2) -> [3, Co, 2S, 2S]."""
    Q = _Q(bf16)
    t = cfg["tokenizer"]
    C, S = t["num_channels"], t["plane_size"]
    x = tokens_ct.reshape(C, 3, S, S).permute(1, 0, 2, 3)  # "Ct (Np Hp Wp) -> Np Ct Hp Wp"
    return F.conv_transpose2d(Q(x), Q(_t(sd["post_processor.upsample.weight"]).float()),
                              _t(sd["post_processor.upsample.bias"]).float(), stride=2)


def tsr_forward(sd, image_hwc, cfg, pos_mode="scale_factor", bf16=False, collect=None):
    """TSR.forward for one image -> scene code [3, Co, 2S, 2S] fp32."""
    with torch.no_grad():
        ctx = vit_forward(sd, image_hwc, cfg, pos_mode, bf16, collect)
        if collect is not None:
            collect["ctx"] = ctx.clone()
        tok = backbone_forward(sd, ctx, cfg, bf16, collect)
        if collect is not None:
            collect["tokens"] = tok.clone()
        return upsample_forward(sd, tok, cfg, bf16)
