"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by sculptmate_amd/).

Plain PyTorch fp32 restatement (CPU) of the StableFast-3D networks and marching tetrahedra (BASELINE config 4,
SURVEY.md section 8f rows 1-2).  Reference file:line followed by each function:

  camera_embedding          StableFast/sf3d/models/camera.py:21-32, sf3d/utils.py:24-50, models/utils.py:223-236
  dino_forward              sf3d/models/tokenizers/image.py:64-96 (mean/std, rearranges),
                            sf3d/models/tokenizers/dinov2.py:68-163 (embeddings, bicubic pos-emb with the +0.1 hack),
                            :213-310 (self attention), :380-396 (LayerScale), :432-448 (MLP), :468-546 (layer with
                            adaLN modulation), :798-830 (final LayerNorm); Modulation transformers/attention.py:5-31
  backbone_forward          sf3d/models/transformers/backbone.py:36-83 (CrossAttention), :86-107 (FeedForward/GEGLU),
                            :110-156 (BasicBlock), :218-257 (FuseBlock), :345-396 (TwoStreamBlock),
                            :398-515 (TwoStreamInterleaveTransformer)
  post_forward              sf3d/models/network.py:29-75 (3x3 convs + ReLU, PixelShuffle)
  query_triplane            sf3d/system.py:170-199 (grid_sample bilinear, align_corners=True, zeros padding)
  decoder_forward           sf3d/models/network.py:148-210 (MaterialMLP heads) and :96-135 (output activations)
  marching_tets             sf3d/models/isosurface.py:108-229 (deformation, unique crossing edges, interpolation,
                            triangle table); all_edges :117-131
  get_scene_codes / triplane_to_meshes   sf3d/system.py:140-168, 201-236

Weights: dict name -> array with the reference's state_dict key names.
PARITY PIN: tests/golden/sf3d_{dino,backbone,post,decoder,mtet,camera}.npz, produced by running the reference's own
classes in the build container (tests/golden/make_sf3d_goldens.py); tests/test_oracle_sf3d.py.

`bf16=True` rounds at the points where the HIP pipeline stores bf16 (see oracle/tsr_ref.py).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .tsr_ref import IMAGE_MEAN, IMAGE_STD, LOG2E, _Q, _attn, _ln_linear, _t


# ----------------------------------------------------------------------------- camera
def default_cond_c2w(distance):
    return torch.tensor([[0, 0, 1, distance], [1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=torch.float32)


def intrinsic_from_fov_deg(fov_deg, H, W):
    fov = np.deg2rad(fov_deg)
    focal = 0.5 * H / np.tan(0.5 * fov)
    K = np.identity(3, dtype=np.float32)
    K[0, 0] = focal
    K[1, 1] = focal
    K[0, 2] = W / 2.0
    K[1, 2] = H / 2.0
    K = torch.from_numpy(K)
    Kn = K.clone()
    Kn[0, 2] /= W
    Kn[1, 2] /= H
    Kn[0, 0] /= W
    Kn[1, 1] /= H
    return K, Kn


def camera_embedding(sd, prefix, distance=1.6, fov_deg=40.0, size=512):
    """LinearCameraEmbedder on (c2w_cond, intrinsic_normed_cond) -> [Cc]."""
    c2w = default_cond_c2w(distance)
    _, Kn = intrinsic_from_fov_deg(fov_deg, size, size)
    cond = torch.cat([c2w.reshape(-1), Kn.reshape(-1)])
    return F.linear(cond, _t(sd[prefix + "linear.weight"]).float(), _t(sd[prefix + "linear.bias"]).float())


# ----------------------------------------------------------------------------- DINOv2 with adaLN modulation
def dino_interpolate_pos(pos, n_side):
    pos = _t(pos).float()
    n_pos = pos.shape[1] - 1
    g = int(math.sqrt(n_pos))
    if n_side * n_side == n_pos:
        return pos[0]
    dim = pos.shape[-1]
    patch = pos[:, 1:].reshape(1, g, g, dim).permute(0, 3, 1, 2)
    s = (n_side + 0.1) / math.sqrt(n_pos)
    patch = F.interpolate(patch, scale_factor=(s, s), mode="bicubic", align_corners=False)
    assert patch.shape[-1] == n_side and patch.shape[-2] == n_side
    patch = patch.permute(0, 2, 3, 1).reshape(-1, dim)
    return torch.cat([pos[0, :1], patch], 0)


def _modulate(x, sd, key, cond):
    """Modulation(single_layer=True): emb = linear2(silu(cond)); x * (1 + scale) + shift."""
    emb = F.linear(F.silu(cond), _t(sd[key + "linear2.weight"]).float(), _t(sd[key + "linear2.bias"]).float())
    scale, shift = emb.chunk(2, dim=-1)
    return x * (1 + scale) + shift


def _modulated(sd, key, gamma, beta, cond):
    """adaLN as LayerNorm parameters: LN(x; g, b) * (1 + scale) + shift = LN(x; g (1 + scale), b (1 + scale) + shift)."""
    emb = F.linear(F.silu(cond), _t(sd[key + "_modulation.linear2.weight"]).float(), _t(sd[key + "_modulation.linear2.bias"]).float())
    scale, shift = emb.chunk(2, dim=-1)
    return gamma * (1 + scale), beta * (1 + scale) + shift


def dino_forward(sd, image_hwc, cond, cfg, prefix="image_tokenizer.", bf16=False):
    """image [S,S,3] in [0,1], cond [Cc] -> last_hidden_state [T, H] (CLS kept)."""
    Q = _Q(bf16)
    H, P, nh, eps, nl = cfg["hidden_size"], cfg["patch_size"], cfg["num_attention_heads"], cfg["layer_norm_eps"], \
        cfg["num_hidden_layers"]
    p = prefix + "model."
    g = lambda k: _t(sd[p + k]).float()  # noqa: E731
    x = _t(image_hwc).float().permute(2, 0, 1)[None]
    x = (x - torch.tensor(IMAGE_MEAN).view(1, 3, 1, 1)) / torch.tensor(IMAGE_STD).view(1, 3, 1, 1)
    x = F.conv2d(Q(x), Q(g("embeddings.patch_embeddings.projection.weight")),
                 g("embeddings.patch_embeddings.projection.bias"), stride=P)
    n_side = x.shape[-1]
    x = x.flatten(2).transpose(1, 2)[0]
    h = torch.cat([g("embeddings.cls_token").view(1, H), x], 0)
    h = h + dino_interpolate_pos(sd[p + "embeddings.position_embeddings"], n_side)
    cond = _t(cond).float().view(-1)
    for i in range(nl):
        q = "encoder.layer.%d." % i
        # bf16 emulation: the HIP pipeline stores bf16(softmax_scale * log2(e) * q) (scale folded into the projection before its
        # bf16 rounding) and _attn takes the scores as exponents of 2 (oracle/tsr_ref.py); every LayerNorm (here with its adaLN
        # modulation, a constant of the camera) is folded into the projection that consumes it (tsr_ref._ln_linear).
        # fp32: the reference's own sequence of calls.
        qc = LOG2E / math.sqrt(H // nh) if Q.on else 1.0
        att = lambda n: (g(q + "attention.attention.%s.weight" % n), g(q + "attention.attention.%s.bias" % n))  # noqa: E731
        if Q.on:
            gm, bm = _modulated(sd, p + q + "norm1", g(q + "norm1.weight"), g(q + "norm1.bias"), cond)
            qq = Q(_ln_linear(h, gm, bm, eps, *att("query"), Q, q_scale=qc))
            kk = Q(_ln_linear(h, gm, bm, eps, *att("key"), Q))
            vv = Q(_ln_linear(h, gm, bm, eps, *att("value"), Q))
        else:
            xn = F.layer_norm(h, (H,), g(q + "norm1.weight"), g(q + "norm1.bias"), eps)
            xn = _modulate(xn, sd, p + q + "norm1_modulation.", cond)
            qq, kk, vv = (F.linear(xn, *att(n)) for n in ("query", "key", "value"))
        a = Q(_attn(qq, kk, vv, nh, Q))
        a = F.linear(a, Q(g(q + "attention.output.dense.weight")), g(q + "attention.output.dense.bias"))
        h = a * g(q + "layer_scale1.lambda1") + h
        if Q.on:
            gm, bm = _modulated(sd, p + q + "norm2", g(q + "norm2.weight"), g(q + "norm2.bias"), cond)
            f = Q(F.gelu(_ln_linear(h, gm, bm, eps, g(q + "mlp.fc1.weight"), g(q + "mlp.fc1.bias"), Q)))
        else:
            xn = F.layer_norm(h, (H,), g(q + "norm2.weight"), g(q + "norm2.bias"), eps)
            xn = _modulate(xn, sd, p + q + "norm2_modulation.", cond)
            f = F.gelu(F.linear(xn, g(q + "mlp.fc1.weight"), g(q + "mlp.fc1.bias")))
        f = F.linear(f, Q(g(q + "mlp.fc2.weight")), g(q + "mlp.fc2.bias"))
        h = f * g(q + "layer_scale2.lambda1") + h
    return F.layer_norm(h, (H,), g("layernorm.weight"), g("layernorm.bias"), eps)


# ----------------------------------------------------------------------------- two-stream backbone
def _lnp(sd, key):
    return _t(sd[key + "weight"]).float(), _t(sd[key + "bias"]).float()


def _cross_attention(sd, key, z, zn_key, x, xn_key, heads, Q):
    """attn(LN(z), x') with x' = z's own normalised rows (x is None: self-attention), LN(x) (xn_key given) or x as it is.
    bf16 emulation: every LayerNorm folded into the projection that reads it (tsr_ref._ln_linear), queries pre-scaled."""
    w = lambda k: _t(sd[key + k]).float()  # noqa: E731
    wq = w("wq.weight")
    if Q.on:
        qc = LOG2E / math.sqrt(wq.shape[0] // heads)   # see dino_forward: bf16(c q), scores = exponents of 2
        gz, bz = _lnp(sd, zn_key)
        q = Q(_ln_linear(z, gz, bz, 1e-5, wq, None, Q, q_scale=qc))
        if x is None:
            k, v = Q(_ln_linear(z, gz, bz, 1e-5, w("wk.weight"), None, Q)), Q(_ln_linear(z, gz, bz, 1e-5, w("wv.weight"), None, Q))
        elif xn_key is not None:
            gx, bx = _lnp(sd, xn_key)
            k, v = Q(_ln_linear(x, gx, bx, 1e-5, w("wk.weight"), None, Q)), Q(_ln_linear(x, gx, bx, 1e-5, w("wv.weight"), None, Q))
        else:
            k, v = Q(F.linear(Q(x), Q(w("wk.weight")))), Q(F.linear(Q(x), Q(w("wv.weight"))))
    else:
        zn = _ln(sd, zn_key, z)
        xin = zn if x is None else (_ln(sd, xn_key, x) if xn_key is not None else x)
        q, k, v = F.linear(zn, wq), F.linear(xin, w("wk.weight")), F.linear(xin, w("wv.weight"))
    a = Q(_attn(q, k, v, heads, Q))
    return F.linear(a, Q(w("proj.weight")), w("proj.bias"))


def _feed_forward(sd, key, z, zn_key, Q):
    w0, b0 = _t(sd[key + "net.0.proj.weight"]).float(), _t(sd[key + "net.0.proj.bias"]).float()
    if Q.on:
        pr = _ln_linear(z, *_lnp(sd, zn_key), 1e-5, w0, b0, Q)
    else:
        pr = F.linear(_ln(sd, zn_key, z), w0, b0)
    val, gate = pr.chunk(2, dim=-1)
    f = Q(val * F.gelu(gate))
    return F.linear(f, Q(_t(sd[key + "net.2.weight"]).float()), _t(sd[key + "net.2.bias"]).float())


def _ln(sd, key, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), _t(sd[key + "weight"]).float(), _t(sd[key + "bias"]).float(), eps)


def _basic_block(sd, key, z, x, heads, Q):
    z = z + _cross_attention(sd, key + "attn1.", z, key + "norm1.", None, None, heads, Q)
    z = z + _cross_attention(sd, key + "attn2.", z, key + "norm2.", x, None, heads, Q)
    return z + _feed_forward(sd, key + "ff.", z, key + "norm3.", Q)


def _fuse_block(sd, key, z, x, heads, Q, norm_x_input):
    z = z + _cross_attention(sd, key + "attn.", z, key + "norm_z1.", x, key + "norm_x." if norm_x_input else None, heads, Q)
    return z + _feed_forward(sd, key + "ff.", z, key + "norm_z2.", Q)


def backbone_forward(sd, tokens_ct, image_tokens, cfg, prefix="backbone.", bf16=False, collect=None):
    """tokens [C, Nt] (raw triplane tokens), image_tokens [Ni, Ci] -> [C, Nt]."""
    Q = _Q(bf16)
    heads = cfg["num_attention_heads"]
    g = lambda k: _t(sd[prefix + k]).float()  # noqa: E731
    tokens_ct = _t(tokens_ct).float()
    image_tokens = _t(image_tokens).float()
    tri = F.group_norm(tokens_ct[None], cfg.get("norm_num_groups", 32), g("norm_triplane.weight"),
                       g("norm_triplane.bias"), 1e-6)[0].t()
    tri = F.linear(Q(tri), Q(g("proj_triplane.weight")), g("proj_triplane.bias"))
    img = F.linear(Q(_ln(sd, prefix + "norm_image.", image_tokens)), Q(g("proj_image.weight")), g("proj_image.bias"))
    lat = F.linear(Q(_ln(sd, prefix + "norm_latent.", g("latent_init")[0])), Q(g("proj_latent.weight")),
                   g("proj_latent.bias"))
    latent = torch.cat([img, lat], 0)
    nxi = cfg.get("norm_x_input", False)
    for b in range(cfg["num_blocks"]):
        key = prefix + "main_blocks.%d." % b
        latent = _fuse_block(sd, key + "fuse_block_in.", latent, tri, heads, Q, nxi)
        for j in range(cfg["num_basic_blocks"]):
            latent = _basic_block(sd, key + "transformer_block.%d." % j, latent, image_tokens, heads, Q)
        tri = _fuse_block(sd, key + "fuse_block_out.", tri, latent, heads, Q, nxi)
        if collect is not None:
            collect["latent%d" % b] = latent.clone()
            collect["tri%d" % b] = tri.clone()
    out = F.linear(Q(tri), Q(g("proj_out.weight")), g("proj_out.bias"))
    return out.t() + tokens_ct


# ----------------------------------------------------------------------------- pixel-shuffle upsampler
def post_forward(sd, planes, cfg, prefix="post_processor.", bf16=False):
    """planes [3, Ci, S, S] -> [3, Co, S*r, S*r]."""
    Q = _Q(bf16)
    x = _t(planes).float()
    n = cfg.get("conv_layers", 4)
    for i in range(n):
        w = _t(sd[prefix + "upsample.%d.weight" % (2 * i)]).float()
        b = _t(sd[prefix + "upsample.%d.bias" % (2 * i)]).float()
        x = F.conv2d(Q(x), Q(w), b, padding=(w.shape[-1] - 1) // 2)
        if i != n - 1:
            x = F.relu(x)
    return F.pixel_shuffle(x, cfg.get("scale_factor", 4))


# ----------------------------------------------------------------------------- triplane query + MaterialMLP
def query_triplane(positions, planes, radius):
    """positions [N,3] world, planes [3,C,H,W] -> [N, 3C]; plane 0 <- (x,y), 1 <- (x,z), 2 <- (y,z)."""
    planes = _t(planes).float()
    pos = _t(positions).float()
    pos = (pos - (-radius)) / (radius - (-radius))
    pos = pos * (1 - (-1)) + (-1)
    idx = torch.stack((pos[..., [0, 1]], pos[..., [0, 2]], pos[..., [1, 2]]), dim=-3)  # [3, N, 2]
    out = F.grid_sample(planes, idx[:, None], align_corners=True, mode="bilinear")  # [3, C, 1, N]
    return out[:, :, 0].permute(2, 0, 1).reshape(pos.shape[0], -1)


DEFAULT_HEADS = (
    dict(name="density", out_channels=1, out_bias=-1.0, n_hidden_layers=2, output_activation="trunc_exp"),
    dict(name="features", out_channels=3, out_bias=0.0, n_hidden_layers=3, output_activation="sigmoid"),
    dict(name="perturb_normal", out_channels=3, out_bias=0.0, n_hidden_layers=3,
         output_activation="normalize_channel_last"),
    dict(name="vertex_offset", out_channels=3, out_bias=0.0, n_hidden_layers=2, output_activation=None),
)


def decoder_forward(sd, feats, heads=DEFAULT_HEADS, include=None, exclude=None, prefix="decoder."):
    feats = _t(feats).float()
    out = {}
    for h in heads:
        if include is not None and h["name"] not in include:
            continue
        if exclude is not None and h["name"] in exclude:
            continue
        x = feats
        key = prefix + "heads.%s." % h["name"]
        for i in range(h["n_hidden_layers"]):
            x = F.silu(F.linear(x, _t(sd[key + "%d.weight" % (2 * i)]).float(), _t(sd[key + "%d.bias" % (2 * i)]).float()))
        j = 2 * h["n_hidden_layers"]
        x = F.linear(x, _t(sd[key + "%d.weight" % j]).float(), _t(sd[key + "%d.bias" % j]).float()) + h["out_bias"]
        act = h["output_activation"]
        if act == "trunc_exp":
            x = torch.exp(x)
        elif act == "sigmoid":
            x = torch.sigmoid(x)
        elif act == "normalize_channel_last":
            x = F.normalize(x, dim=-1, p=2, eps=1e-7)
        else:
            assert act is None
        out[h["name"]] = x
    return out


# ----------------------------------------------------------------------------- marching tetrahedra
TRIANGLE_TABLE = np.array(
    [[-1, -1, -1, -1, -1, -1], [1, 0, 2, -1, -1, -1], [4, 0, 3, -1, -1, -1], [1, 4, 2, 1, 3, 4],
     [3, 1, 5, -1, -1, -1], [2, 3, 0, 2, 5, 3], [1, 4, 0, 1, 5, 4], [4, 2, 5, -1, -1, -1],
     [4, 5, 2, -1, -1, -1], [4, 1, 0, 4, 5, 1], [3, 2, 0, 3, 5, 2], [1, 3, 5, -1, -1, -1],
     [4, 1, 2, 4, 3, 1], [3, 0, 4, -1, -1, -1], [2, 0, 1, -1, -1, -1], [-1, -1, -1, -1, -1, -1]], dtype=np.int64)
NUM_TRIANGLES = np.array([0, 1, 1, 2, 1, 2, 2, 1, 1, 2, 2, 1, 2, 1, 1, 0], dtype=np.int64)
BASE_TET_EDGES = np.array([0, 1, 0, 2, 0, 3, 1, 2, 1, 3, 2, 3], dtype=np.int64)


def all_edges(indices):
    """Sorted unique undirected edges of the tet grid (isosurface.py:117-131)."""
    e = np.asarray(indices)[:, BASE_TET_EDGES].reshape(-1, 2)
    e = np.sort(e, axis=1)
    return np.unique(e, axis=0)


def deform_grid(grid_vertices, deformation, resolution):
    """grid + (1/resolution) * tanh(offsets)   (isosurface.py:108-115, 211-216; points_range = (0, 1))."""
    gv = _t(grid_vertices).float()
    if deformation is None:
        return gv
    return gv + (1 - 0) / resolution * torch.tanh(_t(deformation).float())


def marching_tets(pos, sdf, tets):
    """pos [Nv,3] fp32 (already deformed), sdf [Nv] or [Nv,1], tets [F,4] int64 -> (verts fp32 [Nm,3], faces int64 [Nf,3]).

    Vertex order = crossing edges in lexicographic (min id, max id) order; face order = all one-triangle tets in
    tet order, then all two-triangle tets in tet order (isosurface.py:142-209)."""
    pos = _t(pos).float().numpy()
    sdf = _t(sdf).float().numpy().reshape(-1)
    tets = np.asarray(tets).astype(np.int64)
    occ = sdf > 0
    occ4 = occ[tets]
    s = occ4.sum(-1)
    valid = (s > 0) & (s < 4)
    vt = tets[valid]
    e = vt[:, BASE_TET_EDGES].reshape(-1, 2)
    e = np.stack([e.min(1), e.max(1)], 1)
    uniq, inv = np.unique(e, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    cross = occ[uniq].sum(-1) == 1
    mapping = -np.ones(uniq.shape[0], dtype=np.int64)
    mapping[cross] = np.arange(cross.sum())
    idx_map = mapping[inv].reshape(-1, 6)
    ev = uniq[cross]
    pa, pb = pos[ev[:, 0]], pos[ev[:, 1]]
    sa, sb = sdf[ev[:, 0]], -sdf[ev[:, 1]]
    den = (sa + sb).astype(np.float32)
    wa = (sb / den).astype(np.float32)[:, None]  # weight of endpoint a = flipped sdf / denominator
    wb = (sa / den).astype(np.float32)[:, None]
    verts = (pa * wa).astype(np.float32) + (pb * wb).astype(np.float32)
    tetindex = (occ4[valid] * (1 << np.arange(4))).sum(-1)
    nt = NUM_TRIANGLES[tetindex]
    one = nt == 1
    two = nt == 2
    f1 = np.take_along_axis(idx_map[one], TRIANGLE_TABLE[tetindex[one]][:, :3], 1).reshape(-1, 3)
    f2 = np.take_along_axis(idx_map[two], TRIANGLE_TABLE[tetindex[two]][:, :6], 1).reshape(-1, 3)
    return verts, np.concatenate([f1, f2], 0)


# ----------------------------------------------------------------------------- texture bake composition
def bake_material(rast, color, perturb_normal, nrm, tng):
    """sf3d/system.py:375-440 on full texel images ([res,res,3]; rast [res,res,4], covered where rast[...,-1] >= 0):
    the reference gathers the covered texels, computes, and scatters into zero images -- same values.
    (Inline code of generate_mesh, no callable to run under the reference: restated, not pinned by a golden.)"""
    rast, color = _t(rast).float(), _t(color).float()
    mask = rast[..., -1] >= 0
    albedo = torch.zeros_like(color)
    albedo[mask] = color[mask]
    if perturb_normal is None:
        return albedo.numpy(), None
    gb_nrm = F.normalize(_t(nrm).float()[mask], dim=-1)
    gb_tng = F.normalize(_t(tng).float()[mask], dim=-1)
    gb_btng = F.normalize(torch.cross(gb_tng, gb_nrm, dim=-1), dim=-1)
    normal = F.normalize(F.normalize(_t(perturb_normal).float()[mask], dim=-1, eps=1e-7), dim=-1)
    dot = lambda a, b: (a * b).sum(-1, keepdim=True)  # noqa: E731
    bump = torch.cat((dot(normal, gb_tng), dot(normal, gb_btng), dot(normal, gb_nrm).clip(0.3, 1)), -1)
    bump = (bump * 0.5 + 0.5).clamp(0, 1)
    out = torch.zeros_like(color)
    out[mask] = bump
    return albedo.numpy(), out.numpy()


# ----------------------------------------------------------------------------- glue
def get_scene_codes(sd, image_hwc, cfg, bf16=False):
    """SF3D.get_scene_codes for one image -> (scene_codes [3,Co,S*r,S*r], direct_codes [3,C,S,S])."""
    with torch.no_grad():
        cam = camera_embedding(sd, "camera_embedder.", cfg["default_distance"], cfg["default_fovy_deg"],
                               cfg["cond_image_size"])
        img_tok = dino_forward(sd, image_hwc, cam, cfg["image_tokenizer"], "image_tokenizer.", bf16)
        t = cfg["tokenizer"]
        C, S = t["num_channels"], t["plane_size"]
        emb = _t(sd["tokenizer.embeddings"]).float()
        tokens = emb.permute(1, 0, 2, 3).reshape(C, 3 * S * S)
        tok = backbone_forward(sd, tokens, img_tok, cfg["backbone"], "backbone.", bf16)
        direct = tok.reshape(C, 3, S, S).permute(1, 0, 2, 3)
        return post_forward(sd, direct, cfg["post_processor"], "post_processor.", bf16), direct


def triplane_to_mesh(sd, scene_code, grid_vertices, tets, cfg, heads=DEFAULT_HEADS):
    """SF3D.triplane_to_meshes for one scene code -> (v_pos world, faces, sdf, deformed grid in [0,1])."""
    with torch.no_grad():
        r = cfg["radius"]
        gv = _t(grid_vertices).float()
        world = (gv - 0) / (1 - 0) * (r - (-r)) + (-r)
        feats = query_triplane(world, scene_code, r)
        dec = decoder_forward(sd, feats, heads, include=["vertex_offset", "density"])
        sdf = dec["density"] - cfg["isosurface_threshold"]
        grid = deform_grid(gv, dec["vertex_offset"], cfg["isosurface_resolution"])
        v, f = marching_tets(grid, sdf, tets)
        v = torch.from_numpy(v)
        v = (v - 0) / (1 - 0) * (r - (-r)) + (-r)
        return v.numpy(), f, sdf.numpy().reshape(-1), grid.numpy()
