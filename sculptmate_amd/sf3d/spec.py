"""Hyper-parameters and the checkpoint tensor inventory of StableFast-3D (pure Python, no device code).

Constants: /root/reference/StableFast/checkpoints/config.yaml:1-96, SF3D.Config defaults
(StableFast/sf3d/system.py:44-74), facebook/dinov2-large (hidden 1024, 24 layers, 16 heads, patch 14, image 518,
mlp_ratio 4, layer_norm_eps 1e-6, layerscale).  The CLIP-based image estimator and the illumination estimator have their
own inventories in estimators.py (image_estimator_param_spec / global_estimator_param_spec); SF3D.load_state_dict builds
them when the checkpoint carries their weights, so their prefixes are not "unexpected" here.
"""
import copy

HEADS = (
    dict(name="density", out_channels=1, out_bias=-1.0, n_hidden_layers=2, output_activation="trunc_exp"),
    dict(name="features", out_channels=3, out_bias=0.0, n_hidden_layers=3, output_activation="sigmoid"),
    dict(name="perturb_normal", out_channels=3, out_bias=0.0, n_hidden_layers=3,
         output_activation="normalize_channel_last"),
    dict(name="vertex_offset", out_channels=3, out_bias=0.0, n_hidden_layers=2, output_activation=None),
)

DEFAULT_CFG = dict(
    cond_image_size=512, isosurface_resolution=160, isosurface_threshold=10.0, radius=0.87,
    background_color=(0.5, 0.5, 0.5), default_fovy_deg=40.0, default_distance=1.6,
    camera_embedder=dict(in_channels=25, out_channels=768),
    image_tokenizer=dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, mlp_ratio=4, patch_size=14,
                         image_size=518, layer_norm_eps=1e-6, modulation_cond_dim=768),
    tokenizer=dict(plane_size=96, num_channels=1024),
    backbone=dict(num_attention_heads=16, attention_head_dim=64, raw_triplane_channels=1024, triplane_channels=1024,
                  raw_image_channels=1024, num_latents=1792, num_blocks=4, num_basic_blocks=3, norm_num_groups=32,
                  norm_x_input=False, cross_attention_dim=1024),
    post_processor=dict(in_channels=1024, out_channels=40, scale_factor=4, conv_layers=4),
    decoder=dict(in_channels=120, n_neurons=64, activation="silu", heads=HEADS),
    image_estimator=None,    # None -> estimators.IMAGE_ESTIMATOR_CFG (config.yaml:67-85)
    global_estimator=None,   # None -> estimators.GLOBAL_ESTIMATOR_CFG (config.yaml:87-96)
)

IGNORED_PREFIXES = ("image_estimator.", "global_estimator.", "image_tokenizer.modulations.", "bbox",
                    "image_tokenizer.model.embeddings.mask_token")


def param_spec(cfg):
    """name -> shape of every tensor the geometry/texture path reads (reference state_dict names)."""
    v, t, b, pp, d = cfg["image_tokenizer"], cfg["tokenizer"], cfg["backbone"], cfg["post_processor"], cfg["decoder"]
    spec = {}
    cam = cfg["camera_embedder"]
    spec["camera_embedder.linear.weight"] = (cam["out_channels"], cam["in_channels"])
    spec["camera_embedder.linear.bias"] = (cam["out_channels"],)
    H, P = v["hidden_size"], v["patch_size"]
    M = int(H * v["mlp_ratio"])
    p = "image_tokenizer.model."
    n_pos = (v["image_size"] // P) ** 2 + 1
    spec[p + "embeddings.cls_token"] = (1, 1, H)
    spec[p + "embeddings.position_embeddings"] = (1, n_pos, H)
    spec[p + "embeddings.patch_embeddings.projection.weight"] = (H, 3, P, P)
    spec[p + "embeddings.patch_embeddings.projection.bias"] = (H,)
    for i in range(v["num_hidden_layers"]):
        q = p + "encoder.layer.%d." % i
        for nm in ("query", "key", "value"):
            spec[q + "attention.attention.%s.weight" % nm] = (H, H)
            spec[q + "attention.attention.%s.bias" % nm] = (H,)
        spec[q + "attention.output.dense.weight"] = (H, H)
        spec[q + "attention.output.dense.bias"] = (H,)
        spec[q + "layer_scale1.lambda1"] = (H,)
        spec[q + "layer_scale2.lambda1"] = (H,)
        spec[q + "mlp.fc1.weight"] = (M, H)
        spec[q + "mlp.fc1.bias"] = (M,)
        spec[q + "mlp.fc2.weight"] = (H, M)
        spec[q + "mlp.fc2.bias"] = (H,)
        for ln in ("norm1", "norm2"):
            spec[q + ln + ".weight"] = (H,)
            spec[q + ln + ".bias"] = (H,)
            spec[q + ln + "_modulation.linear2.weight"] = (2 * H, v["modulation_cond_dim"])
            spec[q + ln + "_modulation.linear2.bias"] = (2 * H,)
    spec[p + "layernorm.weight"] = (H,)
    spec[p + "layernorm.bias"] = (H,)
    C, S = t["num_channels"], t["plane_size"]
    spec["tokenizer.embeddings"] = (3, C, S, S)
    D = b["num_attention_heads"] * b["attention_head_dim"]
    Ct, Ci = b["triplane_channels"], b["raw_image_channels"]
    assert Ct == D, "the two streams share one width in every shipped config"
    q = "backbone."
    spec[q + "latent_init"] = (1, b["num_latents"], D)
    spec[q + "norm_triplane.weight"] = (b["raw_triplane_channels"],)
    spec[q + "norm_triplane.bias"] = (b["raw_triplane_channels"],)
    spec[q + "proj_triplane.weight"] = (Ct, b["raw_triplane_channels"])
    spec[q + "proj_triplane.bias"] = (Ct,)
    spec[q + "norm_image.weight"] = (Ci,)
    spec[q + "norm_image.bias"] = (Ci,)
    spec[q + "proj_image.weight"] = (D, Ci)
    spec[q + "proj_image.bias"] = (D,)
    spec[q + "norm_latent.weight"] = (D,)
    spec[q + "norm_latent.bias"] = (D,)
    spec[q + "proj_latent.weight"] = (D, D)
    spec[q + "proj_latent.bias"] = (D,)

    def attn(key, dim, kv):
        spec[key + "wq.weight"] = (dim, dim)
        spec[key + "wk.weight"] = (dim, kv)
        spec[key + "wv.weight"] = (dim, kv)
        spec[key + "proj.weight"] = (dim, dim)
        spec[key + "proj.bias"] = (dim,)

    def ff(key, dim):
        spec[key + "net.0.proj.weight"] = (8 * dim, dim)
        spec[key + "net.0.proj.bias"] = (8 * dim,)
        spec[key + "net.2.weight"] = (dim, 4 * dim)
        spec[key + "net.2.bias"] = (dim,)

    def ln(key, dim):
        spec[key + "weight"] = (dim,)
        spec[key + "bias"] = (dim,)

    for i in range(b["num_blocks"]):
        k = q + "main_blocks.%d." % i
        for name, dz, dx in (("fuse_block_in.", D, Ct), ("fuse_block_out.", Ct, D)):
            if b.get("norm_x_input", False):
                ln(k + name + "norm_x.", dx)
            attn(k + name + "attn.", dz, dx)
            ln(k + name + "norm_z1.", dz)
            ln(k + name + "norm_z2.", dz)
            ff(k + name + "ff.", dz)
        for j in range(b["num_basic_blocks"]):
            kk = k + "transformer_block.%d." % j
            ln(kk + "norm1.", D)
            attn(kk + "attn1.", D, D)
            ln(kk + "norm2.", D)
            attn(kk + "attn2.", D, b["cross_attention_dim"])
            ln(kk + "norm3.", D)
            ff(kk + "ff.", D)
    spec[q + "proj_out.weight"] = (b["raw_triplane_channels"], Ct)
    spec[q + "proj_out.bias"] = (b["raw_triplane_channels"],)
    cin = pp["in_channels"]
    for i in range(pp["conv_layers"]):
        cout = cin if i != pp["conv_layers"] - 1 else pp["out_channels"] * pp["scale_factor"] ** 2
        spec["post_processor.upsample.%d.weight" % (2 * i)] = (cout, cin, 3, 3)
        spec["post_processor.upsample.%d.bias" % (2 * i)] = (cout,)
    for h in d["heads"]:
        key = "decoder.heads.%s." % h["name"]
        dims = [d["in_channels"]] + [d["n_neurons"]] * h["n_hidden_layers"] + [h["out_channels"]]
        for i in range(len(dims) - 1):
            spec[key + "%d.weight" % (2 * i)] = (dims[i + 1], dims[i])
            spec[key + "%d.bias" % (2 * i)] = (dims[i + 1],)
    return spec


def make_cfg(width=1024, heads=16, dino_layers=24, plane_size=96, num_latents=1792, num_blocks=4, num_basic_blocks=3,
             cond_image_size=512, image_size=518, isosurface_resolution=160, cam_dim=768):
    """The reference architecture at other sizes (tests use small ones; widths must be multiples of 128 and
    heads*64 == width for the HIP kernels)."""
    cfg = copy.deepcopy(DEFAULT_CFG)
    cfg.update(cond_image_size=cond_image_size, isosurface_resolution=isosurface_resolution)
    cfg["camera_embedder"].update(out_channels=cam_dim)
    cfg["image_tokenizer"].update(hidden_size=width, num_hidden_layers=dino_layers, num_attention_heads=heads,
                                  image_size=image_size, modulation_cond_dim=cam_dim)
    cfg["tokenizer"].update(plane_size=plane_size, num_channels=width)
    cfg["backbone"].update(num_attention_heads=heads, raw_triplane_channels=width, triplane_channels=width,
                           raw_image_channels=width, num_latents=num_latents, num_blocks=num_blocks,
                           num_basic_blocks=num_basic_blocks, cross_attention_dim=width)
    cfg["post_processor"].update(in_channels=width)
    return cfg


# small but kernel-compatible (GPU tests): 4 heads x 64, 2 DINO layers, 8x8 planes, 56-pixel input (4x4 patches)
SMALL_CFG = make_cfg(width=256, heads=4, dino_layers=2, plane_size=8, num_latents=32, num_blocks=2, num_basic_blocks=2,
                     cond_image_size=56, image_size=70, isosurface_resolution=24, cam_dim=64)
