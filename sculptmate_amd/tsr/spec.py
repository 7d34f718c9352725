"""Hyper-parameters and the checkpoint tensor inventory of TripoSR (pure Python, no device code).

Constants: /root/reference/TripoSR/checkpoints/config.yaml:1-37 and config.json:1-20.
"""

DEFAULT_CFG = dict(
    cond_image_size=512,
    image_tokenizer=dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                         patch_size=16, image_size=224, layer_norm_eps=1e-12),
    tokenizer=dict(plane_size=32, num_channels=1024),
    backbone=dict(in_channels=1024, num_attention_heads=16, attention_head_dim=64, num_layers=16,
                  cross_attention_dim=768, norm_num_groups=32),
    post_processor=dict(in_channels=1024, out_channels=40),
    decoder=dict(in_channels=120, n_neurons=64, n_hidden_layers=9, activation="silu"),
    renderer=dict(radius=0.87, feature_reduction="concat", density_activation="exp", density_bias=-1.0,
                  num_samples_per_ray=128),
)

IMAGE_MEAN = (0.485, 0.456, 0.406)  # tokenizers/image.py:31-38
IMAGE_STD = (0.229, 0.224, 0.225)


def param_spec(cfg):
    """name -> shape of every tensor of the reference checkpoint (strict load, system.py:64-65)."""
    v, t, b = cfg["image_tokenizer"], cfg["tokenizer"], cfg["backbone"]
    H = v["hidden_size"]
    spec = {}
    p = "image_tokenizer.model."
    n_pos = (v["image_size"] // v["patch_size"]) ** 2 + 1
    spec[p + "embeddings.cls_token"] = (1, 1, H)
    spec[p + "embeddings.position_embeddings"] = (1, n_pos, H)
    spec[p + "embeddings.patch_embeddings.projection.weight"] = (H, 3, v["patch_size"], v["patch_size"])
    spec[p + "embeddings.patch_embeddings.projection.bias"] = (H,)
    for i in range(v["num_hidden_layers"]):
        q = p + "encoder.layer.%d." % i
        for nm in ("query", "key", "value"):
            spec[q + "attention.attention.%s.weight" % nm] = (H, H)
            spec[q + "attention.attention.%s.bias" % nm] = (H,)
        spec[q + "attention.output.dense.weight"] = (H, H)
        spec[q + "attention.output.dense.bias"] = (H,)
        spec[q + "intermediate.dense.weight"] = (v["intermediate_size"], H)
        spec[q + "intermediate.dense.bias"] = (v["intermediate_size"],)
        spec[q + "output.dense.weight"] = (H, v["intermediate_size"])
        spec[q + "output.dense.bias"] = (H,)
        for ln in ("layernorm_before", "layernorm_after"):
            spec[q + ln + ".weight"] = (H,)
            spec[q + ln + ".bias"] = (H,)
    spec[p + "layernorm.weight"] = (H,)
    spec[p + "layernorm.bias"] = (H,)
    spec[p + "pooler.dense.weight"] = (H, H)  # present in the checkpoint, unused (image.py:52)
    spec[p + "pooler.dense.bias"] = (H,)
    C, S = t["num_channels"], t["plane_size"]
    spec["tokenizer.embeddings"] = (3, C, S, S)
    D = b["num_attention_heads"] * b["attention_head_dim"]
    spec["backbone.norm.weight"] = (C,)
    spec["backbone.norm.bias"] = (C,)
    spec["backbone.proj_in.weight"] = (D, C)
    spec["backbone.proj_in.bias"] = (D,)
    for i in range(b["num_layers"]):
        q = "backbone.transformer_blocks.%d." % i
        for ln in ("norm1", "norm2", "norm3"):
            spec[q + ln + ".weight"] = (D,)
            spec[q + ln + ".bias"] = (D,)
        for nm, kd in (("attn1", D), ("attn2", b["cross_attention_dim"])):
            spec[q + nm + ".to_q.weight"] = (D, D)
            spec[q + nm + ".to_k.weight"] = (D, kd)
            spec[q + nm + ".to_v.weight"] = (D, kd)
            spec[q + nm + ".to_out.0.weight"] = (D, D)
            spec[q + nm + ".to_out.0.bias"] = (D,)
        spec[q + "ff.net.0.proj.weight"] = (8 * D, D)
        spec[q + "ff.net.0.proj.bias"] = (8 * D,)
        spec[q + "ff.net.2.weight"] = (D, 4 * D)
        spec[q + "ff.net.2.bias"] = (D,)
    spec["backbone.proj_out.weight"] = (C, D)
    spec["backbone.proj_out.bias"] = (C,)
    pp = cfg["post_processor"]
    spec["post_processor.upsample.weight"] = (pp["in_channels"], pp["out_channels"], 2, 2)
    spec["post_processor.upsample.bias"] = (pp["out_channels"],)
    d = cfg["decoder"]
    dims = [d["in_channels"]] + [d["n_neurons"]] * d["n_hidden_layers"] + [4]
    for i in range(len(dims) - 1):
        spec["decoder.layers.%d.weight" % (2 * i)] = (dims[i + 1], dims[i])
        spec["decoder.layers.%d.bias" % (2 * i)] = (dims[i + 1],)
    return spec


def make_cfg(vit_hidden=768, vit_layers=12, vit_heads=12, vit_mlp=3072, channels=1024, plane_size=32,
             heads=16, head_dim=64, layers=16, cross_dim=None, cond_image_size=512, groups=32):
    """A TSR config with the reference architecture at other sizes (tests use small ones)."""
    import copy

    cfg = copy.deepcopy(DEFAULT_CFG)
    cfg["cond_image_size"] = cond_image_size
    cfg["image_tokenizer"].update(hidden_size=vit_hidden, num_hidden_layers=vit_layers,
                                  num_attention_heads=vit_heads, intermediate_size=vit_mlp)
    cfg["tokenizer"].update(plane_size=plane_size, num_channels=channels)
    cfg["backbone"].update(in_channels=channels, num_attention_heads=heads, attention_head_dim=head_dim,
                           num_layers=layers, cross_attention_dim=cross_dim or vit_hidden, norm_num_groups=groups)
    cfg["post_processor"].update(in_channels=channels)
    return cfg


# kernels need widths that are multiples of 256: a small but kernel-compatible model for GPU tests
SMALL_CFG = make_cfg(vit_hidden=256, vit_layers=2, vit_heads=4, vit_mlp=512, channels=256, plane_size=8,
                     heads=4, head_dim=64, layers=2, cond_image_size=128)
# arbitrary small widths: only for oracle-vs-reference goldens (never run through the HIP kernels)
TINY_CFG = make_cfg(vit_hidden=64, vit_layers=2, vit_heads=2, vit_mlp=128, channels=64, plane_size=4,
                    heads=2, head_dim=32, layers=2, cross_dim=64, cond_image_size=64)
