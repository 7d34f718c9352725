"""Golden vectors for StableFast-3D's two estimators, produced by IMPORTING the reference
(build container only: needs /root/reference):  python tests/golden/make_sf3d_est_goldens.py

  sf3d_global_est.npz  MultiHeadEstimator.forward (sf3d/models/global_estimator/multi_head_estimator.py:86-118):
                       two stride-2 3x3 convolutions over the concatenated triplane, max pool, MLP heads.  Small widths,
                       seeded weights in the fixture.
  sf3d_image_est.npz   ClipBasedHeadEstimator.forward (sf3d/models/image_estimator/clip_based_estimator.py:88-168):
                       the bilinear 512 -> 224 resize, the heads (hidden_features = 128 to keep the fixture small) and the Beta-mode read-out
                       on a given CLIP feature vector.  open_clip and torchvision are not installed: the
                       stand-in `open_clip` model's encode_image() returns the feature vector stored in the fixture
                       and records the tensor it was handed; the stand-in torchvision Normalize is
                       (x - mean[c]) / std[c] (torchvision's documented behaviour; the only arithmetic any stand-in
                       here performs) with open_clip's published OPENAI_DATASET_MEAN / _STD constants.
  sf3d_clip.npz        The CLIP ViT visual tower itself lives in open_clip (absent) -> the oracle restates the
                       published architecture and is pinned to an INDEPENDENT implementation of it that is installed:
                       transformers' CLIPVisionModelWithProjection (hidden_act="gelu", as open_clip's ViT-B-32 config
                       has no quick_gelu), small widths, weights renamed to open_clip's parameter names
                       (visual.transformer.resblocks.N.attn.in_proj_weight = cat(q, k, v), visual.proj = W^T, ...);
                       the weights are bf16-representable and stored as their 16-bit patterns ("wb." keys).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _reference_shims as shims  # noqa: E402

shims.install()
import make_reference_goldens as mrg  # noqa: E402

mrg._sf3d_shims()
from make_sf3d_goldens import randomize, sd_np  # noqa: E402

OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)
torch.set_num_threads(8)


def make_global():
    from sf3d.models.global_estimator.multi_head_estimator import MultiHeadEstimator

    cfg = dict(triplane_features=8, n_layers=2, hidden_features=16, activation="relu", pool="max",
               heads=[dict(name="sg_amplitudes", out_channels=6, n_hidden_layers=3, output_activation="softplus",
                           output_bias=1.0, add_to_decoder_features=False, shape=[-1, 6, 1]),
                      dict(name="tint", out_channels=3, n_hidden_layers=1, output_activation="sigmoid",
                           output_bias=0.0, add_to_decoder_features=True, shape=None)])
    torch.manual_seed(0)
    est = MultiHeadEstimator(cfg).eval()
    randomize(est, 21)
    x = torch.randn(2, 3, 8, 11, 11, generator=torch.Generator().manual_seed(22))
    with torch.no_grad():
        out = est(x)
    fx = {"triplane": x.numpy()}
    for k, v in out.items():
        fx["out." + k] = v.numpy()
    fx.update(sd_np(est, "w."))
    np.savez_compressed(os.path.join(HERE, "sf3d_global_est.npz"), **fx)
    print("sf3d_global_est:", {k: tuple(v.shape) for k, v in out.items()})


class _Recorder:
    seen = {}


def _install_clip_stand_ins(features):
    oc = types.ModuleType("open_clip")
    oc.constants = types.ModuleType("open_clip.constants")
    oc.constants.OPENAI_DATASET_MEAN = OPENAI_DATASET_MEAN
    oc.constants.OPENAI_DATASET_STD = OPENAI_DATASET_STD

    class _Model(torch.nn.Module):
        def encode_image(self, image):
            _Recorder.seen["encode_image_input"] = image.detach().clone()
            return features.clone()

    oc.create_model_and_transforms = lambda model, pretrained=None: (_Model(), None, None)
    sys.modules["open_clip"] = oc
    sys.modules["open_clip.constants"] = oc.constants

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = torch.tensor(mean).view(1, 3, 1, 1), torch.tensor(std).view(1, 3, 1, 1)

        def __call__(self, x):
            _Recorder.seen["normalize_input"] = x.detach().clone()
            return (x - self.mean) / self.std

    tv = types.ModuleType("torchvision")
    tv.transforms = types.ModuleType("torchvision.transforms")
    tv.transforms.Normalize = Normalize
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tv.transforms


def make_image():
    g = torch.Generator().manual_seed(31)
    B, HF = 1, 128
    features = torch.randn(B, HF, generator=g)
    _install_clip_stand_ins(features)
    for m in [k for k in sys.modules if k.startswith("sf3d.models.image_estimator")]:
        del sys.modules[m]
    from sf3d.models.image_estimator.clip_based_estimator import ClipBasedHeadEstimator

    head = dict(out_channels=1, n_hidden_layers=3, output_activation="linear", add_to_decoder_features=True,
                output_bias=1.0, shape=[-1, 1, 1])
    cfg = dict(model="ViT-B-32", pretrain="laion2b_s34b_b79k", distribution="beta", distribution_eval="mode",
               activation="relu", hidden_features=HF,
               heads=[dict(name="roughness", **head), dict(name="metallic", **head)])
    torch.manual_seed(0)
    est = ClipBasedHeadEstimator(cfg).eval()
    randomize(est.heads, 32)
    # smooth image + hard mask like the add-on's input (rgb_cond * mask_cond, system.py:326-329)
    S = 512
    yy, xx = torch.meshgrid(torch.linspace(0, 1, S), torch.linspace(0, 1, S), indexing="ij")
    base = torch.stack([0.5 + 0.5 * torch.sin(9 * xx + 3 * yy), xx * yy, 0.5 + 0.5 * torch.cos(7 * yy - 2 * xx)], -1)
    img = (base[None, None] * torch.rand(B, 1, 1, 1, 3, generator=g) + 0.05 * torch.rand(B, 1, S, S, 3, generator=g)).clamp(0, 1)
    img8 = (img * 255).round().to(torch.uint8)                       # stored as 8-bit: k/255 is what the add-on feeds anyway
    img = img8.float() / 255.0
    mask = (((xx - 0.5) ** 2 + (yy - 0.45) ** 2) < 0.16).float()[None, None, :, :, None].expand(B, 1, S, S, 1)
    cond = img * mask
    with torch.no_grad():
        out = est(cond)
    fx = {"rgb_u8": img8[:, 0].numpy(), "mask": mask[:, 0, :, :, 0].numpy().astype(np.uint8), "features": features.numpy(),
          "resized": _Recorder.seen["normalize_input"].numpy(),            # [B,3,224,224], the reference's F.interpolate
          "clip_input_sample": _Recorder.seen["encode_image_input"].numpy()[:, :, ::7, ::7]}
    for name in ("roughness", "metallic"):
        fx["out.decoder_" + name] = out["decoder_" + name].numpy()
        d = out[name + "_dist"]
        fx["alpha." + name], fx["beta." + name] = d.concentration1.numpy(), d.concentration0.numpy()
    fx.update(sd_np(est.heads, "w.heads."))
    np.savez_compressed(os.path.join(HERE, "sf3d_image_est.npz"), **fx)
    print("sf3d_image_est:", {k: (v.shape, float(v.reshape(-1)[0])) for k, v in fx.items() if k.startswith("out.")})


def make_clip():
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection

    W, L, NH, P, S, E, MLP = 256, 2, 4, 8, 32, 128, 512     # LayerNorm kernel: width % 256 == 0; attention: head width 64
    cfg = CLIPVisionConfig(hidden_size=W, intermediate_size=MLP, num_hidden_layers=L, num_attention_heads=NH, image_size=S,
                           patch_size=P, projection_dim=E, hidden_act="gelu", layer_norm_eps=1e-5, attention_dropout=0.0)
    torch.manual_seed(0)
    m = CLIPVisionModelWithProjection(cfg).eval()
    randomize(m, 41)
    with torch.no_grad():
        m.vision_model.embeddings.position_embedding.weight.copy_(0.3 * torch.randn(m.vision_model.embeddings.position_embedding.weight.shape,
                                                                                   generator=torch.Generator().manual_seed(42)))
    with torch.no_grad():   # weights are bf16-representable so that the fixture can hold them as 16-bit patterns (half the size)
        for p_ in m.parameters():
            p_.copy_(p_.to(torch.bfloat16).float())
    x = torch.randn(2, 3, S, S, generator=torch.Generator().manual_seed(43))
    with torch.no_grad():
        r = m(pixel_values=x, output_hidden_states=True)
    hf = {k: v.detach() for k, v in m.state_dict().items()}
    v = "vision_model."
    oc = {"visual.class_embedding": hf[v + "embeddings.class_embedding"],
          "visual.positional_embedding": hf[v + "embeddings.position_embedding.weight"],
          "visual.conv1.weight": hf[v + "embeddings.patch_embedding.weight"],
          "visual.ln_pre.weight": hf[v + "pre_layrnorm.weight"], "visual.ln_pre.bias": hf[v + "pre_layrnorm.bias"],
          "visual.ln_post.weight": hf[v + "post_layernorm.weight"], "visual.ln_post.bias": hf[v + "post_layernorm.bias"],
          "visual.proj": hf["visual_projection.weight"].t().contiguous()}
    for i in range(L):
        s, d = v + "encoder.layers.%d." % i, "visual.transformer.resblocks.%d." % i
        oc[d + "attn.in_proj_weight"] = torch.cat([hf[s + "self_attn.%s_proj.weight" % n] for n in "qkv"], 0)
        oc[d + "attn.in_proj_bias"] = torch.cat([hf[s + "self_attn.%s_proj.bias" % n] for n in "qkv"], 0)
        for a, b in (("attn.out_proj", "self_attn.out_proj"), ("ln_1", "layer_norm1"), ("ln_2", "layer_norm2"),
                     ("mlp.c_fc", "mlp.fc1"), ("mlp.c_proj", "mlp.fc2")):
            oc[d + a + ".weight"], oc[d + a + ".bias"] = hf[s + b + ".weight"], hf[s + b + ".bias"]
    fx = {"image": x.numpy(), "out": r.image_embeds.numpy(), "hidden_last": r.hidden_states[-1].numpy(),
          "cfg": np.array([W, L, NH, P, S, E, MLP], np.int64)}
    for k, t in oc.items():     # "wb.<name>": the upper 16 bits of the float32 pattern (exact, see above)
        a = t.contiguous().numpy()
        assert np.array_equal((a.view(np.uint32) & 0xFFFF), np.zeros(a.shape, np.uint32))
        fx["wb." + k] = (a.view(np.uint32) >> 16).astype(np.uint16)
    np.savez_compressed(os.path.join(HERE, "sf3d_clip.npz"), **fx)
    print("sf3d_clip:", r.image_embeds.shape, len(oc), "tensors")


if __name__ == "__main__":
    for w in sys.argv[1:] or ["global", "image", "clip"]:
        globals()["make_" + w]()
