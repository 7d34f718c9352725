"""Host-side mirror of the reference interface (no GPU): spec, config, facade return codes."""
import os

import numpy as np
import pytest
import torch

from sculptmate_amd import synth
from sculptmate_amd.tsr.spec import DEFAULT_CFG, SMALL_CFG, param_spec


def test_checkpoint_inventory_matches_reference_counts():
    """549 tensors / 419 275 628 parameters measured on the reference."""
    spec = param_spec(DEFAULT_CFG)
    assert len(spec) == 549
    assert sum(int(np.prod(s)) for s in spec.values()) == 419275628
    assert spec["backbone.transformer_blocks.15.ff.net.0.proj.weight"] == (8192, 1024)
    assert spec["image_tokenizer.model.encoder.layer.11.attention.attention.query.weight"] == (768, 768)
    assert spec["post_processor.upsample.weight"] == (1024, 40, 2, 2)
    assert spec["decoder.layers.18.weight"] == (4, 64)


def test_synthetic_state_covers_the_inventory():
    sd = synth.tsr_state(0, SMALL_CFG)
    spec = param_spec(SMALL_CFG)
    assert set(sd) == set(spec)
    assert all(tuple(sd[k].shape) == tuple(spec[k]) and sd[k].dtype == np.float32 for k in spec)


def test_strict_state_dict_loading():
    from sculptmate_amd.tsr import TSR

    sd = synth.tsr_state(0, SMALL_CFG)
    m = TSR(SMALL_CFG)
    m.load_state_dict(sd)
    bad = dict(sd)
    bad.pop("backbone.proj_in.bias")
    with pytest.raises(RuntimeError):
        TSR(SMALL_CFG).load_state_dict(bad)
    bad = dict(sd)
    bad["extra.weight"] = np.zeros(1, np.float32)
    with pytest.raises(RuntimeError):
        TSR(SMALL_CFG).load_state_dict(bad)
    bad = dict(sd)
    bad["backbone.proj_in.bias"] = np.zeros(3, np.float32)
    with pytest.raises(RuntimeError):
        TSR(SMALL_CFG).load_state_dict(bad)


def test_reference_config_files_parse(tmp_path):
    from sculptmate_amd.tsr import load_config

    (tmp_path / "config.yaml").write_text(
        "cond_image_size: 512\ntokenizer:\n  plane_size: 32\n  num_channels: 1024\n"
        "backbone:\n  in_channels: ${tokenizer.num_channels}\n  num_attention_heads: 16\n  attention_head_dim: 64\n"
        "  num_layers: 16\n  cross_attention_dim: 768\npost_processor:\n  in_channels: 1024\n  out_channels: 40\n"
        "decoder:\n  in_channels: 120\n  n_neurons: 64\n  n_hidden_layers: 9\n  activation: silu\n"
        "renderer:\n  radius: 0.87\n  feature_reduction: concat\n  density_activation: exp\n  density_bias: -1.0\n"
        "  num_samples_per_ray: 128\n")
    (tmp_path / "config.json").write_text('{"hidden_size": 768, "num_hidden_layers": 12, "num_attention_heads": 12, '
                                          '"intermediate_size": 3072, "patch_size": 16, "image_size": 224, "layer_norm_eps": 1e-12}')
    cfg = load_config(str(tmp_path / "config.yaml"), str(tmp_path / "config.json"))
    assert param_spec(cfg) == param_spec(DEFAULT_CFG)
    assert cfg["renderer"]["radius"] == 0.87


def test_generator_facade_return_codes(tmp_path):
    """TripoGenerator: 1 = model not loaded, 2 = initialisation error (generate.py:17-43)."""
    from sculptmate_amd.generate import TripoGenerator

    g = TripoGenerator(torch.device("cpu"))
    assert g.chunk_size == 8192 and g.mc_resolution == 256 and g.model is None
    assert g.precision == "bf16"   # the one attribute beyond the reference's: the transformer's arithmetic
    assert g.generate_mesh(np.zeros((512, 512, 3), np.float32)) == 1
    g.checkpoint_dir = str(tmp_path / "nope")
    assert g.initiate_model() == 2
    assert g.model is None


def test_generator_facade_precision_from_environment(monkeypatch):
    from sculptmate_amd.generate import TripoGenerator

    monkeypatch.setenv("SCULPT_PRECISION", "bf16l3")
    assert TripoGenerator(torch.device("cpu")).precision == "bf16l3"
    monkeypatch.delenv("SCULPT_PRECISION")
    assert TripoGenerator(torch.device("cpu")).precision == "bf16"


def test_from_pretrained_passes_model_arguments(tmp_path):
    from sculptmate_amd.tsr import TSR

    _write_checkpoint(str(tmp_path), SMALL_CFG, seed=5)
    m = TSR.from_pretrained(str(tmp_path), "config.yaml", "model.ckpt", precision="bf16l3", decoder_filter=False)
    assert m.precision == "bf16l3" and m.decoder_filter is False and m.decoder_precision == "bf16l3"
    assert TSR.from_pretrained(str(tmp_path), "config.yaml", "model.ckpt").precision == "bf16"
    with pytest.raises(ValueError):
        TSR.from_pretrained(str(tmp_path), "config.yaml", "model.ckpt", precision="fp8")


def test_tsr_refuses_cpu_device():
    from sculptmate_amd._lib import SculptError
    from sculptmate_amd.tsr import TSR

    with pytest.raises(SculptError):
        TSR(SMALL_CFG).to("cpu")


def test_from_pretrained_missing_dir():
    from sculptmate_amd.tsr import TSR

    with pytest.raises(FileNotFoundError):
        TSR.from_pretrained("/nonexistent/dir", "config.yaml", "model.ckpt")


def test_scale_tensor_and_renderer_chunk_api():
    from sculptmate_amd.tsr import TriplaneNeRFRenderer
    from sculptmate_amd.tsr.utils import scale_tensor

    x = torch.tensor([0.0, 0.5, 1.0])
    y = scale_tensor(x, (0, 1), (-0.87, 0.87))
    assert torch.equal(y, x * (0.87 - (-0.87)) + (-0.87))
    r = TriplaneNeRFRenderer(DEFAULT_CFG["renderer"])
    r.set_chunk_size(8192)
    with pytest.raises(AssertionError):
        r.set_chunk_size(-1)


def _write_checkpoint(dirpath, cfg, seed=5):
    """A checkpoint directory laid out like the reference's TripoSR/checkpoints (config.yaml, config.json, model.ckpt)."""
    import json

    import yaml

    sd = synth.tsr_state(seed, cfg)
    torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, os.path.join(dirpath, "model.ckpt"))
    b, t = cfg["backbone"], cfg["tokenizer"]
    y = {"cond_image_size": cfg["cond_image_size"], "tokenizer": dict(t),
         "backbone": {"in_channels": "${tokenizer.num_channels}", "num_attention_heads": b["num_attention_heads"],
                      "attention_head_dim": b["attention_head_dim"], "num_layers": b["num_layers"],
                      "cross_attention_dim": b["cross_attention_dim"]},
         "post_processor": dict(cfg["post_processor"]), "decoder": dict(cfg["decoder"]), "renderer": dict(cfg["renderer"])}
    with open(os.path.join(dirpath, "config.yaml"), "w") as f:
        yaml.safe_dump(y, f)
    with open(os.path.join(dirpath, "config.json"), "w") as f:
        json.dump(cfg["image_tokenizer"], f)
    return sd


def test_from_pretrained_reads_reference_style_checkpoint(tmp_path):
    from sculptmate_amd.tsr import TSR

    sd = _write_checkpoint(str(tmp_path), SMALL_CFG)
    m = TSR.from_pretrained(str(tmp_path), config_name="config.yaml", weight_name="model.ckpt")
    assert param_spec(m.cfg) == param_spec(SMALL_CFG)
    got = m.state_dict()
    assert set(got) == set(sd) and all(np.array_equal(got[k].numpy(), sd[k]) for k in sd)


def test_obj_round_trip(tmp_path):
    from sculptmate_amd import meshio

    rng = np.random.default_rng(0)
    v = rng.standard_normal((50, 3)).astype(np.float32)
    f = rng.integers(0, 50, (80, 3)).astype(np.int64)
    c = rng.random((50, 3)).astype(np.float32)
    meshio.write_obj(str(tmp_path / "m.obj"), v, f, c)
    v2, f2, c2 = meshio.read_obj(str(tmp_path / "m.obj"))
    assert np.array_equal(f2, f) and np.allclose(v2, v, rtol=1e-6) and np.allclose(c2, c, atol=1e-5)
    meshio.write_obj(str(tmp_path / "n.obj"), v, f)
    assert meshio.read_obj(str(tmp_path / "n.obj"))[2] is None


def _random_mesh(n_v=60, n_f=90, seed=1):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((n_v, 3)).astype(np.float32), rng.integers(0, n_v, (n_f, 3)).astype(np.int64),
            rng.random((n_v, 3)).astype(np.float32), rng)


def test_png_codec_round_trip():
    from sculptmate_amd import meshio

    rng = np.random.default_rng(2)
    for shape in [(5, 7, 3), (16, 16, 4), (3, 9, 1)]:
        img = rng.integers(0, 256, shape).astype(np.uint8)
        png = meshio.encode_png(img)
        assert np.array_equal(meshio.decode_png(png), img)
    PIL = pytest.importorskip("PIL.Image")
    import io

    img = rng.integers(0, 256, (11, 13, 3)).astype(np.uint8)
    assert np.array_equal(np.array(PIL.open(io.BytesIO(meshio.encode_png(img)))), img)   # an independent decoder agrees


def test_glb_round_trip(tmp_path):
    import json
    import struct

    from sculptmate_amd import meshio

    v, f, c, rng = _random_mesh()
    uv = (rng.integers(0, 1025, (len(v), 2)) / 1024.0).astype(np.float32)   # 1 - v is exact on this lattice
    nrm = rng.standard_normal((len(v), 3)).astype(np.float32)
    tex = rng.integers(0, 256, (32, 32, 3)).astype(np.uint8)
    bump = rng.integers(0, 256, (32, 32, 3)).astype(np.uint8)
    path = str(tmp_path / "m.glb")
    meshio.write_glb(path, v, f, vertex_colors=c, normals=nrm, uvs=uv, basecolor_tex=tex, normal_tex=bump,
                     roughness=np.float32(0.25), metallic=[0.75])
    got = meshio.read_glb(path)
    assert np.array_equal(got["vertices"], v) and np.array_equal(got["faces"], f)
    assert np.array_equal(got["vertex_colors"], c) and np.array_equal(got["uvs"], uv) and np.array_equal(got["normals"], nrm)
    assert np.array_equal(got["basecolor_tex"], tex) and np.array_equal(got["normal_tex"], bump)
    assert got["roughness"] == 0.25 and got["metallic"] == 0.75
    # glTF's origin is the top-left of the image: the stored TEXCOORD_0 is (u, 1 - v)
    stored = meshio.read_glb(path, uv_origin="top_left")["uvs"]
    assert np.array_equal(stored[:, 0], uv[:, 0]) and np.array_equal(stored[:, 1], 1.0 - uv[:, 1])
    meshio.write_sf3d_glb(str(tmp_path / "s.glb"), dict(vertices=v, faces=f, uvs=uv, basecolor_tex=np.dstack([tex, tex[..., :1]]),
                                                        bump_tex=None, roughness=0.5, metallic=0.0))
    s = meshio.read_glb(str(tmp_path / "s.glb"))
    assert np.array_equal(s["basecolor_tex"], tex) and s["normal_tex"] is None and np.array_equal(s["uvs"], uv)
    # container-level checks a strict loader makes: 4-byte alignment, declared lengths, POSITION bounds
    raw = open(path, "rb").read()
    assert len(raw) % 4 == 0 and struct.unpack("<I", raw[8:12])[0] == len(raw)
    n_json = struct.unpack("<I", raw[12:16])[0]
    doc = json.loads(raw[20:20 + n_json])
    assert n_json % 4 == 0 and all(bv["byteOffset"] % 4 == 0 for bv in doc["bufferViews"])
    pos = doc["accessors"][doc["meshes"][0]["primitives"][0]["attributes"]["POSITION"]]
    assert np.allclose(pos["min"], v.min(0)) and np.allclose(pos["max"], v.max(0))
    # geometry only, and the bad-index error
    meshio.write_glb(str(tmp_path / "g.glb"), v, f)
    bare = meshio.read_glb(str(tmp_path / "g.glb"))
    assert bare["uvs"] is None and bare["basecolor_tex"] is None and np.array_equal(bare["faces"], f)
    with pytest.raises(ValueError):
        meshio.write_glb(str(tmp_path / "bad.glb"), v, f + len(v))


def test_ply_round_trip(tmp_path):
    from sculptmate_amd import meshio

    v, f, c, _ = _random_mesh(seed=3)
    meshio.write_ply(str(tmp_path / "m.ply"), v, f, c)
    v2, f2, c2 = meshio.read_ply(str(tmp_path / "m.ply"))
    assert np.array_equal(v2, v) and np.array_equal(f2, f) and np.abs(c2 - c).max() <= 0.5 / 255 + 1e-7
    meshio.write_ply(str(tmp_path / "n.ply"), v, f)
    v3, f3, c3 = meshio.read_ply(str(tmp_path / "n.ply"))
    assert np.array_equal(v3, v) and np.array_equal(f3, f) and c3 is None


def test_bench_refuses_rank_count_mismatch():
    """bench.py --gpus N is launched by torch.distributed.run with N ranks; a lone process must not report an N-GPU number."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "needs 2 ranks" in (p.stderr + p.stdout)
    assert p.stdout.strip() == ""            # no JSON line


def test_limb_tiled_layout_helpers_follow_the_documented_offsets():
    """The limb-tiled layout of include/sculpt_hip.h / csrc/limbs.h, host side: ops.limbs_join inverts an array assembled element by
    element from the documented byte offset; ops.geglu_row_blocks orders a GEGLU weight the way gemm_l3p's tiles read it
    (csrc/gemm_f32.h f32_tile_wrow: tile row j -> value / gate row)."""
    import numpy as np
    import torch

    from sculptmate_amd import ops

    rng = np.random.default_rng(0)
    R, K = 45, 64
    x = (rng.standard_normal((R, K)) * np.exp2(rng.integers(-20, 20, (R, K)))).astype(np.float32)
    xt = torch.from_numpy(x)
    l1 = xt.to(torch.bfloat16)
    r1 = xt - l1.float()
    l2 = r1.to(torch.bfloat16)
    l3 = (r1 - l2.float()).to(torch.bfloat16)
    assert torch.equal((l1.float() + l2.float()) + l3.float(), xt)          # three bf16 limbs carry an fp32 value exactly
    nbytes = ops.limbs_bytes(R, K)
    assert nbytes == 2 * K * 192
    flat = torch.zeros(nbytes // 2, dtype=torch.bfloat16)
    for l, limb in enumerate((l1, l2, l3)):
        for r in range(R):
            for k in range(K):
                flat[((((r // 32) * (K // 8) + k // 8) * 3 + l) * 512 + (r % 32) * 16 + (k % 8) * 2) // 2] = limb[r, k]
    assert torch.equal(ops.limbs_join(flat.view(torch.uint8), R, K), xt)
    # GEGLU: 2N rows (value rows, then gate rows) -> per 64 output columns the four 32-row blocks value, gate, value, gate
    N = 128
    W = torch.arange(2 * N, dtype=torch.float32)[:, None].repeat(1, 4)
    P = ops.geglu_row_blocks(W)
    for t in range(N // 64):
        for j in range(128):
            sub, within = j >> 5, j & 31
            assert int(P[t * 128 + j, 0]) == ((N if (sub & 1) else 0) + t * 64 + (sub >> 1) * 32 + within)
