"""U^2-Net (background removal network) on the MI355X vs the torch oracle of the published architecture."""
import numpy as np
import pytest
import torch

from oracle import u2net_ref as R
from sculptmate_amd import synth

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12)), float((a - b).abs().max())


def test_conv_blocks_vs_torch(cuda):
    import torch.nn.functional as F

    from sculptmate_amd import ops

    g = torch.Generator().manual_seed(0)
    H, W, ci, co, d = 13, 9, 24, 40, 2
    x = torch.randn(1, ci, H, W, generator=g).to(torch.bfloat16).float()
    w = (torch.randn(co, ci, 3, 3, generator=g) / 10).to(torch.bfloat16).float()
    b = torch.randn(co, generator=g)
    ref = F.relu(F.conv2d(x, w, b, padding=d, dilation=d))
    buf = torch.zeros(H * W, 64, dtype=torch.bfloat16, device=cuda)
    buf[:, 8:8 + ci] = x[0].permute(1, 2, 0).reshape(H * W, ci).to(torch.bfloat16).to(cuda)  # a slice at offset 8
    xin = ops.Act(buf, 8, ci, H, W)
    W2 = torch.zeros(128, 9, 64)
    W2[:co, :, :ci] = w.permute(0, 2, 3, 1).reshape(co, 9, ci)
    b2 = torch.zeros(128)
    b2[:co] = b
    obuf = torch.full((H * W, 128), 7.0, dtype=torch.bfloat16, device=cuda)
    out = ops.Act(obuf, 16, co, H, W)
    col = torch.empty(H * W * 9 * 64, dtype=torch.bfloat16, device=cuda)
    W2d = W2.reshape(128, -1).to(torch.bfloat16).to(cuda)
    ops.conv3x3_bf16(xin, W2d, b2.to(cuda), out, co, d, True, col, implicit=False)
    got = obuf[:, 16:16 + co].float().cpu().reshape(H, W, co).permute(2, 0, 1)
    assert _rel(got, ref[0])[0] < 4e-3
    assert (obuf[:, :16] == 7).all() and (obuf[:, 16 + co:] == 7).all()  # nothing outside the slice is written
    # implicit GEMM (no im2col rows): needs 64 readable channels from the slice start -> a wider buffer; same result
    wide = torch.zeros(H * W, 128, dtype=torch.bfloat16, device=cuda)
    wide[:, 8:8 + ci] = buf[:, 8:8 + ci]
    wide[:, 8 + ci:72] = 3.0  # finite junk in the channels whose weights are zero
    obuf2 = torch.full((H * W, 128), 7.0, dtype=torch.bfloat16, device=cuda)
    ops.conv3x3_bf16(ops.Act(wide, 8, ci, H, W), W2d, b2.to(cuda), ops.Act(obuf2, 16, co, H, W), co, d, True, implicit=True)
    assert torch.equal(obuf2, obuf)
    f32o = torch.empty(H * W, 128, device=cuda)
    ops.conv3x3_bf16(ops.Act(wide, 8, ci, H, W), W2d, b2.to(cuda), f32o, 0, d, False, implicit=True)
    pre = F.conv2d(x, w, b, padding=d, dilation=d)[0].permute(1, 2, 0).reshape(H * W, co)
    assert _rel(f32o[:, :co], pre)[0] < 1e-5
    # pool / upsample / add
    pb = torch.zeros(((H + 1) // 2) * ((W + 1) // 2), 64, dtype=torch.bfloat16, device=cuda)
    ops.maxpool2x2_ceil(xin, ops.Act(pb, 0, ci, (H + 1) // 2, (W + 1) // 2))
    rp = F.max_pool2d(x, 2, stride=2, ceil_mode=True)[0]
    assert torch.equal(pb[:, :ci].float().cpu().reshape((H + 1) // 2, (W + 1) // 2, ci).permute(2, 0, 1), rp)
    ub = torch.zeros(31 * 20, 64, dtype=torch.bfloat16, device=cuda)
    ops.upsample_bilinear(xin, ops.Act(ub, 0, ci, 31, 20))
    ru = F.interpolate(x, size=(31, 20), mode="bilinear", align_corners=False)[0]
    assert _rel(ub[:, :ci].float().cpu().reshape(31, 20, ci).permute(2, 0, 1), ru)[1] < 3e-2  # bf16 output rounding
    ab = torch.zeros(H * W, 64, dtype=torch.bfloat16, device=cuda)
    ops.add_bf16(xin, xin, ops.Act(ab, 0, ci, H, W))
    assert torch.equal(ab[:, :ci].float().cpu(), (2 * x[0]).permute(1, 2, 0).reshape(H * W, ci).to(torch.bfloat16).float())


@pytest.mark.parametrize("size", [(64, 64), (72, 56)])
def test_u2net_forward_vs_oracle(cuda, size):
    from sculptmate_amd.rembg.u2net import U2Net

    sd = synth.u2net_state(0)
    net = U2Net()
    net.load_state_dict(sd)
    net.to(cuda)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, *size, generator=g)
    d0 = net.forward(x.to(cuda))
    assert d0.shape == size and d0.dtype == torch.float32
    ref_bf = R.u2net_forward(sd, x[None], bf16=True)[0, 0]
    ref_32 = R.u2net_forward(sd, x[None])[0, 0]
    rb, r32 = _rel(d0, ref_bf), _rel(d0, ref_32)
    assert rb[1] < 2e-2 and r32[1] < 5e-2, (rb, r32)  # ~112 bf16 convolutions deep
    with pytest.raises(RuntimeError):
        bad = dict(sd)
        bad.pop("outconv.bias")
        U2Net().load_state_dict(bad)


def test_remove_end_to_end(cuda):
    """bg.remove() through the HIP session: same mask as the oracle network + the reference's pre/post-processing."""
    from PIL import Image

    from sculptmate_amd.rembg import bg, session

    sd = synth.u2net_state(0)
    s = session.U2netSession(device=cuda, state_dict=sd)
    rgba = synth.image_rgba(3, 160)
    img = Image.fromarray(rgba[..., :3], mode="RGB")
    mask = bg.remove(img, session=s, only_mask=True)
    assert mask.size == img.size and mask.mode == "L"
    x = session.normalize(img)
    ref = R.u2net_forward(sd, torch.from_numpy(x), bf16=True)[0].numpy()
    ref_mask = np.asarray(session.prediction_to_mask(ref, img.size)).astype(np.int32)
    d = np.abs(np.asarray(mask).astype(np.int32) - ref_mask)
    assert d.max() <= 12 and d.mean() < 1.5, (d.max(), d.mean())  # 8-bit levels after min-max stretching
    cut = bg.remove(np.asarray(img), session=s)
    assert cut.shape == (160, 160, 4)
    png = bg.remove(img, session=s, bgcolor=(255, 0, 0, 255))
    assert png.mode == "RGBA"
    with pytest.raises(NotImplementedError):
        bg.remove(img, session=s, alpha_matting=True)
    import time

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        d0 = s.net.forward(torch.from_numpy(x[0]).to(cuda))
    torch.cuda.synchronize()
    print("U2Net 320x320 forward: %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))


def test_session_from_onnx_container(cuda, tmp_path):
    """The reference opens checkpoints/u2net.onnx (rembg/sessions/base.py:38-42): a session built from an ONNX container
    (initialisers named like the exporter would, with BatchNormalization nodes) predicts bit-identically to one built
    from the state dict."""
    from PIL import Image

    from sculptmate_amd.rembg import onnx_weights as ow
    from sculptmate_amd.rembg import session

    sd = synth.u2net_state(0)
    path = str(tmp_path / "u2net.onnx")
    with open(path, "wb") as fh:
        fh.write(ow.encode_model([], [ow.encode_tensor(k, np.asarray(v, np.float32)) for k, v in sd.items()]))
    a = session.U2netSession(device=cuda, state_dict=sd)
    b = session.U2netSession(device=cuda, weights_path=path)
    img = Image.fromarray(synth.image_rgba(5, 96)[..., :3], mode="RGB")
    ma, mb = a.predict(img)[0], b.predict(img)[0]
    assert np.array_equal(np.asarray(ma), np.asarray(mb))


def test_preprocess_image_end_to_end(cuda, tmp_path):
    """preprocessing.preprocess_image (the add-on's caller, GUIPanel.py:158-160) with the HIP session."""
    from PIL import Image

    from sculptmate_amd import preprocessing
    from sculptmate_amd.rembg import session

    s = session.U2netSession(device=cuda, state_dict=synth.u2net_state(0))
    path = str(tmp_path / "in.png")
    Image.fromarray(synth.image_rgba(4, 400)[..., :3], mode="RGB").save(path)
    rgba = preprocessing.preprocess_image(path, ratio=0.85, use_alpha=True, session=s)
    assert rgba.mode == "RGBA" and rgba.size[0] == rgba.size[1]
    rgb = preprocessing.preprocess_image(path, ratio=0.75, session=s)
    assert rgb is None or (rgb.mode == "RGB" and rgb.size == (1024, 1024))
