"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by sculptmate_amd/).

Plain PyTorch fp32 (CPU) statement of U^2-Net, the salient-object network the reference runs as an opaque ONNX graph
(`checkpoints/u2net.onnx`, rembg/sessions/u2net.py:16-46).  **PARITY UNPINNED**: the ONNX file is not in the reference
checkout and the repository holds no network definition, so this follows the published architecture (Qin et al. 2020;
the authors' model/u2net.py module tree, see sculptmate_amd/rembg/spec.py).  What IS pinned to the reference is the
pre/post-processing around the network (tests/golden/rembg_prepost.npz from rembg/sessions/base.py:44-69 and
u2net.py:34-46).

`bf16=True` rounds weights and stored activations to bfloat16 where the HIP pipeline does.
"""
import torch
import torch.nn.functional as F

from .tsr_ref import _Q, _t

BN_EPS = 1e-5


def fold_bn(sd, prefix):
    """conv + eval-mode BatchNorm -> (weight, bias) of the equivalent convolution."""
    w = _t(sd[prefix + "conv_s1.weight"]).float()
    b = _t(sd[prefix + "conv_s1.bias"]).float()
    g, be = _t(sd[prefix + "bn_s1.weight"]).float(), _t(sd[prefix + "bn_s1.bias"]).float()
    mu, var = _t(sd[prefix + "bn_s1.running_mean"]).float(), _t(sd[prefix + "bn_s1.running_var"]).float()
    s = g / torch.sqrt(var + BN_EPS)
    return w * s[:, None, None, None], (b - mu) * s + be


def _rebnconv(sd, prefix, x, d, Q):
    w, b = fold_bn(sd, prefix)
    return Q(F.relu(F.conv2d(Q(x), Q(w), b, padding=d, dilation=d)))


def _up(src, tar):
    return F.interpolate(src, size=tar.shape[2:], mode="bilinear", align_corners=False)


def _pool(x):
    return F.max_pool2d(x, 2, stride=2, ceil_mode=True)


def rsu(sd, name, kind, x, Q):
    c = lambda l, inp, d=1: _rebnconv(sd, "%s.%s." % (name, l), inp, d, Q)  # noqa: E731
    hxin = c("rebnconvin", x)
    if kind == "RSU4F":
        h1 = c("rebnconv1", hxin)
        h2 = c("rebnconv2", h1, 2)
        h3 = c("rebnconv3", h2, 4)
        h4 = c("rebnconv4", h3, 8)
        h3d = c("rebnconv3d", torch.cat((h4, h3), 1), 4)
        h2d = c("rebnconv2d", torch.cat((h3d, h2), 1), 2)
        h1d = c("rebnconv1d", torch.cat((h2d, h1), 1))
        return Q(h1d + hxin)
    L = {"RSU7": 7, "RSU6": 6, "RSU5": 5, "RSU4": 4}[kind]
    hs = [c("rebnconv1", hxin)]
    for i in range(2, L):
        hs.append(c("rebnconv%d" % i, _pool(hs[-1])))
    bottom = c("rebnconv%d" % L, hs[-1], 2)
    d = c("rebnconv%dd" % (L - 1), torch.cat((bottom, hs[-1]), 1))
    for i in range(L - 2, 0, -1):
        d = c("rebnconv%dd" % i, torch.cat((Q(_up(d, hs[i - 1])), hs[i - 1]), 1))
    return Q(d + hxin)


def u2net_forward(sd, x, bf16=False):
    """x [1,3,H,W] normalised image -> d0 (sigmoid of the fused side outputs) [1,1,H,W]."""
    Q = _Q(bf16)
    with torch.no_grad():
        x = _t(x).float()
        h1 = rsu(sd, "stage1", "RSU7", x, Q)
        h2 = rsu(sd, "stage2", "RSU6", _pool(h1), Q)
        h3 = rsu(sd, "stage3", "RSU5", _pool(h2), Q)
        h4 = rsu(sd, "stage4", "RSU4", _pool(h3), Q)
        h5 = rsu(sd, "stage5", "RSU4F", _pool(h4), Q)
        h6 = rsu(sd, "stage6", "RSU4F", _pool(h5), Q)
        h5d = rsu(sd, "stage5d", "RSU4F", torch.cat((Q(_up(h6, h5)), h5), 1), Q)
        h4d = rsu(sd, "stage4d", "RSU4", torch.cat((Q(_up(h5d, h4)), h4), 1), Q)
        h3d = rsu(sd, "stage3d", "RSU5", torch.cat((Q(_up(h4d, h3)), h3), 1), Q)
        h2d = rsu(sd, "stage2d", "RSU6", torch.cat((Q(_up(h3d, h2)), h2), 1), Q)
        h1d = rsu(sd, "stage1d", "RSU7", torch.cat((Q(_up(h2d, h1)), h1), 1), Q)
        side = lambda n, t: F.conv2d(Q(t), Q(_t(sd[n + ".weight"]).float()), _t(sd[n + ".bias"]).float(), padding=1)  # noqa: E731
        d1 = side("side1", h1d)
        ds = [d1] + [_up(side("side%d" % (i + 2), t), d1) for i, t in enumerate((h2d, h3d, h4d, h5d, h6))]
        d0 = F.conv2d(torch.cat(ds, 1), _t(sd["outconv.weight"]).float(), _t(sd["outconv.bias"]).float())
        return torch.sigmoid(d0)
