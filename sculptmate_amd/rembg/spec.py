"""U^2-Net (the network inside the reference's `checkpoints/u2net.onnx`, run by rembg/sessions/u2net.py:16-46 through
onnxruntime) -- layer inventory.  The ONNX file is absent from the reference checkout and no network definition is in
the repository; this is the published architecture (Qin et al., "U^2-Net: Going Deeper with Nested U-Structure for
Salient Object Detection", Pattern Recognition 2020, and the authors' model/u2net.py): REBNCONV = 3x3 conv (dilation
d, padding d) + BatchNorm + ReLU; RSU-L blocks; six encoder and five decoder stages; six 3x3 side convolutions to one
channel and a 1x1 fusion convolution; sigmoid.  Parameter names follow the authors' PyTorch module tree.
PARITY UNPINNED against the reference's ONNX graph (opaque binary, not present)."""

# stage name -> (kind, in_ch, mid_ch, out_ch)
STAGES = (
    ("stage1", "RSU7", 3, 32, 64), ("stage2", "RSU6", 64, 32, 128), ("stage3", "RSU5", 128, 64, 256),
    ("stage4", "RSU4", 256, 128, 512), ("stage5", "RSU4F", 512, 256, 512), ("stage6", "RSU4F", 512, 256, 512),
    ("stage5d", "RSU4F", 1024, 256, 512), ("stage4d", "RSU4", 1024, 128, 256), ("stage3d", "RSU5", 512, 64, 128),
    ("stage2d", "RSU6", 256, 32, 64), ("stage1d", "RSU7", 128, 16, 64),
)
SIDES = (("side1", 64), ("side2", 64), ("side3", 128), ("side4", 256), ("side5", 512), ("side6", 512))
DEPTH = {"RSU7": 7, "RSU6": 6, "RSU5": 5, "RSU4": 4}
BN_EPS = 1e-5


def rsu_layers(kind, cin, mid, cout):
    """[(layer name, in channels, out channels, dilation)] of one RSU block, in module order."""
    if kind == "RSU4F":
        return [("rebnconvin", cin, cout, 1), ("rebnconv1", cout, mid, 1), ("rebnconv2", mid, mid, 2),
                ("rebnconv3", mid, mid, 4), ("rebnconv4", mid, mid, 8), ("rebnconv3d", 2 * mid, mid, 4),
                ("rebnconv2d", 2 * mid, mid, 2), ("rebnconv1d", 2 * mid, cout, 1)]
    L = DEPTH[kind]
    layers = [("rebnconvin", cin, cout, 1), ("rebnconv1", cout, mid, 1)]
    for i in range(2, L):
        layers.append(("rebnconv%d" % i, mid, mid, 1))
    layers.append(("rebnconv%d" % L, mid, mid, 2))
    for i in range(L - 1, 1, -1):
        layers.append(("rebnconv%dd" % i, 2 * mid, mid, 1))
    layers.append(("rebnconv1d", 2 * mid, cout, 1))
    return layers


def param_spec():
    spec = {}
    for name, kind, cin, mid, cout in STAGES:
        for lname, ci, co, _d in rsu_layers(kind, cin, mid, cout):
            p = "%s.%s." % (name, lname)
            spec[p + "conv_s1.weight"] = (co, ci, 3, 3)
            spec[p + "conv_s1.bias"] = (co,)
            for k in ("weight", "bias", "running_mean", "running_var"):
                spec[p + "bn_s1." + k] = (co,)
    for name, c in SIDES:
        spec[name + ".weight"] = (1, c, 3, 3)
        spec[name + ".bias"] = (1,)
    spec["outconv.weight"] = (1, 6, 1, 1)
    spec["outconv.bias"] = (1,)
    return spec
