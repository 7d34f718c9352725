"""U^2-Net on the MI355X: the network the reference runs through onnxruntime for background removal
(rembg/sessions/u2net.py:16-46, `checkpoints/u2net.onnx`).  Architecture and parameter names: spec.py (published
network; the ONNX file is absent from the reference checkout -> parity against it is unpinned).

Every REBNCONV is im2col + one bf16 MFMA GEMM with BatchNorm folded into the weights and ReLU in the epilogue, writing
straight into its slice of the concatenation buffer that the decoder side will read (no torch.cat, no copies);
max-pool / bilinear upsample / residual add are small channel-last kernels.  Activations: bf16 [H*W][channels]."""
import numpy as np
import torch

from .. import _lib, ops
from .spec import BN_EPS, DEPTH, SIDES, STAGES, param_spec, rsu_layers

BF16 = torch.bfloat16


def _r(x, m):
    return ((x + m - 1) // m) * m


class U2Net:
    def __init__(self):
        self._spec = param_spec()
        self._sd = None
        self.device = None
        self._w = None
        self._buffers = {}

    def load_state_dict(self, sd, strict=True):
        sd = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in sd.items()
              if not k.endswith("num_batches_tracked")}
        missing = [k for k in self._spec if k not in sd]
        unexpected = [k for k in sd if k not in self._spec]
        if strict and (missing or unexpected):
            raise RuntimeError("Error(s) in loading state_dict for U2Net: missing %s unexpected %s" % (missing[:5], unexpected[:5]))
        for k, shp in self._spec.items():
            if tuple(sd[k].shape) != tuple(shp):
                raise RuntimeError("size mismatch for %s: %s vs %s" % (k, tuple(sd[k].shape), shp))
        self._sd = {k: sd[k].astype(np.float32) for k in self._spec}
        if self.device is not None:
            self._prepare(self.device)
        return self

    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise _lib.SculptError("U2Net runs on an MI355X only (device %s requested; there is no CPU fallback)" % device)
        self.device = device
        if self._sd is not None:
            self._prepare(device)
        return self

    # ------------------------------------------------------------------ weights
    @staticmethod
    def _pack(w, b, dev):
        """[co, ci, 3, 3] -> bf16 [round_up(co,128)][9 * round_up(ci,64)] with k = (ky*3+kx)*C_pad + c, bias fp32 padded."""
        co, ci = w.shape[:2]
        cp, npad = _r(ci, 64), _r(co, 128)
        W2 = np.zeros((npad, 9, cp), np.float32)
        W2[:co, :, :ci] = w.transpose(0, 2, 3, 1).reshape(co, 9, ci)
        b2 = np.zeros(npad, np.float32)
        b2[:co] = b
        return (torch.from_numpy(W2.reshape(npad, 9 * cp)).to(device=dev, dtype=BF16).contiguous(),
                torch.from_numpy(b2).to(dev), co)

    def _prepare(self, dev):
        sd, w = self._sd, {}
        for name, kind, cin, mid, cout in STAGES:
            for lname, ci, co, d in rsu_layers(kind, cin, mid, cout):
                p = "%s.%s." % (name, lname)
                s = sd[p + "bn_s1.weight"] / np.sqrt(sd[p + "bn_s1.running_var"] + BN_EPS)
                wf = sd[p + "conv_s1.weight"] * s[:, None, None, None]
                bf = (sd[p + "conv_s1.bias"] - sd[p + "bn_s1.running_mean"]) * s + sd[p + "bn_s1.bias"]
                w[p] = self._pack(wf, bf, dev) + (d,)
        for name, c in SIDES:
            w[name] = self._pack(sd[name + ".weight"], sd[name + ".bias"], dev)
        w["fuse_w"] = torch.from_numpy(sd["outconv.weight"].reshape(6).copy()).to(dev)
        w["fuse_b"] = float(sd["outconv.bias"][0])
        self._w = w
        self._buffers = {}

    # ------------------------------------------------------------------ buffers
    def _buf(self, key, H, W, C):
        k = (key, H, W, C)
        t = self._buffers.get(k)
        if t is None:
            # pad channels stay zero forever; 64 more so that a slice starting inside the row still has a whole
            # 64-channel K-step readable for the implicit convolution (weights of channels past the slice are zero)
            t = torch.zeros((H * W, _r(C, 64) + 64), dtype=BF16, device=self.device)
            self._buffers[k] = t
        return t

    def _conv(self, key, x, out, relu=True):
        W2, b, co, d = self._w[key]
        ops.conv3x3_bf16(x, W2, b, out, co, d, relu)

    # ------------------------------------------------------------------ blocks
    def _rsu(self, name, kind, cin, mid, cout, x, out):
        H, W = x.H, x.W
        A = ops.Act
        hxin = A(self._buf(name + ".in", H, W, cout), 0, cout, H, W)
        self._conv(name + ".rebnconvin.", x, hxin)
        tmp = A(self._buf(name + ".d1", H, W, cout), 0, cout, H, W)
        if kind == "RSU4F":
            cats = [None] + [self._buf(name + ".cat%d" % i, H, W, 2 * mid) for i in (1, 2, 3)]
            first = lambda i: A(cats[i], 0, mid, H, W)        # noqa: E731
            second = lambda i: A(cats[i], mid, mid, H, W)     # noqa: E731
            full = lambda i: A(cats[i], 0, 2 * mid, H, W)     # noqa: E731
            self._conv(name + ".rebnconv1.", hxin, second(1))
            self._conv(name + ".rebnconv2.", second(1), second(2))
            self._conv(name + ".rebnconv3.", second(2), second(3))
            self._conv(name + ".rebnconv4.", second(3), first(3))
            self._conv(name + ".rebnconv3d.", full(3), first(2))
            self._conv(name + ".rebnconv2d.", full(2), first(1))
            self._conv(name + ".rebnconv1d.", full(1), tmp)
            ops.add_bf16(tmp, hxin, out)
            return
        L = DEPTH[kind]
        sizes = [None, (H, W)]
        for _ in range(2, L):
            sizes.append(((sizes[-1][0] + 1) // 2, (sizes[-1][1] + 1) // 2))
        cats = [None] + [self._buf(name + ".cat%d" % i, sizes[i][0], sizes[i][1], 2 * mid) for i in range(1, L)]
        first = lambda i: A(cats[i], 0, mid, *sizes[i])        # noqa: E731
        second = lambda i: A(cats[i], mid, mid, *sizes[i])     # noqa: E731
        full = lambda i: A(cats[i], 0, 2 * mid, *sizes[i])     # noqa: E731
        self._conv(name + ".rebnconv1.", hxin, second(1))
        for i in range(2, L):
            pooled = A(self._buf(name + ".pool%d" % i, sizes[i][0], sizes[i][1], mid), 0, mid, *sizes[i])
            ops.maxpool2x2_ceil(second(i - 1), pooled)
            self._conv(name + ".rebnconv%d." % i, pooled, second(i))
        self._conv(name + ".rebnconv%d." % L, second(L - 1), first(L - 1))  # dilated bottom, same resolution
        for i in range(L - 1, 0, -1):
            dst = tmp if i == 1 else A(self._buf(name + ".d%d" % i, sizes[i][0], sizes[i][1], mid), 0, mid, *sizes[i])
            self._conv(name + ".rebnconv%dd." % i, full(i), dst)
            if i > 1:
                ops.upsample_bilinear(dst, first(i - 1))
        ops.add_bf16(tmp, hxin, out)

    def forward(self, x_chw: torch.Tensor) -> torch.Tensor:
        """x [3,H,W] fp32 normalised image on the device -> d0 [H,W] fp32 (sigmoid of the fused side outputs)."""
        if self._w is None:
            raise _lib.SculptError("U2Net: weights not loaded / model not on a device")
        A = ops.Act
        _, H, W = x_chw.shape
        x0b = self._buf("x0", H, W, 8)
        x0b[:, :3] = x_chw.permute(1, 2, 0).reshape(H * W, 3).to(BF16)
        x = A(x0b, 0, 8, H, W)
        sz = [None, (H, W)]
        for _ in range(5):
            sz.append(((sz[-1][0] + 1) // 2, (sz[-1][1] + 1) // 2))
        spec = {s[0]: s[1:] for s in STAGES}
        enc_c = [None, 64, 128, 256, 512, 512]
        cat = [None] + [self._buf("CAT%d" % i, sz[i][0], sz[i][1], 2 * enc_c[i]) for i in range(1, 6)]
        inp = x
        for i in range(1, 6):  # encoder stages 1..5 write into the second half of CAT_i
            dst = A(cat[i], enc_c[i], enc_c[i], *sz[i])
            self._rsu("stage%d" % i, *spec["stage%d" % i], inp, dst)
            nxt = A(self._buf("P%d" % (i + 1), sz[i + 1][0], sz[i + 1][1], enc_c[i]), 0, enc_c[i], *sz[i + 1])
            ops.maxpool2x2_ceil(dst, nxt)
            inp = nxt
        h6 = A(self._buf("H6", sz[6][0], sz[6][1], 512), 0, 512, *sz[6])
        self._rsu("stage6", *spec["stage6"], inp, h6)
        dec_out = {6: h6}
        prev = h6
        for i in range(5, 0, -1):  # decoder: up(prev) -> first half of CAT_i, stage_id on the whole CAT_i
            ops.upsample_bilinear(prev, A(cat[i], 0, enc_c[i], *sz[i]))
            kind, cin, mid, cout = spec["stage%dd" % i]
            o = A(self._buf("D%d" % i, sz[i][0], sz[i][1], cout), 0, cout, *sz[i])
            self._rsu("stage%dd" % i, kind, cin, mid, cout, A(cat[i], 0, 2 * enc_c[i], *sz[i]), o)
            dec_out[i] = o
            prev = o
        maps = torch.empty((6, H * W), dtype=torch.float32, device=self.device)
        for k in range(1, 7):
            a = dec_out[k]
            W2, b, _co = self._w["side%d" % k]
            sideo = self._buffers.get(("side", k, a.H, a.W))
            if sideo is None:
                sideo = torch.empty((a.H * a.W, W2.shape[0]), dtype=torch.float32, device=self.device)
                self._buffers[("side", k, a.H, a.W)] = sideo
            ops.conv3x3_bf16(a, W2, b, sideo, 0, 1, False)
            ops.upsample_bilinear_f32(sideo, sideo.stride(0), a.H, a.W, maps[k - 1], H, W)
        d0 = torch.empty(H * W, dtype=torch.float32, device=self.device)
        ops.fuse_sigmoid(maps, self._w["fuse_w"], self._w["fuse_b"], d0)
        return d0.view(H, W)
