"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by sculptmate_amd/).

Plain torch-fp32 (CPU) restatement of the StableFast geometry tail:
  dilate_fill                    /root/reference/StableFast/sf3d/models/utils.py:96-133
  Mesh._compute_vertex_normal    /root/reference/StableFast/sf3d/models/mesh.py:66-92
  Mesh._compute_vertex_tangent   /root/reference/StableFast/sf3d/models/mesh.py:94-139
written as explicit neighbourhood sums (what unfold / fold / max_pool / conv2d compute there).
PARITY PIN: tests/golden/sf3d_tail.npz, produced by importing the reference modules
(tests/golden/make_reference_goldens.py sf3d_tail).
"""
import torch
import torch.nn.functional as F


def _box3(x):
    """3x3 zero-padded neighbourhood sum of [C,H,W]."""
    p = F.pad(x, (1, 1, 1, 1))
    H, W = x.shape[-2:]
    return sum(p[..., dy:dy + H, dx:dx + W] for dy in range(3) for dx in range(3))


def dilate_fill(img, mask, iterations=10):
    img = torch.as_tensor(img, dtype=torch.float32)[0].clone()      # [3,H,W]
    m = torch.as_tensor(mask).to(torch.float32)[0]                  # [1,H,W]
    H, W = img.shape[-2:]
    interior = torch.zeros(1, H, W)
    interior[:, 1:-1, 1:-1] = 1.0
    for _ in range(iterations):
        new_m = F.max_pool2d(m[None], 3, 1, 1)[0]
        mean = _box3(img) / _box3(m).clamp(min=1) * interior        # patches exist only at interior centres
        new_img = new_m * _box3(mean) / _box3(new_m).clamp(min=1)
        diff = new_m - m
        img = img + diff * (new_img - img)
        m = new_m
    return img[None]


def vertex_normals(v_pos, faces):
    v = torch.as_tensor(v_pos, dtype=torch.float32)
    f = torch.as_tensor(faces).long()
    fn = torch.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]], dim=-1)
    n = torch.zeros_like(v)
    for k in range(3):
        n.index_add_(0, f[:, k], fn)
    bad = (n * n).sum(-1, keepdim=True) <= 1e-20
    n = torch.where(bad, torch.tensor([0.0, 0.0, 1.0]), n)
    return F.normalize(n, dim=1)


def vertex_tangents(v_pos, v_tex, v_nrm, faces):
    v = torch.as_tensor(v_pos, dtype=torch.float32)
    t = torch.as_tensor(v_tex, dtype=torch.float32)
    n = torch.as_tensor(v_nrm, dtype=torch.float32)
    f = torch.as_tensor(faces).long()
    duv1, duv2 = t[f[:, 1]] - t[f[:, 0]], t[f[:, 2]] - t[f[:, 0]]
    dp1, dp2 = v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]]
    nom = dp1 * duv2[:, 1:2] - dp2 * duv1[:, 1:2]
    den = (duv1[:, 0:1] * duv2[:, 1:2] - duv1[:, 1:2] * duv2[:, 0:1]).clamp(min=1e-6)
    tang = nom / den
    acc, cnt = torch.zeros_like(v), torch.zeros_like(v)
    for k in range(3):
        acc.index_add_(0, f[:, k], tang)
        cnt.index_add_(0, f[:, k], torch.ones_like(tang))
    tg = F.normalize(acc / cnt, dim=1)
    return F.normalize(tg - (tg * n).sum(-1, keepdim=True) * n)
