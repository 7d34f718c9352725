"""Stage timing of the StableFast-3D path (BASELINE config 4) at full size on one MI355X (HIP events)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sculptmate_amd import synth  # noqa: E402
from sculptmate_amd.sf3d.spec import DEFAULT_CFG  # noqa: E402
from sculptmate_amd.sf3d.system import SF3D  # noqa: E402

dev = torch.device("cuda:0")
_pos = [a for a in sys.argv[1:] if not a.startswith("--")]
res = int(_pos[0]) if _pos else DEFAULT_CFG["isosurface_resolution"]
cfg = dict(DEFAULT_CFG, isosurface_resolution=res)
t0 = time.time()
sd = synth.sf3d_state(0, cfg)
sd.update(synth.sf3d_estimator_state(0))   # CLIP image estimator + illumination estimator, like the shipped checkpoint
print("synthetic weights: %.1f s, %.1f M params" % (time.time() - t0, sum(v.size for v in sd.values()) / 1e6))
m = SF3D(cfg)
m.load_state_dict(sd)
t0 = time.time()
m.to(dev)
torch.cuda.synchronize()
g = m.isosurface_helper.grid
print("prepare: %.1f s; tet grid %s: %d vertices, %d tets, %d edges" % (time.time() - t0, m.isosurface_helper.source,
                                                                    g.n_vertices, g.n_tets, g.n_edges))
img = torch.from_numpy(synth.composite_rgb(synth.image_rgba(0, 512))).to(dev)
codes = m.scene_code(img)
torch.cuda.synchronize()
# calibrate the density head so ~10 % of the grid is inside
q = m.query_triplane(m._grid_world, codes)
pre = m.decoder(q, include=["density"])["density"].reshape(-1).log().cpu().numpy()
shift = np.log(cfg["isosurface_threshold"]) - np.quantile(pre.astype(np.float64), 0.9)
sd["decoder.heads.density.4.bias"] = (sd["decoder.heads.density.4.bias"] + np.float32(shift)).astype(np.float32)
m.load_state_dict(sd)


def ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def stage_times(n=5):
    rows = []
    for _ in range(n):
        e = [ev()]
        tok = m.image_tokens(img); e.append(ev())
        direct = m.backbone_tokens(tok); e.append(ev())
        planes = m.post_process(direct); e.append(ev())
        mesh = m.triplane_to_meshes(planes[None])[0]; e.append(ev())
        torch.cuda.synchronize()
        rows.append([e[i].elapsed_time(e[i + 1]) for i in range(4)])
    return np.median(np.array(rows), 0), mesh


def estimator_times(n=7):
    mask = torch.ones(1, 512, 512, device=dev)
    toks = [torch.randn(3 * 96 * 96, 1024, device=dev)]
    rows = []
    for _ in range(n):
        e = [ev()]
        m.image_estimator(img[None], mask=mask); e.append(ev())
        m.global_estimator(toks, 96); e.append(ev())
        torch.cuda.synchronize()
        rows.append([e[i].elapsed_time(e[i + 1]) for i in range(2)])
    return np.median(np.array(rows), 0)


stage_times(2)
t, mesh = stage_times(7)
te = estimator_times()
print("image estimator (CLIP ViT-B/32 + heads, every image) %.2f ms | illumination estimator (only on request) %.2f ms" % (te[0], te[1]))
print("dinov2 %.2f ms | backbone %.2f ms | upsampler %.2f ms | query+mtet %.2f ms | total %.2f ms -> %.1f meshes/s"
      % (t[0], t[1], t[2], t[3], t.sum(), 1e3 / t.sum()))
print("mesh: %d vertices, %d faces" % (mesh.v_pos.shape[0], mesh.t_pos_idx.shape[0]))


def unwrap_time(n=3):
    from sculptmate_amd.sf3d.system import Mesh

    ts = []
    for _ in range(n):
        mm = Mesh(mesh.v_pos.clone(), mesh.t_pos_idx.clone(), unwrapper=m.unwrapper)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mm.unwrap_uv()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    a = m.unwrapper.last["assigned"]
    return min(ts), float((a < 6).float().mean()), float(((a >= 6) & (a < 12)).float().mean()), float((a >= 12).float().mean())


tu, f0, f1, f2 = unwrap_time()
print("box-projection unwrap (incl. normals / tangents of the unrolled mesh): %.2f ms; front layer %.1f %%, overlap slice %.1f %%, "
      "remaining %.1f %%" % (tu, 100 * f0, 100 * f1, 100 * f2))
fl = dict(dino=0.95e12, backbone=8.5e12, post=1.9e12)
print("approx TFLOP/s: dino %.0f backbone %.0f upsampler %.0f" % (fl["dino"] / t[0] / 1e9, fl["backbone"] / t[1] / 1e9, fl["post"] / t[2] / 1e9))


def bake_time(res=512, n=3):
    """Texture bake on the unwrapped bench mesh: rasterize -> interpolate -> heads at the texels -> bake_material -> dilate_fill."""
    from sculptmate_amd.sf3d.bake import bake_textures
    from sculptmate_amd.sf3d.system import Mesh

    mm = Mesh(mesh.v_pos.clone(), mesh.t_pos_idx.clone(), unwrapper=m.unwrapper)
    mm.unwrap_uv()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tex = bake_textures(m, mm, codes, res, {}, 0)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts), tex


try:
    tb, tex = bake_time()
    print("texture bake at 512^2 (rasterize + interpolate + 3 heads at the texels + material + dilate, incl. the PIL hand-off): %.2f ms" % tb)
except Exception as e:  # noqa: BLE001  (a timing tool: report, do not hide the stages above)
    print("texture bake: failed (%s: %s)" % (type(e).__name__, e))


def remesh_time():
    """Mesh.triangle_remesh on the host (native decimate + Botsch-Kobbelt, sf3d/remesh.py) on the full raw mesh, 'high' setting
    (vertex_count = 0.75 x vertices, like generate_mesh)."""
    t0 = time.perf_counter()
    out = m.remesher(mesh, "triangle", round(0.75 * mesh.v_pos.shape[0]))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, out


if "--remesh" in sys.argv:
    tr, rm = remesh_time()
    print("triangle remesh on the host (%d -> %d vertices): %.1f ms" % (mesh.v_pos.shape[0], rm.v_pos.shape[0], tr))
