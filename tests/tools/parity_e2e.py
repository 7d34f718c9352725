"""End-to-end image -> mesh parity of the fp32 mode against the CPU path (oracle), full-size model, and timing."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import capi, tsr_ref
from sculptmate_amd import ops, synth
from sculptmate_amd.tsr import TSR
from sculptmate_amd.tsr.spec import DEFAULT_CFG
dev = torch.device("cuda:0")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 128
sd = synth.tsr_state(0)
img = synth.composite_rgb(synth.image_rgba(seed=100))
torch.set_num_threads(min(32, os.cpu_count()))
t0 = time.time(); ref = tsr_ref.tsr_forward(sd, img, DEFAULT_CFG, pos_mode="scale_factor"); print("oracle forward %.1fs" % (time.time() - t0))
Ws, bs = synth.decoder_lists(sd)
for prec in ("fp32", "bf16"):
    m = TSR(pos_embed_mode="scale_factor", precision=prec); m.load_state_dict(sd); m.to(dev)
    codes = m([img], device=dev); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): codes = m([img], device=dev)
    torch.cuda.synchronize(); tf = (time.perf_counter() - t0) / 3 * 1e3
    rel = float((codes[0].cpu() - ref).norm() / ref.norm())
    print("%s: forward %.1f ms, scene code rel err vs fp32 oracle %.3e" % (prec, tf, rel))
    if prec == "fp32":
        dens = ops.density_grid(codes[0].contiguous(), m.decoder, R)
        t0 = time.time(); dref = capi.density_grid(ref.numpy(), Ws, bs, R); print("oracle density %d^3 %.1fs" % (R, time.time() - t0))
        thr = float(np.quantile(dref, 0.97))
        print("density (log) max diff %.3e" % float(np.abs(np.log(dens.cpu().numpy()) - np.log(dref)).max()))
        mesh = m.extract_meshes(codes, resolution=R, threshold=thr)[0]
        rv, rf = capi.reference_isosurface(-(dref - np.float32(thr)), R)
        rv = rv * np.float32(1.74) + np.float32(-0.87)
        v, f = mesh.vertices.cpu().numpy(), mesh.faces.cpu().numpy()
        print("mesh: gpu %d verts %d faces | cpu %d verts %d faces" % (len(v), len(f), len(rv), len(rf)))
        if f.shape == rf.shape and np.array_equal(f, rf):
            print("identical topology; max vertex diff %.3e (rel to extent 1.74: %.3e)" % (np.abs(v - rv).max(), np.abs(v - rv).max() / 1.74))
        else:
            # compare as point sets: nearest-neighbour distance of a sample
            from scipy.spatial import cKDTree
            d, _ = cKDTree(rv).query(v[:: max(1, len(v) // 20000)])
            print("topology differs in places; sampled nearest-vertex distance: median %.3e max %.3e" % (np.median(d), d.max()))
    del m; torch.cuda.empty_cache()
