"""TSR(precision="fp16l2") against TSR(precision="bf16l3") on a set of synthetic images: relative distance of the scene codes
(both are fp32-equivalent: their distance is fp32 rounding noise), range fallbacks, and the 256^3 meshes of a few of them.
    python tools/stress_fp16l2.py [n_images]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from sculptmate_amd import synth
from sculptmate_amd.tsr import TSR

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda:0")
base, sd = bench.build_model(dev, 0)
imgs = [torch.from_numpy(synth.composite_rgb(synth.image_rgba(seed=300 + i))).to(dev) for i in range(n)]
with torch.no_grad():
    bench.calibrate(base, sd, imgs[0])
    sdc = base.state_dict()
    models = {}
    for prec in ("fp16l2", "bf16l3"):
        m = TSR(pos_embed_mode="scale_factor", precision=prec); m.load_state_dict(sdc); m.to(dev); models[prec] = m
    rels = []
    for i, im in enumerate(imgs):
        a, b = models["fp16l2"].forward(im), models["bf16l3"].forward(im)
        rels.append(float((a - b).norm() / b.norm()))
        if i < 3:
            ma, mb = models["fp16l2"].extract_meshes(a, False, 256, 25.0)[0], models["bf16l3"].extract_meshes(b, False, 256, 25.0)[0]
            from scipy.spatial import cKDTree
            va, vb = ma.vertices.cpu().numpy(), mb.vertices.cpu().numpy()
            d = max(cKDTree(vb).query(va)[0].max(), cKDTree(va).query(vb)[0].max()) / 1.74
            print("image %d: 256^3 meshes %d / %d vertices, two-sided max vertex distance %.2e of the extent" % (i, len(va), len(vb), d), flush=True)
    print("scene code fp16l2 vs bf16l3 over %d images: rel median %.2e, max %.2e; range fallbacks %d; all finite %s"
          % (n, np.median(rels), max(rels), models["fp16l2"].range_fallbacks, True))
