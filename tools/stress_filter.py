"""Stress of the filtered density grid through the model's own entry point: N synthetic images through ONE full-size model whose
margin is calibrated on the first scene code only; every mesh (256^3) against TSR(decoder_filter=False) on the same scene code,
bit for bit; guard statistics per image.      python tools/stress_filter.py [N] [--levels]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from sculptmate_amd import synth

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
model, sd = bench.build_model(dev, 0)
imgs = [torch.from_numpy(synth.composite_rgb(synth.image_rgba(seed=200 + i))).to(dev) for i in range(N)]
with torch.no_grad():
    bench.calibrate(model, sd, imgs[0])
    bad = 0
    levels = (25.0, 10.0, 60.0) if "--levels" in sys.argv else (25.0,)
    for i, im in enumerate(imgs):
        codes = model([im], device=dev)
        for thr in levels:
            model.decoder_filter = True
            try:
                a = model.extract_meshes(codes, False, 256, thr)[0]
            except Exception as e:
                a = e
            st = dict(model.filter_info["last"] or {})
            model.decoder_filter = False
            try:
                b = model.extract_meshes(codes, False, 256, thr)[0]
            except Exception as e:
                b = e
            if isinstance(a, Exception) or isinstance(b, Exception):
                same = type(a) is type(b)
                desc = "both raised %s" % type(a).__name__ if same else "DIFFERENT OUTCOME %r / %r" % (a, b)
            else:
                same = torch.equal(a.faces, b.faces) and torch.equal(a.vertices.view(torch.int32), b.vertices.view(torch.int32))
                desc = "%d verts %d faces" % (a.vertices.shape[0], a.faces.shape[0])
            bad += 0 if same else 1
            print("image %d level %5.1f: identical %s (%s); refined %.2f %%, marked %.2f %%, guard %.2f of the margin; fallbacks so far %d"
                  % (i, thr, same, desc, 100.0 * st.get("n_refined", 0) / max(st.get("n_points", 1), 1), 100.0 * st.get("n_marked", 0) / max(st.get("n_points", 1), 1),
                     st.get("max_err", 0.0) / (model.filter_info["margin"] or float("nan")), model.filter_info["fallbacks"]), flush=True)
    print("margin %.4f (%s operands), calibrations %d, filtered %d, fallbacks %d, meshes that differ: %d"
          % (model.filter_info["margin"], model.filter_info["coarse"], model.filter_info["calibrations"], model.filter_info["filtered"],
             model.filter_info["fallbacks"], bad))
    sys.exit(1 if bad else 0)
