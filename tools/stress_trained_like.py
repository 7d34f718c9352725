"""Both 16-bit shortcuts of the engine on a model with TRAINED-LIKE weight statistics (synth.tsr_state(outliers=factor): massive
activation channels in both residual streams, heavy-tailed decoder): the real model.ckpt is absent, and a plain initialiser has
none of the outliers the fp16 range of precision="fp16l2" and the margin of the two-pass density grid must survive.
  * TSR(precision="fp16l2") against TSR(precision="bf16l3") per image: distance of the scene codes, range fallbacks (a scene code
    that left the fp16 range is redone on the bf16l3 twin: then the two are bit-identical);
  * TSR.extract_meshes with the two-pass grid against decoder_filter=False at 256^3: identical meshes; which coarse operands
    the calibration chose (fp16 / bf16 / filter off), guard fallbacks.
    python tools/stress_trained_like.py [--factor 100] [--images 4] [--small]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sculptmate_amd import ops, synth
from sculptmate_amd.tsr import TSR
from sculptmate_amd.tsr.spec import DEFAULT_CFG, SMALL_CFG

ap = argparse.ArgumentParser()
ap.add_argument("--factor", type=float, nargs="+", default=[30.0, 100.0, 1000.0])
ap.add_argument("--images", type=int, default=3)
ap.add_argument("--small", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = SMALL_CFG if args.small else DEFAULT_CFG
R = 96 if args.small else 256
bad = 0
with torch.no_grad():
    for factor in args.factor:
        sd = synth.tsr_state(7, cfg, outliers=factor)
        models = {}
        for prec in ("fp16l2", "bf16l3"):
            m = TSR(cfg, pos_embed_mode="scale_factor", precision=prec)
            m.load_state_dict(sd)
            m.to(dev)
            models[prec] = m
        nof = TSR(cfg, pos_embed_mode="scale_factor", precision="bf16l3", decoder_filter=False)
        nof.load_state_dict(sd)
        nof.to(dev)
        for i in range(args.images):
            img = torch.from_numpy(synth.composite_rgb(synth.image_rgba(seed=400 + i, size=cfg["cond_image_size"]))).to(dev)
            before = models["fp16l2"].range_fallbacks
            a, b = models["fp16l2"].forward(img), models["bf16l3"].forward(img)
            fell = models["fp16l2"].range_fallbacks - before
            rel = float((a - b).norm() / b.norm())
            finite = bool(torch.isfinite(a).all()) and bool(torch.isfinite(b).all())
            absmax = float(b.abs().max())
            # the level: the 98.5 % quantile of the density on a 64^3 probe (a surface exists whatever the weights do)
            dens = ops.density_grid(b[0].contiguous(), nof.decoder, 64, precision="bf16l3").double().cpu().numpy()
            fin = dens[np.isfinite(dens)]
            thr = float(np.quantile(fin, 0.985)) if fin.size else 25.0
            line = "factor %6g image %d: scene code |max| %.3g, fp16l2 vs bf16l3 rel %.2e%s, finite %s" % (
                factor, i, absmax, rel, " (range fallback: bit-identical %s)" % bool(torch.equal(a, b)) if fell else "", finite)
            if not (np.isfinite(thr) and 1e-30 < thr < 1e30):
                print(line + "; density not usable for a level (%r)" % thr, flush=True)
                continue
            f0 = (models["bf16l3"].filter_info["filtered"], models["bf16l3"].filter_info["fallbacks"])
            try:
                ma = models["bf16l3"].extract_meshes(b, False, R, thr)[0]
            except Exception as e:
                ma = e
            try:
                mb = nof.extract_meshes(b, False, R, thr)[0]
            except Exception as e:
                mb = e
            info = models["bf16l3"].filter_info
            how = "filtered" if info["filtered"] > f0[0] else ("guard fallback" if info["fallbacks"] > f0[1] else "filter off")
            if isinstance(ma, Exception) or isinstance(mb, Exception):
                same = type(ma) is type(mb)
                desc = "raised %s / %s" % (type(ma).__name__, type(mb).__name__)
            else:
                same = ma.faces.shape == mb.faces.shape and torch.equal(ma.faces, mb.faces) and torch.equal(ma.vertices.view(torch.int32), mb.vertices.view(torch.int32))
                desc = "%d verts" % ma.vertices.shape[0]
            bad += 0 if same and finite else 1
            print(line + "; mesh at %d^3 %s, identical to the unfiltered model's %s; grid: %s, coarse operands %s, margin %s"
                  % (R, desc, same, how, info["coarse"] if info["usable"] else "off", "%.4f" % info["margin"] if info["margin"] else "-"), flush=True)
        print("factor %g: fp16l2 range fallbacks %d of %d forwards; two-pass grid: filtered %d, guard fallbacks %d, calibrations %d, usable %s (%s)"
              % (factor, models["fp16l2"].range_fallbacks, args.images, models["bf16l3"].filter_info["filtered"], models["bf16l3"].filter_info["fallbacks"],
                 models["bf16l3"].filter_info["calibrations"], models["bf16l3"].filter_info["usable"], models["bf16l3"].filter_info["coarse"]), flush=True)
        del models, nof
        torch.cuda.empty_cache()
print("meshes that differ or non-finite scene codes: %d" % bad)
sys.exit(1 if bad else 0)
