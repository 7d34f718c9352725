"""Adversarial stress of the two-pass density grid's guard, through the model's own entry point (TSR.extract_meshes).

For every DECODER of a family of hostile decoders (heavy-tailed weights, rows / columns x30, large biases, besides the reference's
initialiser) the margin is calibrated ONCE, on the decoder's first scene code -- the production situation -- and then a series of
scene codes it was not calibrated on (smooth / white-noise / spiky triplanes scaled x0.1 ... x30) is meshed at levels from the 50 %
to the 99.9 % quantile of the density, at 256^3 (and 512^3 with --r512): every mesh against TSR(decoder_filter=False) on the same
scene code, bit for bit.  A field counts as
    filtered   the guard passed and the mesh is the full evaluation's            (must be every field the guard passes)
    fallback   the guard tripped: the grid was redone in full before a mesh was returned (identical by construction; checked).
               For each of these the unguarded two-pass volume (same margin) is meshed as well: "needed" counts the fallbacks
               without which a different mesh would have been returned, the rest are the guard's caution
    ESCAPED    the guard passed and the mesh differs                              (must be 0)
--safety S calibrates with S x the probe's largest error instead of 8 x (S = 1: the margin IS the largest error seen, so that
coarse errors beyond the margin do occur and the guard has something to catch).

    python tools/stress_filter.py [--decoders 14] [--scenes 5] [--levels 3] [--safety 8] [--r512] [--seed 0]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from sculptmate_amd import ops, synth
from sculptmate_amd.tsr import TSR
from sculptmate_amd.tsr.spec import SMALL_CFG

ap = argparse.ArgumentParser()
ap.add_argument("--decoders", type=int, default=14)
ap.add_argument("--scenes", type=int, default=5)
ap.add_argument("--levels", type=int, default=3)
ap.add_argument("--safety", type=float, default=8.0)
ap.add_argument("--r512", action="store_true")
ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
dev = torch.device("cuda:0")

KINDS = ("kaiming", "heavy_tail", "outlier_rows", "outlier_cols", "big_bias", "heavy_tail+rows", "sharp")


def decoder(kind, seed):
    """(Ws, bs) in torch layout: the reference's NeRFMLP shapes (network_utils.py:48-79) with hostile statistics."""
    rng = np.random.default_rng([seed, 501])
    Ws, bs = synth.decoder_lists(synth.decoder_state(seed=seed))
    Ws, bs = [w.copy() for w in Ws], [b.copy() for b in bs]
    nh = len(Ws)
    strength = (3.0, 10.0, 30.0)[(seed // len(KINDS)) % 3]      # graded: the mild ones keep the filter usable, the strong ones switch it off
    if "heavy_tail" in kind:      # Student t at the initialiser's scale: a few weights 5-50 x the rest
        df = {3.0: 6.0, 10.0: 3.5, 30.0: 2.5}[strength]
        for l in range(nh):
            t = rng.standard_t(df, Ws[l].shape).astype(np.float32)
            Ws[l] = (t * np.float32(np.sqrt(2.0 / Ws[l].shape[1]) / np.sqrt(df / (df - 2.0)))).astype(np.float32)
    if "rows" in kind:            # a few output neurons x3 ... x30 in two hidden layers (massive activations)
        for l in rng.choice(np.arange(1, nh - 1), 2, replace=False):
            for r in rng.choice(64, 3, replace=False):
                Ws[l][r] *= np.float32(strength)
                bs[l][r] *= np.float32(strength)
    if "cols" in kind:            # a few input channels x3 ... x30
        for l in rng.choice(np.arange(1, nh - 1), 2, replace=False):
            for c in rng.choice(64, 3, replace=False):
                Ws[l][:, c] *= np.float32(strength)
    if kind == "big_bias":
        for l in range(nh - 1):
            bs[l] = (bs[l] * np.float32(strength / 2.5)).astype(np.float32)
    if kind == "sharp":           # every hidden layer x2: steep fields, thin shells
        for l in range(1, nh - 1):
            Ws[l] = (Ws[l] * np.float32(2.0)).astype(np.float32)
    return Ws, bs


def scene(kind, seed, scale):
    if kind == "smooth":
        t = synth.smooth_triplane(seed=seed, scale=scale)
    elif kind == "noise":
        t = synth.triplane(seed=seed, scale=scale)
    else:                         # smooth + sparse spikes of 20 x the scale
        rng = np.random.default_rng([seed, 502])
        t = synth.smooth_triplane(seed=seed, scale=scale)
        m = rng.random(t.shape) < 0.002
        t = (t + m * rng.standard_normal(t.shape).astype(np.float32) * np.float32(20.0 * scale)).astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(t, np.float32)).to(dev)


def same_mesh(a, b):
    if isinstance(a, Exception) or isinstance(b, Exception):
        return type(a) is type(b)
    return (a.vertices.shape == b.vertices.shape and a.faces.shape == b.faces.shape and torch.equal(a.faces, b.faces)
            and torch.equal(a.vertices.view(torch.int32), b.vertices.view(torch.int32)))


def mesh_of(model, planes, R, thr):
    try:
        return model.extract_meshes(planes[None], False, R, thr)[0]
    except Exception as e:   # an empty surface / a NaN volume raises in both evaluations
        return e


SCALES = (0.1, 0.3, 1.0, 3.0, 10.0, 30.0)
QUANTS = (0.5, 0.8, 0.95, 0.99, 0.999)
sd = synth.tsr_state(3, SMALL_CFG)
filt = TSR(SMALL_CFG)
filt.load_state_dict(sd)
filt.to(dev)
full = TSR(SMALL_CFG, decoder_filter=False)
full.load_state_dict(sd)
full.to(dev)
filt.FILTER_SAFETY = args.safety
rng = np.random.default_rng([args.seed, 503])
tot = {"fields": 0, "filtered": 0, "fallback": 0, "escaped": 0, "unfiltered": 0, "raised": 0, "needed": 0}
per_kind = {}
worst_ratio = 0.0
with torch.no_grad():
    for di in range(args.decoders):
        kind = KINDS[di % len(KINDS)]
        Ws, bs = decoder(kind, 1000 * len(KINDS) * args.seed + di)
        mlp = ops.PackedMLP(Ws, bs, dev)
        filt.decoder = full.decoder = mlp
        filt.filter_info.update(margin=None, usable=True, filtered=0, fallbacks=0, calibrations=0, last=None)
        k = per_kind.setdefault(kind, {"fields": 0, "filtered": 0, "fallback": 0, "escaped": 0, "unfiltered": 0, "coarse": set()})
        for si in range(args.scenes):
            skind = ("smooth", "noise", "spiky")[(si + di) % 3]
            scale = SCALES[int(rng.integers(len(SCALES)))]
            planes = scene(skind, 7000 + 100 * di + si, scale)
            dens = ops.density_grid(planes, mlp, 64, precision="bf16l3").double().cpu().numpy()
            finite = dens[np.isfinite(dens)]
            qs = [QUANTS[int(q)] for q in rng.choice(len(QUANTS), args.levels, replace=False)]
            for q in qs:
                thr = float(np.quantile(finite, q)) if finite.size else 25.0
                if not (np.isfinite(thr) and 1e-30 < thr < 1e30):
                    continue
                for R in ((256, 512) if args.r512 and si == 0 and q == qs[0] else (256,)):
                    before = (filt.filter_info["filtered"], filt.filter_info["fallbacks"])
                    used = (filt.filter_info["margin"], filt.filter_info["coarse"])
                    a = mesh_of(filt, planes, R, thr)
                    after = (filt.filter_info["filtered"], filt.filter_info["fallbacks"])
                    b = mesh_of(full, planes, R, thr)
                    same = same_mesh(a, b)
                    st = filt.filter_info["last"] or {}
                    how = "filtered" if after[0] > before[0] else ("fallback" if after[1] > before[1] else "unfiltered")
                    if how == "filtered" and not same:
                        how = "escaped"
                    if how != "escaped" and not same:
                        how = "escaped"   # a fallback / unfiltered call that differs would be a bug of another kind: count it
                    if how == "fallback" and used[0] is not None and not isinstance(b, Exception):
                        # what the call would have returned without the guard
                        cfg = filt.renderer.cfg
                        vol, _ = ops.density_grid_filtered(planes, mlp, R, used[0], radius=cfg.radius, density_bias=cfg.density_bias,
                                                           out_add=-thr, coarse=used[1])
                        try:
                            v, f = ops.marching_cubes(vol.view(R, R, R), 0.0, reference_order=True, vert_div=R - 1.0,
                                                      vert_mul=2 * cfg.radius, vert_add=-cfg.radius)
                            differs = not (f.shape == b.faces.shape and torch.equal(f, b.faces)
                                           and torch.equal(v.view(torch.int32), b.vertices.view(torch.int32)))
                        except Exception:
                            differs = True
                        tot["needed"] += differs
                        how_note = " (unguarded mesh %s)" % ("DIFFERS" if differs else "identical")
                    else:
                        how_note = ""
                    tot["fields"] += 1
                    tot[how] += 1
                    tot["raised"] += isinstance(a, Exception)
                    k["fields"] += 1
                    k[how] += 1
                    k["coarse"].add(filt.filter_info["coarse"] if filt.filter_info["usable"] else "off")
                    margin = filt.filter_info["margin"]
                    ratio = ops.filter_guard_error(st) / margin if (st and margin and how == "filtered") else float("nan")
                    if ratio == ratio:
                        worst_ratio = max(worst_ratio, ratio)
                    desc = ("raised %s" % type(a).__name__) if isinstance(a, Exception) else "%d verts" % a.vertices.shape[0]
                    print("decoder %2d %-16s scene %d %-6s x%-4g level q%.3f R %d: %-10s %s; refined %.2f %%, audit %d pts err %.2e, "
                          "re-evaluated err %.2e, mismatches %d, margin %s (%s)"
                          % (di, kind, si, skind, scale, q, R, (how.upper() if how == "escaped" else how) + how_note, desc,
                             100.0 * st.get("n_refined", 0) / max(st.get("n_points", 1), 1), st.get("n_audit", 0),
                             st.get("audit_err", 0.0), st.get("max_err", 0.0), st.get("n_mismatch", 0),
                             "%.4f" % margin if margin else "-", filt.filter_info["coarse"] if filt.filter_info["usable"] else "off"),
                          flush=True)
print()
for kind, k in per_kind.items():
    print("%-16s fields %3d: filtered %3d, fallback %3d, unfiltered %3d, ESCAPED %d; coarse operands %s"
          % (kind, k["fields"], k["filtered"], k["fallback"], k["unfiltered"], k["escaped"], sorted(k["coarse"])))
print("safety %g: %d fields (%d raised in both evaluations): filtered %d, fallback (caught by the guard) %d of which %d would have "
      "returned a different mesh, filter off %d, ESCAPED %d; largest guard figure among the passing calls %.3f of the margin (limit %.3f)"
      % (args.safety, tot["fields"], tot["raised"], tot["filtered"], tot["fallback"], tot["needed"], tot["unfiltered"], tot["escaped"],
         worst_ratio, TSR.FILTER_GUARD))
sys.exit(1 if tot["escaped"] else 0)
