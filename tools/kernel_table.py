"""Per-kernel table of a `rocprofv3 --kernel-trace --stats` run of bench.py: calls per image, average duration, ms per image,
and for the dominant kernels the fraction of the roofline that binds them (SURVEY.md 8d figures).
    python tools/kernel_table.py kernel_stats.csv <bench steps incl. warm-up and stage-split passes>"""
import csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
# round 5: the default dense grid is the two-pass ("filtered") pipeline -- one density_coarse_kernel launch per image (plus the
# small probe launches of the calibration), filter_cells / filter_points, one density_list_l3k_kernel
coarse = [r for r in rows if "density_coarse_kernel" in r["Name"]]
FILTERED = bool(coarse)
dens = [r for r in rows if "density_grid_" in r["Name"]]
dens.sort(key=lambda r: -int(r["Calls"]))
mc_rows = [r for r in rows if "mc_classify" in r["Name"]]
n_img = int(mc_rows[0]["Calls"]) if mc_rows else (int(dens[0]["Calls"]) if dens else int(sys.argv[2]))   # one classification per mesh
L3 = FILTERED or (bool(dens) and "l3" in dens[0]["Name"])
PEAK_BF16, PEAK_F32 = 2500.0, 157.3


def flops(name):
    # algorithmic TFLOP per image of the kernel families: dense grid 1.382; transformer GEMMs 2.10; attention 0.86
    if "density_grid_kernel" in name: return 1.382, PEAK_F32
    return None, None


# the transformer runs once more than the dense grid (bench.py's calibration pass): images for its kernels = FF1 launches / 16
# (FF1 + GEGLU = EPI 2: on the 128-row tiles until round 3, on the 192 x 256 tiles of gemm256_kernel since round 4)
ff1 = [r for r in rows if "gemm_bf16_kernel<2," in r["Name"] or "gemm256_kernel<2," in r["Name"]]
n_tr = sum(int(r["Calls"]) for r in ff1) // 16 if ff1 else n_img


def images(name):
    return n_tr if ("gemm" in name or "attention" in name or "norm" in name or "patchify" in name or "vit_" in name
                    or "row_slice" in name or "upsample" in name) else n_img


tot = sum(float(r["TotalDurationNs"]) / images(r["Name"]) for r in rows) / 1e6
print("| kernel | launches / image | avg us | ms / image | share |")
print("|---|---|---|---|---|")
fam = {"gemm": 0.0, "attention": 0.0, "mc_": 0.0, "density": 0.0, "filter_": 0.0}
for r in rows[:26]:
    name = r["Name"].split("(")[0].replace("void ", "").replace("sculpt::", "")
    ms = float(r["TotalDurationNs"]) / images(r["Name"]) / 1e6
    for k in fam:
        if k in name: fam[k] += ms
    print("| `%s` | %.1f | %.1f | %.3f | %.1f %% |" % (name[:60], int(r["Calls"]) / images(r["Name"]), float(r["AverageNs"]) / 1e3, ms, 100 * ms / tot))
print()
print("images: %d (dense grid, marching cubes) / %d (transformer); all kernels: %.3f ms / image" % (n_img, n_tr, tot))
print("bf16 GEMMs %.3f ms (2.10 TFLOP -> %.0f TFLOP/s, %.2f of the 2.5 PFLOP/s peak); attention %.3f ms (0.86 TFLOP -> %.0f TFLOP/s, %.2f);"
      % (fam["gemm"], 2.10 / fam["gemm"] * 1e3, 2.10 / fam["gemm"] * 1e3 / PEAK_BF16, fam["attention"], 0.86 / fam["attention"] * 1e3,
         0.86 / fam["attention"] * 1e3 / PEAK_BF16))
if FILTERED:
    ca = sum(float(r["TotalDurationNs"]) for r in coarse) / n_img / 1e6
    cl = sum(float(r["TotalDurationNs"]) for r in rows if "density_list_l3k" in r["Name"]) / n_img / 1e6
    fu = sum(float(r["TotalDurationNs"]) for r in rows if "density_grid_l3k" in r["Name"]) / n_img / 1e6
    print("dense grid, two passes: density_coarse_kernel %.3f ms / image (1.0995 TFLOP executed per 256^3 launch: one fp16 product per layer; the "
          "kernel is bound by the issue cost of its fp32 SiLU, not by the matrix pipe) + filter_cells / filter_points %.3f ms + "
          "density_list_l3k_kernel %.3f ms (six bf16-limb products at the re-evaluated points); the full six-product kernel appears only in "
          "bench.py's identity check / sibling measurement (%.3f ms / image of this trace); marching cubes %.3f ms"
          % (ca, fam["filter_"], cl, fu, fam["mc_"]))
elif L3:  # executed MFMA work of the three-limb kernel: 8 layers x 48 MFMAs x 32 768 FLOP per 32 points = 6.597 TFLOP per 256^3 launch
    print("dense grid %.3f ms (6.597 TFLOP executed on the bf16 matrix pipe -> %.0f TFLOP/s, %.3f of the 2.5 PFLOP/s peak; 1.382 TFLOP "
          "algorithmic -> %.0f TFLOP/s); marching cubes %.3f ms"
          % (fam["density"], 6.597 / fam["density"] * 1e3, 6.597 / fam["density"] * 1e3 / PEAK_BF16, 1.382 / fam["density"] * 1e3, fam["mc_"]))
else:
    print("dense grid %.3f ms (1.0995 TFLOP executed on the fp32 matrix pipe -> %.1f TFLOP/s, %.3f of the 157.3 TFLOP/s peak); marching cubes %.3f ms"
          % (fam["density"], 1.0995 / fam["density"] * 1e3, 1.0995 / fam["density"] * 1e3 / PEAK_F32, fam["mc_"]))
