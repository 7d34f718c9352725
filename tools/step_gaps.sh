#!/bin/bash
# The bench's step (TSR.forward + extract_meshes on a resident image) under rocprofv3 --kernel-trace: per image the kernels, their
# summed duration, the time between them (gaps), and the largest gaps with the kernels either side.  bash tools/step_gaps.sh [N]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/gaps; rm -rf $OUT; mkdir -p $OUT; cd $R
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 tools/step_only.py ${1:-8} > $OUT/log.txt 2>&1
tail -n 2 $OUT/log.txt | cut -c1-200
python3 - <<PY
import csv, glob, collections
import numpy as np
N = int("${1:-8}")
kt = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(kt)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "mc_emit_brick" in r["Kernel_Name"]]
ph = rows[ends[-N - 1] + 1: ends[-1] + 1]          # the last N images, from the kernel after an emit to the last emit
st = np.array([int(r["Start_Timestamp"]) for r in ph]); en = np.array([int(r["End_Timestamp"]) for r in ph])
span = (en[-1] - st[0]) / 1e3 / N
busy = (en - st).sum() / 1e3 / N
gaps = (st[1:] - en[:-1]) / 1e3
print("last %d images: %.1f kernels / image, span %.1f us / image, kernels %.1f us, between kernels %.1f us (%.1f %%)" % (
    N, len(ph) / N, span, busy, span - busy, 100 * (span - busy) / span))
print("gap between consecutive kernels: median %.2f us, mean %.2f, p90 %.2f; gaps > 8 us: %d / image (%.1f us / image)" % (
    np.median(gaps), gaps.mean(), np.quantile(gaps, 0.9), (gaps > 8).sum() / N, gaps[gaps > 8].sum() / N))
name = lambda r: r["Kernel_Name"].split("(")[0][-48:]
big = collections.Counter(); bigt = collections.Counter()
for i in np.nonzero(gaps > 8)[0]:
    k = (name(ph[i]), name(ph[i + 1])); big[k] += 1; bigt[k] += gaps[i]
for k, c in big.most_common(12):
    print("   %5.1f / image  %7.1f us each   %s  ->  %s" % (c / N, bigt[k] / c, k[0], k[1]))
agg = collections.Counter(); dur = collections.Counter()
for r in ph:
    agg[name(r)] += 1; dur[name(r)] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("kernels by time:")
for n, d in dur.most_common(40):
    print("   %-50s %6.1f / image  %8.1f us / image" % (n, agg[n] / N, d / N / 1e3))
PY
