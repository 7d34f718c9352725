#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/ksweep; rm -rf $OUT; mkdir -p $OUT; cd $R
for mode in plain residual g256 g256full; do
rocprofv3 --kernel-trace --output-format csv -d $OUT/$mode -- python3 tools/gemm_ksweep.py $mode > $OUT/$mode.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/$mode/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
cur = []
Ks = [64, 128, 256, 512, 1024, 2048, 4096]
i = 0
for r in rows:
    if "gemm_bf16_kernel" in r["Kernel_Name"] or "gemm256" in r["Kernel_Name"]:
        cur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        name = r["Kernel_Name"].split("(")[0]
    elif "FillFunctor" in r["Kernel_Name"] or "fill" in r["Kernel_Name"].lower():
        if cur:
            c = sorted(cur[2:])
            print("$mode K=%d: median %.2f us min %.2f us (%d launches) %s" % (Ks[i], c[len(c)//2], c[0], len(c), name[:60]))
            i += 1
        cur = []
PY
done
