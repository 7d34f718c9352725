#!/bin/bash
# Which kernels does hipBLASLt run on TripoSR's three largest GEMM shapes (calibration only)?  The Tensile kernel names encode the
# macro tile (MT), depth-U, wave tiling, split / stream-K (GSU / SK) and the LDS / prefetch schedule.
#   tools/blas_kernel_names.sh   ->  gpurun_out/blas_names/{kernel_trace.csv,kernel_stats.csv,names.txt}
set -e
export TMPDIR=/tmp
OUT=${OUT:-gpurun_out/blas_names}
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o blas -- python3 tools/time_vs_blas.py > $OUT/time_vs_blas.log 2>&1
cp $(find $OUT/raw -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tr = glob.glob(out + "/raw/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(tr)))
agg = collections.defaultdict(list)
for r in rows:
    agg[(r["Kernel_Name"], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "")))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(out + "/names.txt", "w") as f:
    for (name, grid, wg, lds), ts in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        if not (name.startswith("Cijk") or "gemm" in name.lower()):
            continue
        ts.sort()
        f.write("n=%4d median %7.1f us  grid %s wg %s lds %s  %s\n" % (len(ts), ts[len(ts) // 2] / 1e3, grid, wg, lds, name))
print(open(out + "/names.txt").read())
PY
