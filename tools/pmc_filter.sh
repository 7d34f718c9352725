#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate --pmc passes, kernel trace only) of the kernels of the two-pass density grid inside the bench
# command: per kernel the median over its full-size (256^3) launches, raw KB (FETCH_SIZE x 2 = bytes, MI355X_MICROARCH.md).
#   tools/pmc_filter.sh [FETCH_SIZE|WRITE_SIZE ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_filter; rm -rf $OUT; mkdir -p $OUT; cd $R
CTRS=${@:-FETCH_SIZE WRITE_SIZE}
for c in $CTRS; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-optional-modes --no-extras --no-siblings --check-rounds 0 > $OUT/$c.log 2>&1
done
python3 - $CTRS <<PY
import csv, glob, collections, sys
for c in sys.argv[1:]:
    by = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].split("(")[0].replace("void sculpt::", "")
            if any(k in n for k in ("density_", "filter_", "mc_")):
                by[n[:60]].append(float(r["Counter_Value"]))
    for n, v in sorted(by.items()):
        big = sorted(x for x in v if x > 0.5 * max(v))
        print(c, n, "full-size launches", len(big), "median KB", big[len(big) // 2])
PY
