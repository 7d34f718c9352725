#!/bin/bash
# Vector-L1 (TCP) counters of the dense density kernel: do the third-limb fragment reads hit L1?   gpurun -- 'bash tools/pmc_density_cache.sh'
# (one pass only: a second pass with TCC_* counters aborted inside rocprofv3 on this pool and hung until the time limit)
MODE=${1:-bf16l3}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_cache_$MODE; rm -rf $OUT; mkdir -p $OUT; cd $R
timeout 300 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum --output-format csv -d $OUT/a -- python3 tools/time_density.py --modes $MODE --rounds 2 > $OUT/a.log 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/a/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "density_grid" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()): print(k[0], k[1], "n", len(v), "median %.4g" % sorted(v)[len(v)//2])
PY
