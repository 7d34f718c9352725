#!/bin/bash
# rocprofv3 kernel-trace stats of a short bench run (gpurun): prints the top rows
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pq; rm -rf $OUT; mkdir -p $OUT; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-optional-modes --no-extras > $OUT/bench.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats*.csv" | head -1); cp $f $OUT/kernel_stats.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/kernel_stats.csv")))
for r in rows[:28]:
    print("%-70s calls %5s avg %9.1f us  %5.2f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
find $OUT -name "*.csv" -size +4M -delete
