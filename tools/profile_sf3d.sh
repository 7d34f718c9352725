#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of the full-size SF3D stage bench.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_sf3d
rm -rf $OUT; mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_sf3d.py > $OUT/bench_under_trace.log 2>&1
find $OUT/trace -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT -name "*.csv" -size +8M -delete
tail -5 $OUT/bench_under_trace.log
head -40 $OUT/kernel_stats.csv
