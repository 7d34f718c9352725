#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of the bench command, then separate PMC passes.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-optional-modes > $OUT/bench_under_trace.log 2>&1
find $OUT/trace -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-optional-modes > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-optional-modes > $OUT/pmc_write.log 2>&1
python3 tools/summarize_pmc.py $OUT > $OUT/pmc_summary.txt 2>&1
python3 tools/pmc_to_json.py $OUT $OUT/pmc_density_grid.json > $OUT/pmc_to_json.log 2>&1
# keep only small files
find $OUT -name "*.csv" -size +8M -delete
ls -la $OUT $OUT/trace/* | head -40
head -30 $OUT/kernel_stats.csv
cat $OUT/pmc_summary.txt
