#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of the bench command, then separate PMC passes (FETCH_SIZE and
# WRITE_SIZE cannot share a pass; never combined with sys/runtime traces).  Copies the summaries into profiles/<round>/ by hand.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd $R
BENCH="python3 bench.py --no-cpu-baseline --no-optional-modes --no-extras --no-siblings --check-rounds 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH --steps 10 --warmup 2 > $OUT/bench_under_trace.log 2>&1
find $OUT/trace -name "*kernel_stats*.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH --steps 3 --warmup 1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH --steps 3 --warmup 1 > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/pmc_mfma -- $BENCH --steps 3 --warmup 1 > $OUT/pmc_mfma.log 2>&1
python3 tools/summarize_pmc.py $OUT > $OUT/pmc_summary.txt 2>&1
python3 tools/pmc_to_json.py $OUT $OUT/pmc_density_grid.json > $OUT/pmc_to_json.log 2>&1
python3 tools/kernel_table.py $OUT/kernel_stats.csv 12 > $OUT/kernel_table.md 2>&1
# keep only small files
find $OUT -name "*.csv" -size +8M -delete
cat $OUT/kernel_table.md
cat $OUT/pmc_summary.txt
tail -2 $OUT/bench_under_trace.log | cut -c1-600
