#!/bin/bash
# Round-5 evidence, all from the code in this tree (run through gpurun; summaries are copied into profiles/round5/ by hand):
#   1. tools/profile_bench.sh           kernel trace + PMC passes of the bench command -> gpurun_out/prof/
#   2. tools/time_density_filter.py     the filtered density grid against the full evaluation, pass by pass
#      tools/stress_filter.py          8 images x 3 levels at 256^3 through TSR.extract_meshes, filtered vs unfiltered, bit for bit
#   3. tools/gemm_bm192_ab.py           128 x 64 tiles against one round of 192 x 64 tiles
#   4. tools/blas_kernel_names.sh       which kernels hipBLASLt runs on the big shapes (calibration only)
#   6. tools/time_l3p.py             the tolerance mode's Linears: splitting kernel against limbs once (gemm_l3p), tile forms
#   5. kernel traces of the batched (B = 4) bf16 forward and of the bf16l3 forward (from the final code)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=$R/gpurun_out/r5; rm -rf $OUT; mkdir -p $OUT
tools/profile_bench.sh > $OUT/profile_bench.log 2>&1
python3 tools/time_density_filter.py --rounds 7 > $OUT/density_filter.txt 2>&1
python3 tools/gemm_bm192_ab.py > $OUT/gemm_bm192_ab.txt 2>&1
python3 tools/stress_filter.py 8 --levels > $OUT/stress_filter.txt 2>&1
OUT=$OUT/blas_names tools/blas_kernel_names.sh > $OUT/blas_kernel_names.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b4 -- python3 tools/time_batched.py --prof 4 > $OUT/b4.log 2>&1
cp $(find $OUT/b4 -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_b4.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/l3 -- python3 tools/time_parity_modes.py --prof bf16l3 > $OUT/l3.log 2>&1
cp $(find $OUT/l3 -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_l3.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/h2 -- python3 tools/time_parity_modes.py --prof fp16l2 > $OUT/h2.log 2>&1
cp $(find $OUT/h2 -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_fp16l2.csv
python3 tools/time_batched.py > $OUT/batched_forward.txt 2>&1
tools/pmc_batched.sh > $OUT/pmc_batched.txt 2>&1
python3 tools/time_parity_modes.py > $OUT/parity_modes.txt 2>&1
python3 tools/time_l3p.py > $OUT/l3p_gemm.txt 2>&1
python3 tools/time_l3p_f16_forms.py > $OUT/l3p_f16_forms.txt 2>&1
python3 tools/time_l2_attention.py > $OUT/l2_attention.txt 2>&1
find $OUT -name "*.csv" -size +8M -delete
ls -la $OUT
