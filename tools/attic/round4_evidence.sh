#!/bin/bash
# Round-4 evidence run (on the GPU box, via gpurun): every GEMM / mode number quoted in DESIGN.md as a file.  Outputs land in
# gpurun_out/r4ev/ and are copied into profiles/round4/ by hand.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4ev; rm -rf $OUT; mkdir -p $OUT; cd $R
python3 tools/gemm_order_ab.py 2>&1 | grep -v amdgpu.ids > $OUT/gemm_variants_ab.txt
bash tools/gemm_ksweep.sh 2>&1 | grep -v amdgpu.ids > $OUT/gemm_ksweep.txt
python3 tools/gemm_m_sweep.py 2>&1 | grep -v amdgpu.ids > $OUT/gemm_m_sweep.txt
python3 tools/time_vs_blas.py 2>&1 | grep -v amdgpu.ids > $OUT/vs_hipblaslt_calibration.txt
python3 tools/time_batched.py 2>&1 | grep -v amdgpu.ids > $OUT/batched_forward.txt
python3 tools/time_parity_modes.py 2>&1 | grep -v amdgpu.ids > $OUT/parity_modes.txt
# rocprofv3 rows: the batched pass (B = 4), the TripoSR forward with the 256-row kernel forced, the bf16l3 parity mode, SF3D
cd /tmp
for tag in b4 b1_g256 l3; do
  case $tag in
    b4) env_="" ; cmd="tools/time_batched.py --prof 4" ;;
    b1_g256) export SCULPT_GEMM_256=2 SCULPT_GEMM_192=0; cmd="tools/time_batched.py --prof 1" ;;
    l3) unset SCULPT_GEMM_256 SCULPT_GEMM_192; cmd="tools/time_parity_modes.py --prof bf16l3" ;;
  esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$tag -o t -- python3 $R/$cmd > /dev/null 2>&1
  unset SCULPT_GEMM_256 SCULPT_GEMM_192
  f=$(find $OUT/trace_$tag -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats_$tag.csv; rm -rf $OUT/trace_$tag
done
cd $R
bash tools/sf3d_kernel_stats.sh > $OUT/sf3d_kernel_table.txt 2>&1
f=$(find /tmp/sf3d_prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats_sf3d.csv
bash tools/micro/density_shape_experiment.sh > $OUT/density_levers.txt 2>&1
ls -la $OUT
