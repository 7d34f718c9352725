#!/bin/bash
# PMC passes over the GEMM shapes of the backbone (run through gpurun); prints per-kernel medians
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_gemm; rm -rf $OUT; mkdir -p $OUT; cd $R
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/a -- python3 tools/gemm_pmc_run.py > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/b -- python3 tools/gemm_pmc_run.py > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/c -- python3 tools/gemm_pmc_run.py > $OUT/c.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("a","b","c"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "gemm_bf16_kernel" in r["Kernel_Name"]:
                key = (r["Kernel_Name"].split("<")[1].split(">")[0], r.get("Grid_Size", "?"), r["Counter_Name"])
                agg[key].append(float(r["Counter_Value"]))
        for k in sorted(agg): v = agg[k]; print(d, k, "n", len(v), "median %.4g" % sorted(v)[len(v)//2])
    for f in glob.glob("$OUT/%s/**/*kernel_trace.csv" % d, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "gemm_bf16_kernel" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"].split("<")[1].split(">")[0], r.get("Grid_Size", "?"))].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
        for k in sorted(agg): v = agg[k]; print(d, k, "duration median ns", sorted(v)[len(v)//2])
PY
tail -3 $OUT/a.log $OUT/c.log
