#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_density; rm -rf $OUT; mkdir -p $OUT; cd $R
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/a -- python3 tools/time_kernels.py 256 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/b -- python3 tools/time_kernels.py 256 > $OUT/b.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("a","b"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "density_grid_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items(): print(d, k, "n", len(v), "median", sorted(v)[len(v)//2])
    for f in glob.glob("$OUT/%s/**/*kernel_trace.csv" % d, recursive=True):
        ds = [int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "density_grid_kernel" in r["Kernel_Name"]]
        print(d, "kernel duration median ns", sorted(ds)[len(ds)//2])
PY
tail -3 $OUT/a.log
