#!/bin/bash
# PMC counters of the dense density kernels (one mode per run of tools/time_density.py), two SQ passes.
#   gpurun -- 'bash tools/pmc_density_modes.sh bf16l3'
MODE=${1:-bf16l3}; shift; EXTRA="$@"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_$MODE; rm -rf $OUT; mkdir -p $OUT; cd $R
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/a -- python3 tools/time_density.py --modes $MODE --rounds 3 $EXTRA > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/b -- python3 tools/time_density.py --modes $MODE --rounds 3 $EXTRA > $OUT/b.log 2>&1
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
for d in ("a","b"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "density_grid" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()): print(d, k[0], k[1], "n", len(v), "median", sorted(v)[len(v)//2])
    for f in glob.glob("$OUT/%s/**/*kernel_trace.csv" % d, recursive=True):
        ds = [int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "density_grid" in r["Kernel_Name"]]
        print(d, "kernel duration median ns", sorted(ds)[len(ds)//2])
PY
cat $OUT/summary.txt; tail -n 3 $OUT/a.log; tail -n 3 $OUT/b.log
