#!/bin/bash
# PMC passes over the backbone's attention shapes (run through gpurun); prints per-kernel medians
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_attn; rm -rf $OUT; mkdir -p $OUT; cd $R
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/a -- python3 tools/attn_pmc_run.py ${SHAPE:-3072,3072,16} > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/b -- python3 tools/attn_pmc_run.py ${SHAPE:-3072,3072,16} > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/c -- python3 tools/attn_pmc_run.py ${SHAPE:-3072,3072,16} > $OUT/c.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("a","b","c"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "attention_" in r["Kernel_Name"]:
                key = (r["Kernel_Name"].split("(")[0][-28:], r.get("Grid_Size", "?"), r["Counter_Name"])
                agg[key].append(float(r["Counter_Value"]))
        for k in sorted(agg): v = agg[k]; print(d, k, "n", len(v), "median %.4g" % sorted(v)[len(v)//2])
    for f in glob.glob("$OUT/%s/**/*kernel_trace.csv" % d, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "attention_" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"].split("(")[0][-28:], r.get("Grid_Size", "?"))].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
        for k in sorted(agg): v = agg[k]; print(d, k, "duration median ns", sorted(v)[len(v)//2])
PY
tail -n 3 $OUT/a.log $OUT/b.log $OUT/c.log
