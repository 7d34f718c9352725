#!/bin/bash
# SQ counters of the marching-cubes kernels on the bench's 256^3 volume (tools/time_mc.py), one --pmc pass per group.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_mc; rm -rf $OUT; mkdir -p $OUT; cd $R
GROUPS_=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU" "FETCH_SIZE" "WRITE_SIZE")
i=0
for g in "${GROUPS_[@]}"; do
  rocprofv3 --kernel-trace --pmc $g --output-format csv -d $OUT/g$i -- python3 tools/time_mc.py > $OUT/g$i.log 2>&1
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections
by = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void sculpt::", "").replace("sculpt::", "")
        if "mc_" in n:
            by[n[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in sorted(by.items()):
    print(n)
    for c, v in sorted(d.items()):
        v = sorted(v)
        print("    %-26s median %14.0f  (n=%d)" % (c, v[len(v) // 2], len(v)))
PY
