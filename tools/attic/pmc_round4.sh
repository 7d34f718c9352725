#!/bin/bash
# Matrix-pipe counters of the round-4 paths (gpurun): the three-limb parity mode's forward and a 4-image batched bf16 forward.
# One --pmc pass each with --kernel-trace only; per kernel name: launches, median duration, matrix-pipe busy share
# (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs), executed bf16 MFMA FLOPs (MOPS x 512).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_r4; rm -rf $OUT; mkdir -p $OUT; cd $R
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/l3 -- python3 tools/time_parity_modes.py --prof bf16l3 > $OUT/l3.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/b4 -- python3 tools/time_batched.py --prof 4 > $OUT/b4.log 2>&1
python3 - <<PY
import csv, glob, collections
for tag in ("l3", "b4"):
    print("==", tag, "== (per kernel name and grid: launches | median us | matrix pipe busy | executed TFLOP per launch | TFLOP/s | VALU issue share | LDS conflict cycles / busy)")
    dur = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*kernel_trace.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            dur[(r["Kernel_Name"].split("(")[0].replace("void sculpt::", ""), r.get("Grid_Size", "?"))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            cnt[(r["Kernel_Name"].split("(")[0].replace("void sculpt::", ""), r.get("Grid_Size", "?"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows = []
    for k, c in cnt.items():
        med = lambda n: sorted(c[n])[len(c[n]) // 2] if c.get(n) else 0.0
        d = sorted(dur[k])[len(dur[k]) // 2] if dur.get(k) else 0
        gui = med("GRBM_GUI_ACTIVE") / 8.0
        busy = med("SQ_VALU_MFMA_BUSY_CYCLES") / 1024.0
        fl = med("SQ_INSTS_VALU_MFMA_MOPS_BF16") * 512
        rows.append((d * len(dur.get(k, [])), k, len(dur.get(k, [])), d, busy / gui if gui else 0, fl, fl / d / 1e3 if d else 0,
                     med("SQ_ACTIVE_INST_VALU") * 4 / 1024.0 / gui if gui else 0, med("SQ_LDS_BANK_CONFLICT") / med("SQ_BUSY_CYCLES") if med("SQ_BUSY_CYCLES") else 0))
    for tot, k, n, d, b, fl, tf, va, lc in sorted(rows, reverse=True)[:14]:
        print("%-52s grid %-9s n %4d | %8.1f us | %4.0f %% | %7.3f | %6.0f | %4.0f %% | %.3f" % (k[0][:52], k[1], n, d / 1e3, 100 * b, fl / 1e12, tf, 100 * va, lc))
PY
