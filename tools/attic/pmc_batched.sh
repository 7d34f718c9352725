#!/bin/bash
# Matrix-pipe and LDS-conflict counters of a 4-image batched bf16 forward (one --pmc pass, kernel trace only): per kernel name and
# grid -- launches, median us, matrix pipe busy, executed TFLOP per launch, TFLOP/s, vector-issue share, LDS bank-conflict cycles
# over SQ busy cycles.  (The b4 half of tools/pmc_round4.sh.)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_b4; rm -rf $OUT; mkdir -p $OUT; cd $R
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/b4 -- python3 tools/time_batched.py --prof 4 > $OUT/b4.log 2>&1
python3 - <<PY
import csv, glob, collections
disp = {}
for f in glob.glob("$OUT/b4/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d = disp.setdefault(r["Dispatch_Id"], {"k": (r["Kernel_Name"].split("(")[0].replace("void sculpt::", ""), r["Grid_Size"]),
                                              "dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
by = collections.defaultdict(list)
for d in disp.values():
    by[d["k"]].append(d)
rows = []
for k, ds in by.items():
    med = lambda n: sorted(x.get(n, 0.0) for x in ds)[len(ds) // 2]
    d = med("dur")
    gui = med("GRBM_GUI_ACTIVE") / 8.0
    busy = med("SQ_VALU_MFMA_BUSY_CYCLES") / 1024.0
    fl = med("SQ_INSTS_VALU_MFMA_MOPS_BF16") * 512
    rows.append((d * len(ds), k, len(ds), d, busy / gui if gui else 0, fl, fl / d / 1e3 if d else 0,
                 med("SQ_ACTIVE_INST_VALU") * 4 / 1024.0 / gui if gui else 0, med("SQ_LDS_BANK_CONFLICT") / med("SQ_BUSY_CYCLES") if med("SQ_BUSY_CYCLES") else 0))
print("kernel | grid | launches | median us | matrix pipe busy | executed TFLOP | TFLOP/s | VALU issue share | LDS conflict cycles / busy")
for tot, k, n, d, b, fl, tf, va, lc in sorted(rows, reverse=True)[:12]:
    print("%-52s grid %-9s n %4d | %8.1f us | %4.0f %% | %7.3f | %6.0f | %4.0f %% | %.3f" % (k[0][:52], k[1], n, d / 1e3, 100 * b, fl / 1e12, tf, 100 * va, lc))
PY
