"""Summarise tools/pmc_round4.sh's counter passes: per kernel name and grid -- launches, median GPU duration (the dispatch's own
timestamps; slower than untraced: counters serialise dispatches), matrix-pipe busy share = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over
GRBM_GUI_ACTIVE / 8 XCDs, executed bf16 MFMA FLOPs = SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512, vector-issue share, LDS conflict share.
    python tools/pmc_round4_summary.py gpurun_out/pmc_r4"""
import collections, csv, glob, sys
root = sys.argv[1]
for tag, what in (("l3", "TSR(precision='bf16l3') forward"), ("b4", "TSR.forward on 4 images, bf16")):
    disp = {}
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (root, tag), recursive=True):
        for r in csv.DictReader(open(f)):
            d = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"].split("(")[0].replace("void sculpt::", "").replace("sculpt::", ""),
                                                   "grid": r["Grid_Size"], "dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "c": {}})
            d["c"][r["Counter_Name"]] = float(r["Counter_Value"])
    agg = collections.defaultdict(list)
    for d in disp.values():
        agg[(d["name"], d["grid"])].append(d)
    print("== %s ==  kernel | grid (work-items) | launches | median us | matrix pipe busy | executed TFLOP / launch | TFLOP/s | vector issue share | LDS conflict / busy" % what)
    rows = []
    for k, ds in agg.items():
        med = lambda fn: sorted(fn(d) for d in ds)[len(ds) // 2]
        dur = med(lambda d: d["dur"])
        gui = med(lambda d: d["c"].get("GRBM_GUI_ACTIVE", 0.0)) / 8.0
        busy = med(lambda d: d["c"].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)) / 1024.0
        fl = med(lambda d: d["c"].get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)) * 512
        va = med(lambda d: d["c"].get("SQ_ACTIVE_INST_VALU", 0.0)) * 4 / 1024.0
        sqb = med(lambda d: d["c"].get("SQ_BUSY_CYCLES", 0.0))
        lc = med(lambda d: d["c"].get("SQ_LDS_BANK_CONFLICT", 0.0))
        rows.append((dur * len(ds), k, len(ds), dur, busy / gui if gui else 0, fl, va / gui if gui else 0, lc / sqb if sqb else 0))
    for tot, k, n, dur, b, fl, va, lc in sorted(rows, reverse=True)[:12]:
        print("%-44s | %9s | %4d | %8.1f | %3.0f %% | %6.3f | %5.0f | %3.0f %% | %.3f" % (k[0][:44], k[1], n, dur / 1e3, 100 * b, fl / 1e12, fl / dur / 1e3 if dur else 0, 100 * va, lc))
