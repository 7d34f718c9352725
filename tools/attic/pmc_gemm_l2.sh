cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_l2; rm -rf $OUT; mkdir -p $OUT; cd $R
for gm in auto 0; do
  if [ $gm = 0 ]; then export SCULPT_GEMM_GM=0; else unset SCULPT_GEMM_GM; fi
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/b1_$gm -- python3 tools/time_batched.py --prof 1 > $OUT/b1_$gm.log 2>&1
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/b4_$gm -- python3 tools/time_batched.py --prof 4 > $OUT/b4_$gm.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for tag in ("b1_auto", "b1_0", "b4_auto", "b4_0"):
    disp = {}
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            d = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"].split("(")[0].replace("void sculpt::", ""), "grid": r["Grid_Size"], "dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "c": {}})
            d["c"][r["Counter_Name"]] = float(r["Counter_Value"])
    agg = collections.defaultdict(list)
    for d in disp.values():
        if "gemm" in d["name"]: agg[(d["name"], d["grid"])].append(d)
    print("==", tag)
    rows = []
    for k, ds in agg.items():
        med = lambda n: sorted(d["c"].get(n, 0.0) for d in ds)[len(ds) // 2]
        dur = sorted(d["dur"] for d in ds)[len(ds) // 2]
        hit, miss = med("TCC_HIT_sum"), med("TCC_MISS_sum")
        rows.append((dur * len(ds), k, len(ds), dur, hit / (hit + miss + 1e-9), med("TCP_TCC_READ_REQ_sum") * 128 / 1e6, med("TCC_EA0_RDREQ_sum") * 64 / 1e6))
    for tot, k, n, dur, hr, l1l2, ea in sorted(rows, reverse=True)[:8]:
        print("%-44s grid %-8s n %3d | %7.1f us | L2 hit %.2f | L1->L2 read %7.1f MB | L2->fabric read (x64 B raw) %7.1f MB" % (k[0][:44], k[1], n, dur / 1e3, hr, l1l2, ea))
PY
