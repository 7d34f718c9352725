"""Split the per-dispatch durations of each GEMM instantiation in a rocprofv3 kernel trace by grid size and duration mode
(the same instantiation serves several shapes): python tools/gemm_hist.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
groups = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "gemm_bf16_kernel" not in n:
        continue
    key = (n[n.index("<"):n.index(">") + 1], r["Grid_Size_X"], r["Grid_Size_Y"])
    groups[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for key, d in sorted(groups.items()):
    d.sort()
    # split at the largest gap between sorted durations if it separates two clear modes
    cut, gap = None, 0
    for i in range(1, len(d)):
        if d[i] - d[i - 1] > gap and d[i] > 1.5 * d[i - 1]:
            gap, cut = d[i] - d[i - 1], i
    parts = [d] if cut is None else [d[:cut], d[cut:]]
    for p in parts:
        print("%-24s grid %5s x %-4s  n %5d  median %7.1f us  mean %7.1f  sum %8.2f ms" % (key[0], key[1], key[2], len(p), p[len(p) // 2], sum(p) / len(p), sum(p) / 1e3))
