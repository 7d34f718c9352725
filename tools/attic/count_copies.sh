#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; rm -rf /tmp/cc; cd $R
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/cc -- python3 tools/count_copies.py 8 > /tmp/cc.log 2>&1
tail -n 5 /tmp/cc.log | cut -c1-300
python3 - <<'PY'
import csv, glob, collections
kt = glob.glob("/tmp/cc/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(kt)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# everything after the LAST launch whose grid is the marker's reduction is phase 2
faces = [i for i, r in enumerate(rows) if "mc_faces_kernel" in r["Kernel_Name"]]
print("kernels", len(rows), "images", len(faces))
ph2 = rows[faces[-9] + 1:]   # the last 8 images
agg = collections.Counter(); dur = collections.Counter()
for r in ph2:
    n = r["Kernel_Name"].split("(")[0][-60:]
    agg[n] += 1; dur[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for n, c in agg.most_common():
    if "copy" in n.lower() or "fill" in n.lower() or "elementwise" in n.lower() or "at::" in n:
        print("%-62s %6.1f / image  %7.1f us / image" % (n, c / 8, dur[n] / 8 / 1e3))
print("kernels / image: %.1f" % (len(ph2) / 8))
mc = glob.glob("/tmp/cc/**/*memory_copy_trace.csv", recursive=True)
if mc:
    m = list(csv.DictReader(open(mc[0])))
    t0 = int(ph2[0]["Start_Timestamp"])
    m2 = [r for r in m if int(r["Start_Timestamp"]) >= t0]
    c = collections.Counter((r["Direction"], r.get("Bytes", r.get("Size", "?"))) for r in m2)
    for k, v in c.most_common(12): print(k, "%.1f / image" % (v / 8))
PY
