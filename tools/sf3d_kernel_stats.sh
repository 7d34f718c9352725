#!/bin/bash
# rocprofv3 kernel statistics of the SF3D image -> mesh path (tools/bench_sf3d.py): which attention / GEMM shapes carry config 4
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; rm -rf /tmp/sf3d_prof; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sf3d_prof -- python3 tools/bench_sf3d.py > /tmp/sf3d_prof.log 2>&1
tail -n 4 /tmp/sf3d_prof.log | cut -c1-300
f=$(find /tmp/sf3d_prof -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print("%-70s calls %5s avg %8.1f us  share %4.1f %%" % (r["Name"].split("(")[0][-70:], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
