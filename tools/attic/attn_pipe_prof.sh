#!/bin/bash
# rocprofv3 kernel durations of the two attention tile loops on one shape each (run on the GPU box): tools/attn_pipe_prof.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for sh in 3072,3072,16 3072,1025,16 1025,1025,12; do
  rm -rf /tmp/ap_prof
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ap_prof -- python3 $R/tools/time_attn_pipe.py $sh > /dev/null 2>&1
  f=$(find /tmp/ap_prof -name '*kernel_stats.csv' | head -1)
  python3 - "$f" "$sh" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "attention" in r["Name"]:
        print(sys.argv[2], r["Name"].split("(")[0][-60:], r["Calls"], "avg %.2f us" % (float(r["AverageNs"]) / 1e3))
PY
done
