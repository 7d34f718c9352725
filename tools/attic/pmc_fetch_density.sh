#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the 256^3 density launch (separate passes), per launch.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_fd; rm -rf $OUT; mkdir -p $OUT; cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 tools/time_kernels.py 256 > $OUT/$c.log 2>&1
done
python3 - <<PY
import csv, glob
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True):
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "density_grid_kernel" in r["Kernel_Name"]]
        v.sort()
        print(c, "launches", len(v), "median KB", v[len(v)//2], "max KB", v[-1])
PY
