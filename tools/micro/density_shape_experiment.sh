#!/bin/bash
# Wall-clock effect of the 16x16x32 bf16 MFMA shape on the REAL density kernel's instruction stream (values are wrong by
# construction): the library is rebuilt with -DSCULPT_L3_SHAPE_EXPERIMENT in a scratch copy and timed against the product build.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python3 tools/time_density.py --modes bf16l3 --rounds 8 2>&1 | tail -1
rm -rf /tmp/shape_exp && mkdir /tmp/shape_exp && cp -r sculptmate_amd oracle tools include tests /tmp/shape_exp/ 2>/dev/null
cd /tmp/shape_exp
SCULPT_EXTRA_HIPCC_FLAGS="-DSCULPT_L3_SHAPE_EXPERIMENT" python3 -m sculptmate_amd.build --force > /tmp/shape_exp/build.log 2>&1 || { tail -5 /tmp/shape_exp/build.log; exit 1; }
SCULPT_EXTRA_HIPCC_FLAGS="-DSCULPT_L3_SHAPE_EXPERIMENT" python3 tools/time_density.py --modes bf16l3 --rounds 8 2>&1 | tail -1
cd $R
python3 tools/time_density.py --modes bf16l3 --rounds 8 2>&1 | tail -1
