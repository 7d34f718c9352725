#!/bin/bash
# Wall-clock effect of the two priced levers of the three-limb density kernel on its REAL instruction stream (values are wrong by
# construction in every experiment build): (a) every 32x32x16 MFMA replaced by two 16x16x32 of the same FLOPs
# (-DSCULPT_L3_SHAPE_EXPERIMENT), (b) every third-limb fragment read from LDS instead of L2 (-DSCULPT_L3_W3_LDS_EXPERIMENT: an
# upper bound -- only three of the eight layers' W3 fit beside W1 | W2), (c) both.  The library is rebuilt in a scratch copy per
# variant and timed against the product build before and after (tools/time_density.py, 256^3, median of 8 rounds).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
echo "product build         : $(python3 tools/time_density.py --modes bf16l3 --rounds 8 2>&1 | tail -1)"
for v in "-DSCULPT_L3_SHAPE_EXPERIMENT" "-DSCULPT_L3_W3_LDS_EXPERIMENT" "-DSCULPT_L3_SHAPE_EXPERIMENT -DSCULPT_L3_W3_LDS_EXPERIMENT"; do
  rm -rf /tmp/shape_exp && mkdir /tmp/shape_exp && cp -r sculptmate_amd oracle tools include tests /tmp/shape_exp/ 2>/dev/null
  cd /tmp/shape_exp
  SCULPT_EXTRA_HIPCC_FLAGS="$v" python3 -m sculptmate_amd.build --force > /tmp/shape_exp/build.log 2>&1 || { tail -5 /tmp/shape_exp/build.log; exit 1; }
  echo "$v : $(SCULPT_EXTRA_HIPCC_FLAGS="$v" python3 tools/time_density.py --modes bf16l3 --rounds 8 2>&1 | tail -1)"
  cd $R
done
echo "product build (again) : $(python3 tools/time_density.py --modes bf16l3 --rounds 8 2>&1 | tail -1)"
