#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for cfg in 0:0 1:0 0:1; do
  dbg=${cfg%%:*}; export SCULPT_MC_CLASSIFY_ROWS=${cfg##*:}
  OUT=$R/gpurun_out/tmc$dbg; rm -rf $OUT; mkdir -p $OUT
  export SCULPT_MC_DBG=$dbg
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/time_mc.py > $OUT/log.txt 2>&1
  f=$(find $OUT -name "*kernel_stats*.csv" | head -1)
  echo "DBG=$dbg ROWS=$SCULPT_MC_CLASSIFY_ROWS $(grep mesh $OUT/log.txt)"
  python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'mc_' in r['Name']: print('   %-36s calls %3s avg %8.1f us' % (r['Name'].split('(')[0][-36:], r['Calls'], float(r['AverageNs'])/1e3))
"
  rm -rf $OUT
done
