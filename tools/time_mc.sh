#!/bin/bash
# per-kernel times of marching cubes on the bench's own 256^3 density volume (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
OUT=$R/gpurun_out/tmc; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/time_mc.py > $OUT/log.txt 2>&1
f=$(find $OUT -name "*kernel_stats*.csv" | head -1)
echo "$(grep mesh $OUT/log.txt)"
python3 -c "
import csv,sys
tot=0
for r in csv.DictReader(open('$f')):
    if 'mc_' in r['Name']:
        print('   %-36s calls %3s avg %8.1f us' % (r['Name'].split('(')[0][-36:], r['Calls'], float(r['AverageNs'])/1e3)); tot+=float(r['AverageNs'])/1e3
print('   sum of the averages %.1f us' % tot)
"
