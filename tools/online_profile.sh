#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/online_prof -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/online_prof.log 2>&1
python3 -c "
import csv,glob
rows=list(csv.DictReader(open(glob.glob('gpurun_out/online_prof/*/*kernel_stats.csv')[0])))
for r in rows[:14]:
    print('%-70s calls %6s avg_us %9.1f total_ms %8.1f' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
"
