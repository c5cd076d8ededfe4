#!/bin/bash
# usage: tools/time_kernels_cfg.sh TAG CONFIG  -> avg us per kernel of one bench config (rocprofv3 kernel stats)
TAG=$1; CFG=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-online --config $CFG > gpurun_out/prof_$TAG.log 2>&1
python3 -c "
import csv,glob
rows=list(csv.DictReader(open(glob.glob('gpurun_out/prof_$TAG/*/*kernel_stats.csv')[0])))
for r in rows[:14]:
    print('%-60s calls %5s avg_us %8.1f' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
"
tail -1 gpurun_out/prof_$TAG.log | cut -c1-200
