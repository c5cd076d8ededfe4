#!/bin/bash
# usage: tools/time_kernels.sh TAG [env assignments...]  -> prints avg us per kernel (rocprofv3 kernel stats)
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/prof_$TAG.log 2>&1
python3 -c "
import csv,glob
rows=list(csv.DictReader(open(glob.glob('gpurun_out/prof_$TAG/*/*kernel_stats.csv')[0])))
print('$TAG', ' | '.join('%s %.0f' % (r['Name'].split('::')[-1][:14], float(r['AverageNs'])/1e3) for r in rows[:8]))
"
