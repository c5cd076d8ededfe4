#!/bin/bash
# kernel durations of the Krylov loops (tools/solve_time.py) -> gpurun_out/solve_prof.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/solve_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/solve_prof -- python3 tools/solve_time.py 32 32 40 10 > gpurun_out/solve_prof.log 2>&1
python3 -c "
import csv,glob
rows=list(csv.DictReader(open(glob.glob('gpurun_out/solve_prof/*/*kernel_stats.csv')[0])))
for r in rows[:24]:
    print('%-70s calls %6s avg_us %9.1f total_ms %8.1f' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
" > gpurun_out/solve_prof.txt
cat gpurun_out/solve_prof.txt
