#!/bin/bash
# F1 ablations: prints the average k_f1 time for LRBMS_F1_DBG = 0, 1 (no staging math), 2 (no MFMA), 4 (no prefetch loads), 3, 7
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in ${ABL:-0 1 2 4 3 6 7}; do
  export LRBMS_F1_DBG=$d
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl_$d -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-online > gpurun_out/abl_$d.log 2>&1
  python3 -c "
import csv,glob
rows=list(csv.DictReader(open(glob.glob('gpurun_out/abl_$d/*/*kernel_stats.csv')[0])))
for r in rows:
    if 'k_f1' in r['Name']: print('dbg=$d k_f1 avg_us %.1f' % (float(r['AverageNs'])/1e3))
"
done
