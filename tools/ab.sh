#!/bin/bash
# usage (on the GPU box): tools/ab.sh CONFIG NAME1 NAME2 ...   -> per-kernel averages of every variant, same box, interleaved twice
CFG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = "base" ]; then unset LRBMS_HIP_LIB; else export LRBMS_HIP_LIB=$GRAFT_REPO_ROOT/pylrbms_amd/_variants/$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_${v}_$rep -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-online --config $CFG > gpurun_out/ab_${v}_$rep.log 2>&1
  python3 -c "
import csv,glob,json
rows=list(csv.DictReader(open(glob.glob('gpurun_out/ab_${v}_$rep/*/*kernel_stats.csv')[0])))
ms=[json.loads(l)['ms_per_step'] for l in open('gpurun_out/ab_${v}_$rep.log') if l.startswith('{')][-1]
print('%-10s rep$rep ms/step %.3f | ' % ('$v', ms) + ' | '.join('%s %.0f' % (r['Name'].split('::')[-1].split('(')[0][:12], float(r['AverageNs'])/1e3) for r in rows[:8]))
"
done
done
