#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in whole overlap; do
rm -rf gpurun_out/tr_$m
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_$m -- python3 tools/overlap_trace.py $m > gpurun_out/tr_$m.log 2>&1
python3 - <<PY
import csv,glob
rows=list(csv.DictReader(open(glob.glob('gpurun_out/tr_$m/*/*kernel_trace.csv')[0])))
rows=[r for r in rows if 'assemble' not in r['Kernel_Name'] and 'build_tables' not in r['Kernel_Name'] and 'rocclr' not in r['Kernel_Name'] and 'at::' not in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last pass: last 8-12 kernels
import re
last=rows[-7:] if '$m'=='overlap' else rows[-4:]
t0=int(last[0]['Start_Timestamp'])
print('$m')
for r in last:
    nm=re.search(r'(k_[a-z0-9_]+)', r['Kernel_Name']).group(1)
    print('  %-16s start %7.1f end %7.1f  queue %s' % (nm, (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-t0)/1e3, r.get('Queue_Id','?')))
PY
done
