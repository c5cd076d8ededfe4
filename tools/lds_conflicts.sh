#!/bin/bash
# LDS bank-conflict cycles against LDS-active cycles per kernel of the offline pass -> gpurun_out/lds_conflicts.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/ldsc
rm -rf $O
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $O -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O.log 2>&1
python3 - <<'PY' > gpurun_out/lds_conflicts.txt
import csv, glob, collections
f = glob.glob('gpurun_out/ldsc/*/*counter_collection.csv')[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:60]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_INSTS_LDS': n[k] += 1
rows = []
for k, v in acc.items():
    act = v.get('SQ_LDS_IDX_ACTIVE', 0.0)
    if act <= 0: continue
    rows.append((v.get('SQ_LDS_BANK_CONFLICT', 0.0) / act, k, n[k], act / max(n[k], 1), v.get('SQ_INSTS_LDS', 0.0) / max(n[k], 1)))
for frac, k, c, act, ins in sorted(rows, reverse=True):
    print('%-60s calls %5d conflict/active %5.2f  active cycles/launch %12.0f  lds insts/launch %10.0f' % (k, c, frac, act, ins))
PY
cat gpurun_out/lds_conflicts.txt
