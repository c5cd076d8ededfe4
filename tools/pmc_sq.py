"""Per-kernel SQ counters (rocprofv3 --pmc, one or more passes) -> MFMA-busy and issue-stall fractions.

usage: pmc_sq.py DIR [DIR...]   prints a table: kernel, calls, and for every counter its per-call average, plus
MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES (when both were collected) and
SQ_INSTS_VALU_MFMA_* counts per call."""
import collections
import csv
import glob
import re
import sys

tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.Counter())
names = []
for d in sys.argv[1:]:
    for f in glob.glob(d + '/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            m = re.search(r'(k_[a-z0-9_]+)', r['Kernel_Name'])
            k = m.group(1) if m else r['Kernel_Name'][:24]
            c = r['Counter_Name']
            if c not in names:
                names.append(c)
            tot[k][c] += float(r['Counter_Value'])
            calls[k][c] += 1
print('{:20s} {:>5s} '.format('kernel', 'calls') + ' '.join('{:>26s}'.format(c) for c in names))
for k in sorted(tot):
    if not k.startswith('k_'):
        continue
    row = '{:20s} {:5d} '.format(k, max(calls[k].values()))
    avg = {c: tot[k][c] / calls[k][c] for c in names if calls[k][c]}
    row += ' '.join('{:26.4g}'.format(avg.get(c, float('nan'))) for c in names)
    extra = []
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in avg and 'SQ_BUSY_CU_CYCLES' in avg and avg['SQ_BUSY_CU_CYCLES']:
        extra.append('mfma_busy/cu_busy={:.3f}'.format(avg['SQ_VALU_MFMA_BUSY_CYCLES'] / avg['SQ_BUSY_CU_CYCLES']))
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in avg and 'SQ_BUSY_CYCLES' in avg and avg['SQ_BUSY_CYCLES']:
        extra.append('mfma_busy/sq_busy={:.3f}'.format(avg['SQ_VALU_MFMA_BUSY_CYCLES'] / avg['SQ_BUSY_CYCLES']))
    if 'SQ_WAVE_CYCLES' in avg and avg['SQ_WAVE_CYCLES']:
        for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_INST_LDS'):
            if c in avg:
                extra.append('{}/wave_cycles={:.3f}'.format(c[3:].lower(), avg[c] / avg['SQ_WAVE_CYCLES']))
    print(row + '  ' + ' '.join(extra))
