"""Per-kernel PMC summary of the 3D / P2 pass (rocprofv3 --pmc passes of `bench.py --config cfg5`).

usage: pmc3d.py DIR [DIR...]   -- every DIR is one rocprofv3 -d output; prints per kernel (k3_* only; the template arguments
of k3_pg are reduced to its KIND: 0 SYS, 1 AAA, 2 NC, 3 AB, 4 BB, 6 CPL) the per-call average of every counter collected,
FETCH_SIZE / WRITE_SIZE in MiB (FETCH_SIZE also doubled, as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950),
and the derived MFMA-busy and wait fractions."""
import collections
import csv
import glob
import json
import os
import re
import sys

# optional: --json OUT CONFIG  -> {config, csrc_sha, per_kernel: {name: {fetch_bytes_raw, write_bytes, bytes}}, per_pass_bytes} for bench3d.py
JSON_OUT = None
if '--json' in sys.argv:
    i = sys.argv.index('--json')
    JSON_OUT, JSON_CFG = sys.argv[i + 1], sys.argv[i + 2]
    del sys.argv[i:i + 3]

KINDS = {'0': 'SYS', '1': 'AAA', '2': 'NC', '3': 'AB', '4': 'BB', '5': 'RDD', '6': 'CPL'}
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(collections.Counter)
names = []
for d in sys.argv[1:]:
    for f in glob.glob(d + '/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            m = re.search(r'(k3_[a-z0-9_]+)(<(\d)[^>]*>)?', r['Kernel_Name'])
            if not m:
                continue
            k = m.group(1) + ('<{}>'.format(KINDS.get(m.group(3), m.group(3))) if m.group(3) else '')
            c = r['Counter_Name']
            if c not in names:
                names.append(c)
            tot[k][c] += float(r['Counter_Value'])
            calls[k][c] += 1
print('{:18s} {:>5s} '.format('kernel', 'calls') + ' '.join('{:>24s}'.format(c) for c in names))
for k in sorted(tot):
    avg = {c: tot[k][c] / calls[k][c] for c in names if calls[k][c]}
    row = '{:18s} {:5d} '.format(k, max(calls[k].values())) + ' '.join('{:24.4g}'.format(avg.get(c, float('nan'))) for c in names)
    extra = []
    if 'FETCH_SIZE' in avg:
        extra.append('fetch_MiB={:.1f} (x2: {:.1f})'.format(avg['FETCH_SIZE'] / 1024, avg['FETCH_SIZE'] / 512))
    if 'WRITE_SIZE' in avg:
        extra.append('write_MiB={:.1f}'.format(avg['WRITE_SIZE'] / 1024))
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in avg and avg.get('SQ_BUSY_CU_CYCLES'):
        extra.append('mfma_busy/(4 cu_busy)={:.3f}'.format(avg['SQ_VALU_MFMA_BUSY_CYCLES'] / avg['SQ_BUSY_CU_CYCLES'] / 4))
    if avg.get('SQ_WAVE_CYCLES'):
        for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_SCA'):
            if c in avg:
                extra.append('{}/wave_cycles={:.3f}'.format(c[3:].lower(), avg[c] / avg['SQ_WAVE_CYCLES']))
    print(row + '  ' + ' '.join(extra))
if JSON_OUT:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    from pylrbms_amd._build import source_sha
    per, total = {}, 0.0
    for k in sorted(tot):
        if 'FETCH_SIZE' not in tot[k] or 'WRITE_SIZE' not in tot[k]:
            continue
        f = tot[k]['FETCH_SIZE'] / calls[k]['FETCH_SIZE'] * 1024.0
        w = tot[k]['WRITE_SIZE'] / calls[k]['WRITE_SIZE'] * 1024.0
        per[k] = {'fetch_bytes_raw': f, 'write_bytes': w, 'bytes': 2 * f + w}
        if k.startswith(('k3_pg<', 'k3_flux', 'k3_node_avg', 'k3_side_nc')) and 'combine' not in k:
            total += 2 * f + w
    with open(JSON_OUT, 'w') as fh:
        json.dump({'config': JSON_CFG, 'csrc_sha': source_sha(), 'per_kernel': per, 'per_pass_bytes': total,
                   'fetch_correction': 'FETCH_SIZE x 2 (gfx950, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported',
                   'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --config ' + JSON_CFG},
                  fh, indent=1)
