"""Sum FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc, KiB units, separate passes) over the kernels of bench.py's passes.

usage: pmc_traffic.py CONFIG OUT.json DIR [DIR...]

Prints the per-kernel table and writes OUT.json = {config, per_kernel: {name: {fetch_bytes_raw, write_bytes, bytes}},
per_pass_bytes}, which bench.py loads as ``roofline.traffic``.  FETCH_SIZE is doubled as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (the counter tallies 128-B requests at 64 B)."""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pylrbms_amd._build import source_sha  # noqa: E402

config, out_json = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for d in sys.argv[3:]:
    f = glob.glob(d + '/*/*counter_collection.csv')[0]
    for r in csv.DictReader(open(f)):
        m = re.search(r'(k_[a-z0-9_]+)', r['Kernel_Name'])
        k = m.group(1) if m else r['Kernel_Name'][:24]
        k = {'k_f1u': 'k_f1', 'k_f1v': 'k_f1', 'k_f1w': 'k_f1'}.get(k, k)   # the forms of the projection kernel are the same step of the pass (bench.py: k_f1)
        if r['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'):
            tot[k][r['Counter_Name']] += float(r['Counter_Value'])
            calls[(k, r['Counter_Name'])] += 1
print('{:24s} {:>6s} {:>18s} {:>18s}'.format('kernel', 'calls', 'FETCH_SIZE MiB/call', 'WRITE_SIZE MiB/call'))
sf = sw = 0.0
# the kernels of ONE pass in the factored layout (the timed region of bench.py); bench.py also times the dense layout once
# (k_thin_nc, k_thin_rt, k_coupling, k_thin_expand instead of k_thin3): listed in the table, not part of the per-pass sum
PASS_KERNELS = ('k_flux_compact', 'k_vertex_avg', 'k_f1', 'k_f2', 'k_f3', 'k_thin3', 'k_prep_lds')
per_kernel = {}
for k, v in sorted(tot.items()):
    nf, nw = calls[(k, 'FETCH_SIZE')], calls[(k, 'WRITE_SIZE')]
    f = v.get('FETCH_SIZE', 0.0) / nf / 1024 if nf else 0.0
    w = v.get('WRITE_SIZE', 0.0) / nw / 1024 if nw else 0.0
    print('{:24s} {:6d} {:18.1f} {:18.1f}'.format(k, max(nf, nw), f, w))
    if k in PASS_KERNELS:        # one call of each per pass: sum the per-call averages (bench.py also times phase 4 alone)
        sf += f
        sw += w
        per_kernel[k] = {'fetch_bytes_raw': f * 2 ** 20, 'write_bytes': w * 2 ** 20, 'bytes': (2 * f + w) * 2 ** 20}
print('per pass (hot-path kernels): FETCH_SIZE {:.1f} MiB (x2 correction for wide coalesced reads on gfx950: {:.1f} MiB), '
      'WRITE_SIZE {:.1f} MiB'.format(sf, 2 * sf, sw))
with open(out_json, 'w') as fh:
    json.dump({'config': config, 'csrc_sha': source_sha(), 'per_kernel': per_kernel, 'per_pass_bytes': (2 * sf + sw) * 2 ** 20,
               'fetch_correction': 'FETCH_SIZE x 2 (gfx950, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported',
               'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py'}, fh,
              indent=1)
