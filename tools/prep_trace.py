"""Cycle stamps of k_prep_lds (experiment build -DPREP_TRACE: tools/build_variant.sh ptrace -DPREP_TRACE): where the time of one
workgroup goes.  usage (GPU box): LRBMS_HIP_LIB=.../_variants/ptrace.so python tools/prep_trace.py"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.argv = [sys.argv[0]]
import torch  # noqa: E402
from bench import CONFIGS, make_bases_host  # noqa: E402
from pylrbms_amd import multiscale_problem  # noqa: E402
from pylrbms_amd.engine import Engine  # noqa: E402

cfg = CONFIGS[os.environ.get('CFG', 'cfg3')]
p = multiscale_problem.init_grid_and_problem({'num_subdomains': cfg['num_subdomains'], 'coarse_per_subdomain': cfg['coarse_per_subdomain']})
lam = p['lambda']
eng = Engine(p['grid'], lam['functions'], p['kappa'], p['f'], p['lambda_bar'], p['lambda_hat'],
             np.array([c.evaluate(p['mu_bar']) for c in lam['coefficients']])).assemble()
N = cfg['N']
V = eng.ctx.from_numpy(make_bases_host(eng.local, eng.t.n, N))
buf = eng.alloc_reduce_buffers(N)
for _ in range(3):
    eng.project_and_estimate(V, buf)
torch.cuda.synchronize()
lib = eng.ctx.lib
out = np.zeros((2, 16), dtype=np.uint64)
lib.lrbms_debug_prep_trace.argtypes = [ctypes.c_void_p]
rc = lib.lrbms_debug_prep_trace(out.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
names = ['start', 'loads issued + staged', 'barrier 0', 'flux rows done', 'barrier 1 (Fl dead)', 'averages done', 'barrier 2 (Al complete)',
         'Z rows done', 'barrier 3', 'MFMA done', 'barrier 4', 'partials written + barrier 5', 'end', 'top of the loop (persistent form: third subdomain)',
         'G_nc stored, in front of the refill']
t0 = int(out[0, 0])
print('k_prep_lds, workgroup 5: s_memtime stamps relative to the start of wave 0 (wave 0: own rows | last wave: neighbours\' shares)')
order = list(range(len(names)))
if int(out[0, 13]) > int(out[0, 1]):       # persistent form: the stamps 2 .. 14 belong to the workgroup's third subdomain
    order = [0, 1, 13] + list(range(2, 13)) + [14]
    t0 = int(out[0, 13])
    print('(persistent: relative to the top of the loop for the third subdomain; 0 / 1 are the first subdomain\'s load phase)')
for k in order:
    nm = names[k]
    print('{:2d} {:32s} {:10d} {:10d}'.format(k, nm, int(out[0, k]) - t0, int(out[1, k]) - t0))
