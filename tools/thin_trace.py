"""Cycle stamps of the three workgroups (side 1, subdomain 500) of k_thin3 (experiment build -DTHIN_TRACE: tools/build_variant.sh ttrace
-DTHIN_TRACE).  usage (GPU box): LRBMS_HIP_LIB=.../_variants/ttrace.so python tools/thin_trace.py"""
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
out = np.zeros((3, 8), dtype=np.uint64)
lib.lrbms_debug_thin_trace.argtypes = [ctypes.c_void_p]
rc = lib.lrbms_debug_thin_trace(out.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
names = [('coupling', ['start', 'rows requested + staged', 'barrier', 'MFMA + stores issued']),
         ('thin_rt', ['start', 'face tables resolved', 'barrier', 'factor rows stored', 'r_fd stored']),
         ('thin_ncf', ['start', 'tables staged', 'barrier', 'Y rows staged', 'barrier', 'factor rows stored'])]
for z, (nm, st) in enumerate(names):
    t0 = int(out[z, 0])
    print(nm + ': ' + ' | '.join('{} {}'.format(s, int(out[z, k]) - t0) for k, s in enumerate(st)))
