"""Cycle stamps of k_f1 (experiment build -DF1_TRACE: tools/build_variant.sh trace -DF1_TRACE): where a chunk's time goes
in producer wave 0 and consumer wave 4 of workgroup 0.  usage (GPU box): LRBMS_HIP_LIB=.../_variants/trace.so python tools/f1_trace.py"""
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
if os.environ.get('F1_PRODUCER_CONSUMER'):      # tool-side switch -> context option
    eng.ctx.set_option('f1_form', 1)
for _ in range(3):
    eng.project_and_estimate(V, buf)
torch.cuda.synchronize()
lib = eng.ctx.lib
out = np.zeros((2, 64, 8), dtype=np.uint64)
lib.lrbms_debug_f1_trace.argtypes = [ctypes.c_void_p]
rc = lib.lrbms_debug_f1_trace(out.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
nch = eng.t.n_T // 4
A, B = out[0, :nch].astype(np.int64), out[1, :nch].astype(np.int64)
t0 = min(A[0, 0], B[0, 0])
names = 'k0 k1 k2 k3 barrier_passed mfma_done'
if os.environ.get('F1_PRODUCER_CONSUMER'):
    print('legacy producer/consumer kernel: stamps of producer wave 0 (6) and consumer wave 4 (3: at_barrier, barrier_passed, mfma_done)')
print('chunk | wave 0 (role A): ' + names + ' | wave 4 (role B): ' + names)
for c in range(nch):
    print('{:3d} | {}  | {}'.format(c, ' '.join('{:8d}'.format(int(x - t0)) for x in A[c, :6]), ' '.join('{:8d}'.format(int(x - t0)) for x in B[c, :6])))
d = lambda a: float(np.mean(a[2:-2]))  # noqa: E731
for tag, W in (('role A (wave 0)', A), ('role B (wave 4)', B)):
    print('{}: k0->k1 {:.0f} | k1->k2 {:.0f} | k2->k3 {:.0f} | k3->barrier passed {:.0f} | MFMA phase {:.0f} | chunk period {:.0f}'.format(
        tag, d(W[:, 1] - W[:, 0]), d(W[:, 2] - W[:, 1]), d(W[:, 3] - W[:, 2]), d(W[:, 4] - W[:, 3]), d(W[:, 5] - W[:, 4]), d(W[1:, 0] - W[:-1, 0])))
print('k_f1u: k0 stage start, k1 loads issued, k2 prefetched set landed, k3 staged.  k_f1v role A: k0 stage start, k1 set landed, k2 applied + '
      'stored, k3 next loads issued; role B as k_f1u')
