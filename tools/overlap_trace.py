"""Runs a few overlapped (sharded-style) passes on cfg3_tile8 for a rocprofv3 kernel trace."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from pylrbms_amd import multiscale_problem
from pylrbms_amd.engine import Engine
cfg = bench.CONFIGS['cfg3_tile8']
p = multiscale_problem.init_grid_and_problem({'num_subdomains': cfg['num_subdomains'], 'coarse_per_subdomain': 4})
lam = p['lambda']
tb = np.array([c.evaluate(p['mu_bar']) for c in lam['coefficients']])
eng = Engine(p['grid'], lam['functions'], p['kappa'], p['f'], p['lambda_bar'], p['lambda_hat'], tb).assemble()
N = 40
V = eng.ctx.from_numpy(bench.make_bases_host(eng.local, eng.t.n, N))
buf = eng.alloc_reduce_buffers(N)


class _NoHalo:
    def start(self, V):
        return lambda: V


mode = sys.argv[1]
for _ in range(6):
    if mode == 'overlap':
        eng.project_and_estimate(V, buf, halo=_NoHalo())
    else:
        eng.project_and_estimate(V, buf)
torch.cuda.synchronize()
