"""Where the time of the batched reduced solve goes (config 3: 32 x 32 subdomains, N = 40, 256 parameters).

    python tools/online_time.py [per_call ...]      # default 16 32 48 64

Per value of ``per_call`` (parameters per lrbms_reduced_solve_batch call): wall time of the sweep over 256 parameters, mu-solves/s,
CG iterations, and the host time spent INSIDE the library calls with the device idle at the end of each (wall - that = what the
device adds).  Under ``rocprofv3 --kernel-trace --stats`` the kernel sum against the wall time tells launch-bound from device-bound."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import CONFIGS, make_bases_host  # noqa: E402
from pylrbms_amd import multiscale_problem  # noqa: E402
from pylrbms_amd.engine import Engine  # noqa: E402

cfg = CONFIGS[os.environ.get('LRBMS_CFG', 'cfg3')]
N = cfg['N']
p = multiscale_problem.init_grid_and_problem({'num_subdomains': cfg['num_subdomains'], 'coarse_per_subdomain': cfg['coarse_per_subdomain']})
lam = p['lambda']
theta_bar = np.array([c.evaluate(p['mu_bar']) for c in lam['coefficients']])
eng = Engine(p['grid'], lam['functions'], p['kappa'], p['f'], p['lambda_bar'], p['lambda_hat'], theta_bar).assemble()
V = eng.ctx.from_numpy(make_bases_host(eng.local, eng.t.n, N))
buf = eng.project_and_estimate(V)
Lh = np.linalg.cholesky(buf['sys'][2].cpu().numpy())
Vo = torch.bmm(V, eng.ctx.from_numpy(np.linalg.inv(Lh).transpose(0, 2, 1))).contiguous()
buf = eng.project_and_estimate(Vo, buf)
mus = np.random.default_rng(7).uniform(0.1, 1.0, size=256)
coeffs = lam['coefficients']
thetas = np.array([[c.evaluate(float(m)) for c in coeffs] for m in mus])
B, rhs = buf['sys'][0], buf['sys'][1]
pc = eng.ctx.reduced_precond_build(np.array([c.evaluate(0.55) for c in coeffs]), B)
eng.ctx.reduced_precond_use(pc)
per_calls = [int(a) for a in sys.argv[1:]] or [16, 32, 48, 64]
for nb in per_calls:
    eng.ctx.reduced_solve_batches(thetas[:nb], B, rhs, per_call=nb, rtol=1e-12)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, info = eng.ctx.reduced_solve_batches(thetas, B, rhs, per_call=nb, rtol=1e-12, concat=False)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('per_call {:2d}: {:7.2f} ms for 256 parameters = {:7.0f} mu-solves/s, {} iterations, rel {:.1e}; host returned after {:.2f} ms'.format(
        nb, 1e3 * dt, 256 / dt, info['iterations'], info['relative_residual'], 1e3 * t_host), flush=True)
eng.ctx.reduced_precond_use(None)
