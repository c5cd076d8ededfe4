"""Online phase at benchmark sizes: reduced solve (O1) + reduced estimate (E1) for a batch of parameters."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pylrbms_amd import multiscale_problem  # noqa: E402
from pylrbms_amd.discretize_elliptic_block_swipdg import discretize  # noqa: E402
from pylrbms_amd.reductor import LRBMSReductor  # noqa: E402
from pylrbms_amd.vectorarrays import BlockVectorArray  # noqa: E402

P, N = (int(sys.argv[1]), int(sys.argv[1])), int(sys.argv[2])
p = multiscale_problem.init_grid_and_problem({'num_subdomains': list(P), 'coarse_per_subdomain': 4})
d, data = discretize(p)
eng = d.engine
red = LRBMSReductor(d, products=None, order=0)
rng = np.random.default_rng(0)
U = BlockVectorArray(eng.ctx.from_numpy(rng.standard_normal((eng.S, eng.t.n, N - 1))), d.solution_space)
t0 = time.time()
red.extend_basis(U)
torch.cuda.synchronize()
print('gram-schmidt of {} vectors: {:.2f}s, N = {}'.format(N - 1, time.time() - t0, red.basis_size()))
rd = red.reduce()
torch.cuda.synchronize()
mus = np.random.default_rng(7).uniform(0.1, 1.0, size=8)
for mu in mus[:2]:
    u = rd.solve(float(mu))
    print('mu {:.3f}: cg iterations {}, rel residual {:.2e}'.format(mu, rd.last_solve_info['iterations'], rd.last_solve_info['relative_residual']))
torch.cuda.synchronize()
t0 = time.time()
for mu in mus:
    u = rd.solve(float(mu))
torch.cuda.synchronize()
ts = (time.time() - t0) / len(mus)
t0 = time.time()
for mu in mus:
    eta = rd.estimate(u, mu=float(mu))
torch.cuda.synchronize()
te = (time.time() - t0) / len(mus)
print('S = {}, N = {}: solve {:.2f} ms/mu ({:.1f} mu-solves/s), estimate {:.2f} ms/mu, eta = {:.4e}'.format(eng.S, N, 1e3 * ts, 1 / ts, 1e3 * te, eta))
# batched solves
for nmu in (4, 8, 16):
    if N * nmu > 768:
        break
    thetas = np.stack([d.theta(float(m)) for m in np.random.default_rng(7).uniform(0.1, 1.0, size=nmu)])
    ub, info = eng.ctx.reduced_solve_batch(thetas, rd.B_sys, rd.rhs_red)
    torch.cuda.synchronize()
    t0 = time.time()
    reps = 3
    for _ in range(reps):
        ub, info = eng.ctx.reduced_solve_batch(thetas, rd.B_sys, rd.rhs_red)
    torch.cuda.synchronize()
    tb = (time.time() - t0) / reps
    print('batch of {:2d}: {} iterations, rel {:.1e}, {:.2f} ms per batch = {:.0f} mu-solves/s'.format(
        nmu, info['iterations'], info['relative_residual'], 1e3 * tb, nmu / tb))
