"""Times the parabolic path on a bench config: full-order implicit Euler, reduced implicit Euler, both estimates.
usage: parabolic_time.py PX PY N NT"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
from pylrbms_amd import multiscale_problem
from pylrbms_amd.discretize_parabolic_block_swipdg import discretize
from pylrbms_amd.reductor import ParabolicLRBMSReductor
px, py, N, nt = (int(a) for a in sys.argv[1:5])
p = multiscale_problem.init_grid_and_problem({'num_subdomains': [px, py], 'coarse_per_subdomain': 4})
d, _ = discretize(p, 0.1, nt)
mu = d.parse_parameter(0.5)


def timed(fn, reps=2):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return out, best


U, t = timed(lambda: d.solve(mu))
print('S', px * py, 'dofs', U.dim, 'nt', nt, 'FOM implicit Euler s', round(t, 4), 'steps/s', round(nt / t, 1), d.last_solve_info)
(est, parts), t = timed(lambda: d.estimate(U, mu), reps=1)
print('FOM parabolic estimate s', round(t, 4), 'est', est)
reductor = ParabolicLRBMSReductor(d, order=0)
rng = np.random.default_rng(0)
idx = sorted(rng.choice(np.arange(1, nt + 1), size=min(N - 1, nt), replace=False).tolist())
reductor.extend_basis(U[idx])
while reductor.basis_size() < N:          # fill up with random vectors so that the reduced model has the bench size
    R = d.solution_space.from_data(rng.standard_normal((1, U.dim)), d.engine.ctx)
    reductor.extend_basis(R)
rd, t = timed(lambda: reductor.reduce())
print('N', reductor.basis_size(), 'reduce s', round(t, 5))
u, t = timed(lambda: rd.solve(mu))
print('reduced implicit Euler s', round(t, 4), 'steps/s', round(nt / t, 1), rd.last_solve_info)
(est_r, parts_r), t = timed(lambda: rd.estimate(u, mu))
print('reduced parabolic estimate s', round(t, 4), 'est', est_r)
