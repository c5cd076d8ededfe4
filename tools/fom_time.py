"""Times d.solve (FOM PCG) on a bench config.  usage: fom_time.py PX PY"""
import sys, time
sys.path.insert(0, '.')
import torch
from pylrbms_amd import multiscale_problem
from pylrbms_amd.discretize_elliptic_block_swipdg import discretize
px, py = int(sys.argv[1]), int(sys.argv[2])
p = multiscale_problem.init_grid_and_problem({'num_subdomains': [px, py], 'coarse_per_subdomain': 4})
d, _ = discretize(p)
for mu in (1.0, 0.3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    U = d.solve(mu)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('S', px * py, 'dofs', U.dim, 'mu', mu, 'solve s', round(dt, 3), getattr(d, 'last_solve_info', None))
