"""Times LRBMSReductor.reduce() (the reference's timed region, reductor.py:33-73) through the API at config 3."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from pylrbms_amd import multiscale_problem
from pylrbms_amd.discretize_elliptic_block_swipdg import discretize
from pylrbms_amd.reductor import LRBMSReductor
p = multiscale_problem.init_grid_and_problem({'num_subdomains': [32, 32], 'coarse_per_subdomain': 4})
d, _ = discretize(p)
Vh = bench.make_bases_host(d.engine.local, d.engine.t.n, 40)
red = LRBMSReductor(d, bases={'domain_{}'.format(ii): Vh[i].T for i, ii in enumerate(d.engine.local)})
for k in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rd = red.reduce()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('reduce() call', k, 'ms', round(1e3 * dt, 3))
