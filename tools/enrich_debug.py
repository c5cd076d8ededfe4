"""Debug helper: hood solves vs oracle on one config; prints per-subdomain info."""
import sys
import numpy as np
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from common import oracle_from_problem
from pylrbms_amd import multiscale_problem
from pylrbms_amd.discretize_elliptic_block_swipdg import discretize

cfg = {'num_subdomains': [int(sys.argv[1]), int(sys.argv[2])], 'coarse_per_subdomain': int(sys.argv[3])}
mu = float(sys.argv[4]) if len(sys.argv) > 4 else 0.6
p = multiscale_problem.init_grid_and_problem(cfg)
d, _ = discretize(p)
o = oracle_from_problem(p)
eng = d.engine
rng = np.random.default_rng(1)
x = rng.standard_normal((o.S, o.n, 1))
y = eng.ctx.fom_apply(d.theta(mu), eng.A_diag, eng.A_cpl, eng.ctx.from_numpy(x)).cpu().numpy()
ref = (o.assemble_global(mu) @ x.reshape(o.ndof, 1)).reshape(o.S, o.n, 1)
print('fom apply err', np.abs(y - ref).max() / np.abs(ref).max())
print('nbr', eng.nbr.tolist(), 'nT', eng.t.n_T, 'ncf', eng.t.ncf)
try:
    corr, info = eng.local_corrections(d.theta(mu), list(range(o.S)), max_iter=5000)
except Exception as e:
    print('ERR', e)
    import ctypes
    th = np.ascontiguousarray(d.theta(mu))
    corr = None
for it in (1, 5, 50, 500, 5000):
    try:
        corr, info = eng.local_corrections(d.theta(mu), list(range(o.S)), rtol=1e-12, max_iter=it)
        print(it, info.tolist())
    except Exception as e:
        print(it, 'ERR', e)
