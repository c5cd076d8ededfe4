"""Times the single-parameter and the batched reduced estimate at config 3."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from pylrbms_amd import multiscale_problem
from pylrbms_amd.engine import Engine
cfg = bench.CONFIGS['cfg3']
p = multiscale_problem.init_grid_and_problem({'num_subdomains': cfg['num_subdomains'], 'coarse_per_subdomain': 4})
lam = p['lambda']
tb = np.array([c.evaluate(p['mu_bar']) for c in lam['coefficients']])
eng = Engine(p['grid'], lam['functions'], p['kappa'], p['f'], p['lambda_bar'], p['lambda_hat'], tb).assemble()
N = 40
V = eng.ctx.from_numpy(bench.make_bases_host(eng.local, eng.t.n, N))
buf = eng.project_and_estimate(V)
u = eng.ctx.from_numpy(np.random.default_rng(0).standard_normal((eng.S, N)))
th = np.array([1.0, 0.4])
def run(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
print('single estimate ms', round(run(lambda: eng.reduced_estimate(th, u, buf['grams'])), 3))
u1 = u[:, :, None].contiguous()
print('batched (nmu=1) ms', round(run(lambda: eng.ctx.reduced_estimate_batch(th[None, :], u1, buf['grams'], eng.f2, eng.ceps, eng.hdiam)), 3))
u16 = u[:, :, None].repeat(1, 1, 16).contiguous()
print('batched (nmu=16) ms', round(run(lambda: eng.ctx.reduced_estimate_batch(np.tile(th, (16, 1)), u16, buf['grams'], eng.f2, eng.ceps, eng.hdiam)), 3))
a = eng.reduced_estimate(th, u, buf['grams']).cpu().numpy()
b = eng.ctx.reduced_estimate_batch(th[None, :], u1, buf['grams'], eng.f2, eng.ceps, eng.hdiam).cpu().numpy()[:, :, 0]
print('max rel diff single vs batched', np.abs(a - b).max() / np.abs(a).max())
