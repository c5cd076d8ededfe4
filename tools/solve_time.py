"""Times the launch-bound Krylov loops on a bench config (single-parameter reduced solve, full-order solve, parabolic
trajectories).  usage: solve_time.py PX PY N [NT]"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
import bench
from pylrbms_amd import multiscale_problem
from pylrbms_amd.discretize_parabolic_block_swipdg import discretize
px, py, N = (int(a) for a in sys.argv[1:4])
nt = int(sys.argv[4]) if len(sys.argv) > 4 else 10
p = multiscale_problem.init_grid_and_problem({'num_subdomains': [px, py], 'coarse_per_subdomain': 4})
d, _ = discretize(p, 0.05, nt)
eng = d.engine
V = eng.ctx.from_numpy(bench.make_bases_host(list(range(eng.S)), eng.t.n, N))
# energy-orthonormalise the columns per subdomain as the bench does (well-conditioned reduced system)
PV = eng.ctx.blockell_apply(eng.P_diag, V)
G = torch.einsum('snk,snl->skl', V, PV)
LinvT = torch.from_numpy(np.linalg.inv(np.linalg.cholesky(G.cpu().numpy())).transpose(0, 2, 1).copy()).to(V.device)
V = torch.bmm(V, LinvT).contiguous()
buf = eng.project_and_estimate(V)
B_sys, rhs_red, E_red, M_red = buf['sys']
theta = d.theta(d.parse_parameter(0.5))
tag = 'launches'


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return out, best


(u, info), t = timed(lambda: eng.reduced_solve(theta, B_sys, rhs_red))
print(tag, 'reduced_solve ms', round(t * 1e3, 3), info, 'us/it', round(t * 1e6 / max(info['iterations'], 1), 2))
(x, info), t = timed(lambda: eng.ctx.fom_solve(theta, eng.A_diag, eng.A_cpl, eng.b), reps=2)
print(tag, 'fom_solve ms', round(t * 1e3, 2), info, 'us/it', round(t * 1e6 / max(info['iterations'], 1), 2))
(U, info), t = timed(lambda: eng.ctx.reduced_implicit_euler(theta, 0.05 / nt, nt, B_sys, M_red, rhs_red), reps=2)
print(tag, 'reduced_implicit_euler ms', round(t * 1e3, 2), info, 'us/it', round(t * 1e6 / max(info['iterations'], 1), 2))
(U, info), t = timed(lambda: eng.ctx.fom_implicit_euler(theta, 0.05 / nt, nt, eng.A_diag, eng.A_cpl, eng.b), reps=1)
print(tag, 'fom_implicit_euler ms', round(t * 1e3, 2), info, 'us/it', round(t * 1e6 / max(info['iterations'], 1), 2))
