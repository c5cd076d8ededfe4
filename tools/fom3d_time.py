"""Full-order solve of the 3D / P2 path (config 5 by default) with and without the coarse level: python tools/fom3d_time.py [P] [kc]."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from pylrbms_amd import multiscale_problem3d  # noqa: E402
from pylrbms_amd.engine3d import Engine3D  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
kc = int(sys.argv[2]) if len(sys.argv) > 2 else 4
p = multiscale_problem3d.init_grid_and_problem({'num_subdomains': (P, P, P), 'cubes_per_subdomain': kc})
eng = Engine3D(p['grid'], p['lambda']['functions'], p['f'], p['lambda_bar'], p['lambda_hat']).assemble()
c = eng.ctx
t = eng.t
x = np.asarray(t.node_coordinates())
ext = x.max(axis=0) - x.min(axis=0)
lin = (x - 0.5 * (x.max(axis=0) + x.min(axis=0))) / ext
spaces = {'none': None, 'constants': np.ones((t.n, 1)), 'P1': np.concatenate([np.ones((t.n, 1)), lin], axis=1)}
ref = None
for mu in (0.5, 0.1, 1.0):
    th = np.array([1.0, mu])
    for name, Phi in spaces.items():
        c.fom_coarse_space(Phi)
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            U, info = c.fom_solve(eng.Q, th, eng.ops['A_diag'], eng.ops['A_cpl'], eng.ops['b'], rtol=1e-8)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3
        if name == 'none':
            ref = U.clone()
        err = float((U - ref).norm() / ref.norm())
        print('mu {:4.2f}  coarse {:10s} {:7.1f} ms  iterations {:5d}  residual {:.2e}  vs none {:.2e}'.format(mu, name, ms, int(info[0]), info[1], err), flush=True)
# the coarse inverse of the first solve kept for the others (lrbms3_fom_precond_keep)
c.fom_coarse_space(spaces['P1'])
c.fom_precond_keep(True)
for mu in (0.55, 0.1, 0.5, 1.0):
    th = np.array([1.0, mu])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    U, info = c.fom_solve(eng.Q, th, eng.ops['A_diag'], eng.ops['A_cpl'], eng.ops['b'], rtol=1e-8)
    torch.cuda.synchronize()
    print('mu {:4.2f}  coarse P1 kept (built at 0.55)  {:7.1f} ms  iterations {:5d}  residual {:.2e}'.format(
        mu, (time.perf_counter() - t0) * 1e3, int(info[0]), info[1]), flush=True)
c.fom_precond_keep(False)
