"""Config 5 (3D): whole pass vs the two halves of a sharded pass (phase 1: rank-local slabs only, phase 2: neighbour rows), no
exchange, on one GPU.  usage: python tools/phase_time3d.py [cfg5|cfg5_tile8]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
cfg_name = sys.argv[1] if len(sys.argv) > 1 else 'cfg5_tile8'
sys.argv = [sys.argv[0]]
import torch  # noqa: E402
import bench3d  # noqa: E402
from pylrbms_amd import multiscale_problem3d  # noqa: E402
from pylrbms_amd.engine3d import Engine3D  # noqa: E402

cfg = bench3d.CONFIGS3D[cfg_name]
p = multiscale_problem3d.init_grid_and_problem({'num_subdomains': cfg['num_subdomains'], 'cubes_per_subdomain': cfg['cubes_per_subdomain']})
lam = p['lambda']
eng = Engine3D(p['grid'], lam['functions'], p['f'], p['lambda_bar'], p['lambda_hat']).assemble()
N = cfg['N']
rng = np.random.default_rng(0)
V = eng.ctx.from_numpy(np.concatenate([np.ones((eng.S_ext, eng.t.n, 1)), rng.standard_normal((eng.S_ext, eng.t.n, N - 1))], axis=2))
out, work = eng.alloc_outputs(N), eng.alloc_work(N)


def timed(fn, steps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


whole = timed(lambda: eng.ctx.project_estimate(eng.Q, V, eng.ops, work, out))
p1 = timed(lambda: eng.ctx.project_estimate(eng.Q, V, eng.ops, work, out, phase=1))
p2 = timed(lambda: eng.ctx.project_estimate(eng.Q, V, eng.ops, work, out, phase=2))


def both():
    eng.ctx.project_estimate(eng.Q, V, eng.ops, work, out, phase=1)
    eng.ctx.project_estimate(eng.Q, V, eng.ops, work, out, phase=2)


print('{}: whole {:.4f} ms | phase 1 {:.4f} | phase 2 {:.4f} | 1 then 2 {:.4f}'.format(cfg_name, whole, p1, p2, timed(both)))
