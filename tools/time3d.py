"""Per-kernel timing of the 3D / P2 pass (config 5 by default): python tools/time3d.py [P] [kc] [N] [steps] [degree] [opt=value ...]
(launch-policy options of the 3D context, e.g. waves=8 ksplit=1)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from pylrbms_amd import multiscale_problem3d  # noqa: E402
from pylrbms_amd.engine3d import Engine3D  # noqa: E402

OPTS = [a for a in sys.argv[1:] if '=' in a]
sys.argv = [a for a in sys.argv if '=' not in a]
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
kc = int(sys.argv[2]) if len(sys.argv) > 2 else 4
N = int(sys.argv[3]) if len(sys.argv) > 3 else 30
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
deg = int(sys.argv[5]) if len(sys.argv) > 5 else 2
t0 = time.time()
p = multiscale_problem3d.init_grid_and_problem({'num_subdomains': (P, P, P), 'cubes_per_subdomain': kc, 'data_degree': deg})
lam = p['lambda']
eng = Engine3D(p['grid'], lam['functions'], p['f'], p['lambda_bar'], p['lambda_hat'], data_degree=deg)
for o in OPTS:
    eng.ctx.set_option(o.split('=')[0], int(o.split('=')[1]))
t1 = time.time()
eng.assemble()
torch.cuda.synchronize()
t2 = time.time()
print('setup {:.1f}s  assemble {:.3f}s  S={} n={} n_rt={} Q={}'.format(t1 - t0, t2 - t1, eng.S, eng.t.n, eng.t.n_rt, eng.Q), flush=True)
g = torch.Generator(device='cuda').manual_seed(0)
V = torch.randn(eng.S_ext, eng.t.n, N, dtype=torch.float64, device='cuda', generator=g)
V[:, :, 0] = 1.0
out, work = eng.alloc_outputs(N), eng.alloc_work(N)
for _ in range(2):
    eng.project_and_estimate(V, out, work)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    eng.project_and_estimate(V, out, work)
torch.cuda.synchronize()
ms = (time.time() - t0) / steps * 1e3
print('pass {:.3f} ms  -> {:.0f} subdomains/s'.format(ms, eng.S / ms * 1e3), flush=True)
eng.ctx.kernel_timing(True)
for _ in range(steps):
    eng.project_and_estimate(V, out, work)
rows = eng.ctx.kernel_timing_read()
eng.ctx.kernel_timing(False)
agg = {}
for name, v in rows:
    agg.setdefault(name, []).append(v)
tot = 0.0
for name, v in agg.items():
    print('  {:16s} {:9.1f} us'.format(name, 1e3 * np.mean(v)))
    tot += 1e3 * np.mean(v)
print('  sum {:.1f} us'.format(tot))
th = np.array([1.0, 0.5])
u = torch.randn(eng.S_ext, N, dtype=torch.float64, device='cuda', generator=g)
for _ in range(2):
    eta = eng.reduced_estimate(th, u, out)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    eta = eng.reduced_estimate(th, u, out)
torch.cuda.synchronize()
print('estimate {:.3f} ms'.format((time.time() - t0) / steps * 1e3))
t0 = time.time()
us, info = eng.reduced_solve(th, out, rtol=1e-10, max_iter=20000)
torch.cuda.synchronize()
print('reduced solve {:.3f} ms, iterations {}, residual {:.2e}'.format((time.time() - t0) * 1e3, *info))
