"""Pass time and per-kernel device times (HIP event pairs on each kernel's own stream, lrbms_kernel_timing) of the fused pass on a bench
config.  usage: kernel_times.py CONFIG [name=value ...]   (context options, e.g. f1_ksplit=1 streams=0)"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from pylrbms_amd import multiscale_problem
from pylrbms_amd.engine import Engine
cfg = bench.CONFIGS[sys.argv[1]]
p = multiscale_problem.init_grid_and_problem({'num_subdomains': cfg['num_subdomains'], 'coarse_per_subdomain': cfg['coarse_per_subdomain']})
lam = p['lambda']
tb = np.array([c.evaluate(p['mu_bar']) for c in lam['coefficients']])
eng = Engine(p['grid'], lam['functions'], p['kappa'], p['f'], p['lambda_bar'], p['lambda_hat'], tb).assemble()
for o in sys.argv[2:]:
    k, v = o.split('=')
    eng.ctx.set_option(k, int(v))
N = cfg['N']
V = eng.ctx.from_numpy(bench.make_bases_host(eng.local, eng.t.n, N))
buf = eng.alloc_reduce_buffers(N)
for _ in range(5):
    eng.project_and_estimate(V, buf)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    eng.project_and_estimate(V, buf)
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / 200
eng.ctx.kernel_timing(True)
for _ in range(20):
    eng.project_and_estimate(V, buf)
rows = eng.ctx.kernel_timing_read()
eng.ctx.kernel_timing(False)
acc = {}
for k, v in rows:
    acc.setdefault(k, []).append(v)
print('{} {}: pass {:.4f} ms; kernels (us): {}'.format(sys.argv[1], ' '.join(sys.argv[2:]), ms, {k: round(1e3 * float(np.mean(v)), 1) for k, v in acc.items()}))
