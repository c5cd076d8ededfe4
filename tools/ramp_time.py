import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from pylrbms_amd import multiscale_problem
from pylrbms_amd.engine import Engine
cfg = bench.CONFIGS['cfg3']
p = multiscale_problem.init_grid_and_problem({'num_subdomains': cfg['num_subdomains'], 'coarse_per_subdomain': cfg['coarse_per_subdomain']})
lam = p['lambda']
tb = np.array([c.evaluate(p['mu_bar']) for c in lam['coefficients']])
eng = Engine(p['grid'], lam['functions'], p['kappa'], p['f'], p['lambda_bar'], p['lambda_hat'], tb).assemble()
N = cfg['N']
V = eng.ctx.from_numpy(bench.make_bases_host(eng.local, eng.t.n, N))
buf = eng.alloc_reduce_buffers(N)
for rep in range(8):
    if rep == 4:
        time.sleep(3.0)      # idle GPU
    for _ in range(3):
        eng.project_and_estimate(V, buf)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        eng.project_and_estimate(V, buf)
    torch.cuda.synchronize()
    print(rep, '20 steps: %.4f ms/step' % (1e3 * (time.perf_counter() - t0) / 20), flush=True)
for n in (20, 50, 100, 200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        eng.project_and_estimate(V, buf)
    torch.cuda.synchronize()
    print(n, 'steps: %.4f ms/step' % (1e3 * (time.perf_counter() - t0) / n), flush=True)

# per-step device times of the first 40 steps after the GPU has been idle for 3 s
time.sleep(3.0)
evs = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
evs[0].record()
for i in range(40):
    eng.project_and_estimate(V, buf)
    evs[i + 1].record()
torch.cuda.synchronize()
print('per-step ms after idle:', ' '.join('%.3f' % evs[i].elapsed_time(evs[i + 1]) for i in range(40)))
