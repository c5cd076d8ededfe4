"""Times the fused pass whole vs in two phases on a bench config.  usage: phase_time.py CONFIG [name=value ...]   (context options)"""
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
args = (V, eng.F, eng.A_diag, eng.A_cpl, eng.P_diag, eng.b, eng.ebar, eng.caa, eng.Aab, eng.Bbb, buf['work'], buf['sys'], buf['grams'])
def run(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
print('whole   ms', round(run(lambda: eng.ctx.project_estimate_fused(*args)), 4))
print('phased  ms', round(run(lambda: (eng.ctx.project_estimate_fused(*args, phase=1), eng.ctx.project_estimate_fused(*args, phase=2))), 4))
print('phase 1 ms', round(run(lambda: eng.ctx.project_estimate_fused(*args, phase=1)), 4))
print('phase 2 ms', round(run(lambda: eng.ctx.project_estimate_fused(*args, phase=2)), 4))


class _NoHalo:
    def start(self, V):
        return lambda: V


print('overlap ms', round(run(lambda: eng.project_and_estimate(V, buf, halo=_NoHalo())), 4), '(stream choreography of a sharded pass, no exchange)')

torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    eng.ctx.project_estimate_fused(*args, phase=3)
host = (time.perf_counter() - t0) / 200
torch.cuda.synchronize()
print('host time of one library call through NativeContext (phase 3, enqueue only): %.1f us' % (1e6 * host))

# host side of one sharded step (pack + phases 3 / 4 / 2 + stream choreography, exchange replaced by two torch index ops of
# the same size): enqueue-only time per step.  If it exceeds the device time per step the multi-GPU run is launch-bound.
idx = torch.arange(0, min(eng.S * eng.t.n, 4096), device=V.device)
flat = V.view(-1, N)
send = torch.empty(len(idx), N, dtype=V.dtype, device=V.device)


class _PackOnly:
    def start(self, V):
        torch.index_select(flat, 0, idx, out=send)
        return lambda: flat.index_copy_(0, idx, send)


h = _PackOnly()
for _ in range(5):
    eng.project_and_estimate(V, buf, halo=h)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    eng.project_and_estimate(V, buf, halo=h)
host = (time.perf_counter() - t0) / 200
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / 200
print('sharded step: host enqueue %.1f us per step, device-paced total %.1f us per step' % (1e6 * host, 1e6 * total))
