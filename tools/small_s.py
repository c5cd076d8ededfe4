"""Host enqueue time vs. device time of one fused pass at a small per-rank subdomain count.
usage: python tools/small_s.py CONFIG [streams=0|1] [f1_ksplit=1|2|4]   (launch-policy options of the context)"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from pylrbms_amd import multiscale_problem
from pylrbms_amd.engine import Engine
name = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
cfg = bench.CONFIGS[name]
p = multiscale_problem.init_grid_and_problem({'num_subdomains': cfg['num_subdomains'], 'coarse_per_subdomain': cfg['coarse_per_subdomain']})
lam = p['lambda']
tb = np.array([c.evaluate(p['mu_bar']) for c in lam['coefficients']])
eng = Engine(p['grid'], lam['functions'], p['kappa'], p['f'], p['lambda_bar'], p['lambda_hat'], tb).assemble()
for o in sys.argv[2:]:
    eng.ctx.set_option(o.split('=')[0], int(o.split('=')[1]))
N = cfg['N']
V = eng.ctx.from_numpy(bench.make_bases_host(eng.local, eng.t.n, N))
buf = eng.alloc_reduce_buffers(N)
args = (V, eng.F, eng.A_diag, eng.A_cpl, eng.P_diag, eng.b, eng.ebar, eng.caa, eng.Aab, eng.Bbb, buf['work'], buf['sys'], buf['grams'])
run = eng.ctx.bind_project_estimate_fused(*args)
for _ in range(20): run(0)
torch.cuda.synchronize()
n = 300
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for _ in range(n): run(0)
t_host = time.perf_counter() - t0
e1.record(); torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print('{}: S={} host enqueue {:.1f} us/pass, wall {:.1f} us/pass, device events {:.1f} us/pass'.format(
    name, eng.S, 1e6 * t_host / n, 1e6 * t_all / n, 1e3 * e0.elapsed_time(e1) / n))
# one pass alone (latency, nothing queued behind it)
lat = []
for _ in range(30):
    torch.cuda.synchronize(); e0.record(); run(0); e1.record(); torch.cuda.synchronize(); lat.append(1e3 * e0.elapsed_time(e1))
print('single pass latency (events): median {:.1f} us, min {:.1f} us'.format(sorted(lat)[len(lat) // 2], min(lat)))
# enqueue cost without back-pressure: short bursts behind an idle queue
burst = []
for _ in range(20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): run(0)
    burst.append(1e6 * (time.perf_counter() - t0) / 10)
    torch.cuda.synchronize()
print('enqueue of a 10-pass burst behind an idle queue: median {:.1f} us/pass, min {:.1f}'.format(sorted(burst)[10], min(burst)))
