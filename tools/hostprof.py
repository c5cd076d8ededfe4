import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from pylrbms_amd import multiscale_problem
from pylrbms_amd.engine import Engine
cfg = bench.CONFIGS['cfg3_tile8']
p = multiscale_problem.init_grid_and_problem({'num_subdomains': cfg['num_subdomains'], 'coarse_per_subdomain': cfg['coarse_per_subdomain']})
lam = p['lambda']
tb = np.array([c.evaluate(p['mu_bar']) for c in lam['coefficients']])
eng = Engine(p['grid'], lam['functions'], p['kappa'], p['f'], p['lambda_bar'], p['lambda_hat'], tb).assemble()
N = cfg['N']
V = eng.ctx.from_numpy(bench.make_bases_host(eng.local, eng.t.n, N))
buf = eng.alloc_reduce_buffers(N)
c = eng.ctx
args = (V, eng.F, eng.A_diag, eng.A_cpl, eng.P_diag, eng.b, eng.ebar, eng.caa, eng.Aab, eng.Bbb, buf['work'], buf['sys'], buf['grams'])
run = c.bind_project_estimate_fused(*args)
main = torch.cuda.current_stream(); side = c.aux_stream(0); ev = torch.cuda.Event()
idx = torch.arange(0, 4096, device=V.device); flat = V.view(-1, N); send = torch.empty(len(idx), N, dtype=V.dtype, device=V.device)
def T(fn, n=300):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    dt=(time.perf_counter()-t0)/n; torch.cuda.synchronize(); return 1e6*dt
print('run(3)', T(lambda: run(3)))
print('run(4)', T(lambda: run(4)))
def p2():
    with torch.cuda.stream(side): run(2)
print('run(2) on side incl ctx mgr', T(p2))
print('stream ctx mgr only', T(lambda: torch.cuda.stream(side).__enter__() or torch.cuda.stream(main).__enter__()))
print('event record', T(lambda: ev.record(main)))
print('wait_event', T(lambda: side.wait_event(ev)))
print('wait_stream', T(lambda: main.wait_stream(side)))
print('index_select', T(lambda: torch.index_select(flat, 0, idx, out=send)))
print('index_copy_', T(lambda: flat.index_copy_(0, idx, send)))
print('current_stream()', T(lambda: torch.cuda.current_stream()))
