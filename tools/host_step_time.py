"""Host time of the pieces of one sharded step, each enqueued onto an IDLE device (synchronise, time the enqueue, repeat): what the
host pays per step whatever the device does.  usage: host_step_time.py CONFIG [name=value ...]   (context options)"""
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
run = eng.ctx.bind_project_estimate_fused(*args)
run(0)
idx = torch.arange(0, min(eng.S * eng.t.n, 4096), device=V.device)
flat = V.view(-1, N)
send = torch.empty(len(idx), N, dtype=V.dtype, device=V.device)
main = torch.cuda.current_stream()
side = eng.ctx.aux_stream(0)
ev = torch.cuda.Event()


def timed(name, fn, reps=300):
    for _ in range(10):
        fn()
    tot = 0.0
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        tot += time.perf_counter() - t0
    torch.cuda.synchronize()
    print('{:44s} {:6.1f} us'.format(name, 1e6 * tot / reps), flush=True)


def switch():
    torch.cuda.set_stream(side)
    torch.cuda.set_stream(main)


def rec_wait():
    ev.record(side)
    main.wait_event(ev)


timed('ctypes call that does nothing (lrbms_version)', lambda: eng.ctx.lib.lrbms_version())
timed('torch.cuda.current_stream().cuda_stream', lambda: torch.cuda.current_stream(V.device).cuda_stream)
timed('library call, phase 3 (preparation)', lambda: run(3))
timed('library call, phase 4 (dense kernels)', lambda: run(4))
timed('library call, phase 1 (= 3 + 4)', lambda: run(1))
timed('library call, phase 2', lambda: run(2))
timed('library call, phase 0 (whole pass)', lambda: run(0))
timed('library call, phase 5 (1 + 2 in one call)', lambda: run(5))
timed('library call, phase 5, stream handle passed', lambda: run(5, main.cuda_stream))
timed('torch.index_select(out=) (pack)', lambda: torch.index_select(flat, 0, idx, out=send))
timed('index_copy_ (unpack)', lambda: flat.index_copy_(0, idx, send))
timed('set_stream pair', switch)
timed('event record + wait_event', rec_wait)
