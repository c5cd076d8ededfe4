"""Config-3 online phase: batched reduced solves issued by 1, 2 or 3 host threads, each on a library side stream (the C call
synchronises its own stream only and ctypes drops the GIL): do independent batches share the chip?"""
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import bench  # noqa: E402
from pylrbms_amd import multiscale_problem  # noqa: E402
from pylrbms_amd.engine import Engine  # noqa: E402
from pylrbms_amd.parallel import Communicator  # noqa: E402

cfg = bench.CONFIGS['cfg3']
N = cfg['N']
p = multiscale_problem.init_grid_and_problem({'num_subdomains': cfg['num_subdomains'], 'coarse_per_subdomain': cfg['coarse_per_subdomain']},
                                             mpi_comm=Communicator(0, 1))
lam = p['lambda']
theta_bar = np.array([c.evaluate(p['mu_bar']) for c in lam['coefficients']])
eng = Engine(p['grid'], lam['functions'], p['kappa'], p['f'], p['lambda_bar'], p['lambda_hat'], theta_bar)
eng.assemble()
t = p['grid'].template
V = eng.ctx.zeros(eng.S_ext, t.n, N)
V[:eng.S] = eng.ctx.from_numpy(bench.make_bases_host(eng.local, t.n, N))
buf = eng.alloc_reduce_buffers(N)
eng.project_and_estimate(V, buf)
Lh = np.linalg.cholesky(buf['sys'][2].cpu().numpy())
Vo = torch.bmm(V[:eng.S], eng.ctx.from_numpy(np.linalg.inv(Lh).transpose(0, 2, 1))).contiguous()
bufo = eng.project_and_estimate(Vo, buf)
mus = np.random.default_rng(7).uniform(0.1, 1.0, size=256)
coeffs = lam['coefficients']
thetas = np.array([[c.evaluate(float(m)) for c in coeffs] for m in mus])
B, rhs = bufo['sys'][0], bufo['sys'][1]
pc = eng.ctx.reduced_precond_build(np.array([c.evaluate(0.55) for c in coeffs]), B)
eng.ctx.reduced_precond_use(pc)
eng.ctx.reduced_solve_batch(thetas[:16], B, rhs, rtol=1e-12)
torch.cuda.synchronize()
batches = [thetas[b0:b0 + 16] for b0 in range(0, 256, 16)]
ref = None
for nthreads in (1, 2, 3, 1):
    results = [None] * len(batches)

    def worker(k):
        stream = eng.ctx.aux_stream(k) if nthreads > 1 else torch.cuda.current_stream()
        with torch.cuda.stream(stream):
            for b in range(k, len(batches), nthreads):
                results[b] = eng.ctx.reduced_solve_batch(batches[b], B, rhs, rtol=1e-12)[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ths = [threading.Thread(target=worker, args=(k,)) for k in range(nthreads)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    U = torch.cat(results, dim=2)
    if ref is None:
        ref = U.clone()
    print('threads {}: {:8.0f} mu-solves/s   max difference to the single-thread run {:.2e}'.format(
        nthreads, 256 / dt, float((U - ref).abs().max() / ref.abs().max())), flush=True)
