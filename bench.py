#!/usr/bin/env python
"""Benchmark of the LRBMS offline hot path (project + estimate-offline: K7 + K8 + P1 + P2 of SURVEY.md section 8).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg3|cfg2] [--no-cpu-baseline]

One "step" = one pass of the hot path over all subdomains of the synthetic multiscale-diffusion problem
(BASELINE.json config 3: 32x32 subdomains, local basis dim 40, fp64): Oswald image bases, RT0 flux-reconstruction
image bases, Galerkin projection of the block SWIPDG system and of every estimator operator.  Inputs (assembled
operators, bases) are resident in HBM before the timed region.  For N > 1 the subdomains are tiled over the ranks
(strong scaling, as BASELINE.json config 4) and every step starts with the halo all-gather of neighbour basis rows.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    'cfg2': {'num_subdomains': [8, 8], 'N': 20, 'coarse_per_subdomain': 4},
    'cfg3': {'num_subdomains': [32, 32], 'N': 40, 'coarse_per_subdomain': 4},
    # diagnostics: what ONE rank of the 8-GPU run of config 3 holds (16x8 tile), without the halo exchange
    'cfg3_tile8': {'num_subdomains': [16, 8], 'N': 40, 'coarse_per_subdomain': 4},    # per-rank tiles of the 8/4/2-GPU runs
    'cfg3_tile4': {'num_subdomains': [16, 16], 'N': 40, 'coarse_per_subdomain': 4},
    'cfg3_tile2': {'num_subdomains': [32, 16], 'N': 40, 'coarse_per_subdomain': 4},
    # SURVEY.md section 8d sweep k_c in {4, 8, 16}: the SAME global mesh as config 3 cut into fewer, larger subdomains
    'cfg3_kc8': {'num_subdomains': [16, 16], 'N': 40, 'coarse_per_subdomain': 8},
    'cfg3_kc16': {'num_subdomains': [8, 8], 'N': 40, 'coarse_per_subdomain': 16},
}

# peaks from /opt/skills/guides/MI355X_MICROARCH.md (HBM3E spec) and SURVEY.md section 8d (fp64 matrix)
PEAK_HBM_GBS = 8000.0
PEAK_FP64_MFMA_TFLOPS = 78.6


def algorithmic_flops_per_subdomain(n, n_rt, n_T, N, Q, m=5, n_c=24):
    """Canonical count of SURVEY.md section 8(d) (full GEMM count, interior subdomain)."""
    nnz_A, nnz_c, nnz_E, nnz_M, nnz_aa = 12 * n, 72, 3 * n, 3 * n, 3 * n
    nnz_Div = nnz_ab = 9 * n_T
    nnz_B = 5 * n_rt
    C = Q * m * N
    F_P1 = Q * (2 * nnz_A * N + 2 * n * N * N) + (m - 1) * Q * (2 * nnz_c * N + 2 * n_c * N * N) + 2 * n * N + \
        (2 * nnz_M * N + 2 * n * N * N)
    F_nc = 2 * nnz_E * m * N + 2 * n * (m * N) ** 2
    F_r = 2 * nnz_Div * C + 2 * n * C + 2 * nnz_M * C + 2 * n * C * C
    F_bb = 2 * nnz_B * C + 2 * n_rt * C * C
    F_ab = Q * (2 * nnz_ab * C + 2 * n * N * C)
    F_aa = Q * Q * (2 * nnz_aa * N + 2 * n * N * N)
    return F_P1 + F_nc + F_r + F_bb + F_ab + F_aa


def algorithmic_bytes_per_subdomain(n, n_rt, n_T, N, Q, m=5, n_c=24):
    """inputs once, outputs once, intermediates W, R, D written + read."""
    C = Q * m * N
    nnz_A = 12 * n
    inputs = 8 * n * N + 8 * (m - 1) * n_c * N + 12 * (Q * nnz_A + (m - 1) * Q * 72 + 3 * n * (2 + Q * Q) + 9 * n_T * (1 + Q) + 5 * n_rt)
    inter = 2 * (8 * m * n * N + 8 * Q * m * n_rt * N + 8 * n * C)
    outputs = 8 * (Q * m * N * N + (m * N) ** 2 + 2 * C * C + Q * N * C + Q * Q * N * N + C)
    return inputs + inter + outputs


def make_bases_host(subdomains, n, N, seed=0):
    V = np.empty((len(subdomains), n, N))
    for i, s in enumerate(subdomains):
        rng = np.random.default_rng(seed + int(s))
        V[i, :, 0] = 1.0
        V[i, :, 1:] = rng.standard_normal((n, N - 1))
    return V


_POOL_STATE = {}


def _pool_reduce(chunk):
    from oracle.lrbms import OracleReductor
    d, V = _POOL_STATE['d'], _POOL_STATE['V']
    OracleReductor(d, [V[ii] for ii in range(d.S)]).reduce(subdomains=chunk)
    return len(chunk)


def host_cores():
    """Cores this process may actually use: scheduler affinity, capped by the cgroup CPU quota when there is one."""
    info = {'os_cpu_count': os.cpu_count()}
    try:
        info['affinity'] = len(os.sched_getaffinity(0))
    except AttributeError:
        info['affinity'] = os.cpu_count() or 1
    quota = None
    try:
        with open('/sys/fs/cgroup/cpu.max') as fh:
            a, b = fh.read().split()[:2]
            if a != 'max':
                quota = max(1, int(float(a) / float(b)))
    except (OSError, ValueError):
        pass
    info['cgroup_quota'] = quota
    info['cores'] = max(1, min(info['affinity'], quota) if quota else info['affinity'])
    return info


def cpu_baseline(num_subdomains, N, coarse):
    """The oracle (kind "port") timed on the SAME workload as the GPU (full subdomain grid, same basis size): only
    OracleReductor.reduce() -- the region the GPU times -- is timed.  Runs BEFORE the process touches the GPU, so the
    fork()ed worker pool never inherits HIP state.  All host cores the process may use."""
    from threadpoolctl import threadpool_limits
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from common import oracle_from_problem
    from oracle.lrbms import OracleReductor
    from pylrbms_amd import multiscale_problem
    p = multiscale_problem.init_grid_and_problem({'num_subdomains': list(num_subdomains), 'coarse_per_subdomain': coarse})
    t_asm = time.perf_counter()
    d = oracle_from_problem(p)
    t_asm = time.perf_counter() - t_asm
    d.precompute_blocks()
    V = make_bases_host(range(d.S), d.n, N)
    import multiprocessing as mp
    hc = host_cores()
    cores = hc['cores']
    with threadpool_limits(limits=1):
        # (i) one core, the way the reference runs (single-threaded per rank, reductor.py:19 ignores num_cpus)
        t0 = time.perf_counter()
        OracleReductor(d, [V[ii] for ii in range(d.S)]).reduce()
        one = d.S / (time.perf_counter() - t0)
        # (ii) all host cores: target subdomains farmed over a fork()ed process pool
        _POOL_STATE['d'], _POOL_STATE['V'] = d, V
        chunks = [list(c) for c in np.array_split(np.arange(d.S), 4 * cores) if len(c)]
        with mp.get_context('fork').Pool(cores) as pool:
            pool.map(_pool_reduce, chunks[:cores])                      # start-up of the workers is not timed
            t0 = time.perf_counter()
            pool.map(_pool_reduce, chunks)
            allc = d.S / (time.perf_counter() - t0)
        # online enrichment: the oracle's neighbourhood corrector (sparse direct solve) on a few interior subdomains
        picks = [d.S // 2 + k for k in range(4)]
        t0 = time.perf_counter()
        for ii in picks:
            d.solve_for_local_correction(ii, 0.3)
        corr = len(picks) / (time.perf_counter() - t0)
        # parabolic path: the oracle's implicit Euler (one sparse LU, then a solve per step)
        from oracle.parabolic import OracleParabolic
        t0 = time.perf_counter()
        OracleParabolic(d, 0.05, 10).solve(0.5)
        par = 10 / (time.perf_counter() - t0)
    _POOL_STATE.clear()
    out = {'value': allc, 'unit': 'subdomains/s', 'cores': cores, 'kind': 'port', 'value_1core': one,
           'assemble_subdomains_per_s_1core': d.S / t_asm, 'local_correction_solves_per_s_1core': corr,
           'implicit_euler_steps_per_s_1core': par,
           'sample': 'oracle.lrbms.OracleReductor.reduce() (NumPy/SciPy fp64) on the full {}x{} subdomains of the same '
                     'synthetic multiscale problem, N={}, k_c={}: value = target subdomains farmed over a {}-process '
                     'pool (1 BLAS thread each), value_1core = one process, one thread; timed before the GPU is touched'
                     .format(num_subdomains[0], num_subdomains[1], N, coarse, cores)}
    out.update({'os_cpu_count': hc['os_cpu_count'], 'affinity': hc['affinity'], 'cgroup_quota': hc['cgroup_quota']})
    return out


def kernel_model(t, N, Q, S):
    """Compulsory HBM bytes (each input once, each output once) and executed fp64-MFMA flops (tile padding included)
    of every kernel of the fused pass for S interior subdomains: {kernel: (bytes_read, bytes_written, mfma_flops)}.
    Shared inputs are charged to every kernel that reads them (each launch must stream them once)."""
    n, nrt, nT, nv, ncf = t.n, t.n_rt, t.n_T, t.n_vertices, t.ncf
    QN = Q * N
    nvs = max(2 * t.kx + 1, 2 * t.ky + 1)
    ntx, nr = (N + 15) // 16, (QN + 15) // 16
    chunks = nT // 4
    d8 = 8
    V_self = d8 * n * N
    V_halo = d8 * 4 * (3 * t.ntouch) * N                    # rows of the neighbours' elements touching the shared side
    R_self, R_side = d8 * nrt * QN, d8 * 4 * ncf * QN
    A_self, A_side = d8 * nv * N, d8 * 4 * nvs * N
    m = {
        'k_flux_compact': (V_self + d8 * 4 * 3 * ncf * N + d8 * Q * nrt * 6, R_self + R_side, 0),
        'k_vertex_avg': (V_self + V_halo, A_self + A_side, 0),
        # both sweeps from one copy of the slab in LDS, and G_nc[self, self] (rank-2 form: 2 K rows per element) from the same copy
        'k_prep_lds': (V_self + V_halo + d8 * 4 * 3 * ncf * N + d8 * Q * nrt * 6 + d8 * nT, R_self + R_side + A_self + A_side + d8 * N * N,
                       (nT // 2) * (ntx * (ntx + 1) // 2) * 2048),
        'k_f1': (V_self + d8 * nT * (36 * Q + 36 + Q * Q + 9 * Q) + R_self + d8 * n,
                 d8 * (Q * N * N + 2 * N * N + Q * Q * N * N + Q * N * QN + N),
                 (chunks * 3 * ntx * 28 + nT * 3 * ntx) * 2048),
        'k_f2': (R_self + d8 * nT * 9 + d8 * n, d8 * (2 * QN * QN + QN), chunks * 4 * (nr * (nr + 1) // 2) * 2048),
        'k_f3': (V_self + A_self + d8 * nT, d8 * N * N, (nT // 16) * 12 * (ntx * (ntx + 1) // 2) * 2048),
        # factored layout (default): the side blocks of G_nc leave the chip as their rank-<=nvs factors F_nc
        'k_thin_ncf': (d8 * 4 * (3 * t.ntouch) * N + A_self + A_side + d8 * nT, d8 * 4 * nvs * (2 * N + 4 * nvs), 0),
        # factored layout (default): the side blocks of G_bb / G_rdd / G_ab leave the chip as their rank-<=ncf factors
        'k_thin_rt': (R_self + R_side + d8 * nT * (9 + 9 * Q) + d8 * n, d8 * (4 * ncf * (4 * QN + 4) + 4 * QN), 0),
        'k_coupling': (V_self + V_halo + d8 * Q * 4 * ncf * 9, d8 * Q * 4 * N * N, None),
    }
    # factored layout: the three thin kernels share one launch (k_thin3)
    parts = [m['k_coupling'], m['k_thin_rt'], m['k_thin_ncf']]
    m['k_thin3'] = (sum(p[0] for p in parts), sum(p[1] for p in parts), None)
    return {k: (r * S, w * S, (f * S if f is not None else None)) for k, (r, w, f) in m.items()}


def load_pmc_traffic(config):
    """Measured HBM traffic per kernel launch from the committed rocprofv3 PMC passes (tools/pmc_traffic.py writes
    profiles/rNN_pmc_traffic.json: FETCH_SIZE and WRITE_SIZE in separate passes; FETCH_SIZE doubled as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950).  None when no profile of this configuration is on file."""
    import glob
    from pylrbms_amd._build import source_sha
    stale = None
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')), reverse=True):
        try:
            with open(path) as fh:
                doc = json.load(fh)
        except (OSError, ValueError):
            continue
        if doc.get('config') == config:
            doc['file'] = os.path.relpath(path, ROOT)
            # a profile speaks for the kernels it was taken from: same kernel sources, or the traffic figures are withheld.
            # Several sets are on file (earlier rounds, mid-round): the one of THIS build wins.
            doc['stale'] = doc.get('csrc_sha') != source_sha()
            if not doc['stale']:
                return doc
            stale = stale or doc
    return stale


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='cfg3', choices=sorted(CONFIGS) + ['cfg5', 'cfg5_tile8'])
    ap.add_argument('--no-config5', action='store_true', help='skip the config-5 (3D, P2) leg of the default run')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-online', action='store_true')
    ap.add_argument('--opt', action='append', default=[], metavar='NAME=VALUE',
                    help='launch-policy option of the 2D context (NativeContext.OPTIONS), e.g. --opt streams=0; measurement only')
    ap.add_argument('--opt3', action='append', default=[], metavar='NAME=VALUE',
                    help='the same for the 3D context (Native3DContext.OPTIONS), e.g. --opt3 serial=1')
    args = ap.parse_args()
    opts = {k: int(v) for k, v in (o.split('=') for o in args.opt)}
    opts3 = {k: int(v) for k, v in (o.split('=') for o in args.opt3)}

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.config.startswith('cfg5'):
        # BASELINE.json config 5 (3D diffusion, SWIPDG p = 2): its own line (bench3d.py); one rank holds all subdomains
        import bench3d
        line = bench3d.run(args.config, args.steps, args.warmup, device_index=int(os.environ.get('LRBMS_BENCH_DEVICE', local_rank)),
                           cpu=not args.no_cpu_baseline, online=not args.no_online, world=world, rank=rank,
                           backend=os.environ.get('LRBMS_BENCH_BACKEND', 'nccl'), options=opts3)
        if line is not None:
            print(json.dumps(line), flush=True)
        return
    cfg = CONFIGS[args.config]
    N = cfg['N']

    # the CPU baseline runs first: its worker pool is fork()ed from a process that has not initialised the GPU
    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cfg['num_subdomains'], N, cfg['coarse_per_subdomain'])
    cpu5 = None
    do_cfg5 = world == 1 and rank == 0 and args.config == 'cfg3' and not args.no_config5
    if do_cfg5 and not args.no_cpu_baseline:
        try:
            import bench3d
            cpu5 = bench3d.cpu_baseline3d(bench3d.CONFIGS3D['cfg5']['N'])
        except Exception as exc:
            cpu5 = {'error': '{}: {}'.format(type(exc).__name__, exc)}

    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # before the HIP runtime is loaded (dmabuf IPC only)
    import torch
    import torch.distributed as dist
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node {} for --gpus {}'.format(args.gpus, args.gpus))
    # LRBMS_BENCH_BACKEND=gloo LRBMS_BENCH_DEVICE=0: rehearsal of the multi-rank path with all ranks on one GPU (the halo
    # rows are then staged through host memory); the driver's runs use RCCL, one rank per GPU
    backend = os.environ.get('LRBMS_BENCH_BACKEND', 'nccl')
    local_rank = int(os.environ.get('LRBMS_BENCH_DEVICE', local_rank))
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == 'nccl':
            from pylrbms_amd.parallel import init_rccl
            init_rccl(torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)

    from pylrbms_amd import multiscale_problem
    from pylrbms_amd.engine import Engine
    from pylrbms_amd.parallel import Communicator, HaloExchange, HaloPlan
    comm = Communicator(rank, world)
    pcfg = {'num_subdomains': cfg['num_subdomains'], 'coarse_per_subdomain': cfg['coarse_per_subdomain']}
    p = multiscale_problem.init_grid_and_problem(pcfg, mpi_comm=comm)
    grid = p['grid']
    lam = p['lambda']
    theta_bar = np.array([c.evaluate(p['mu_bar']) for c in lam['coefficients']])
    eng = Engine(grid, lam['functions'], p['kappa'], p['f'], p['lambda_bar'], p['lambda_hat'], theta_bar,
                 device_index=local_rank)
    for name, value in opts.items():
        eng.ctx.set_option(name, value)
    eng.assemble()
    torch.cuda.synchronize()
    ta0, ta1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ta0.record()
    eng.assemble()                                   # K1-K6, K8 coefficients, K9 (reported separately)
    ta1.record()
    torch.cuda.synchronize()
    assemble_ms = ta0.elapsed_time(ta1)
    t = grid.template
    S_total = grid.num_subdomains

    # local bases (constant + seeded random columns); the halo slabs are filled by the exchange inside each step
    V = eng.ctx.zeros(eng.S_ext, t.n, N)
    V[:eng.S] = eng.ctx.from_numpy(make_bases_host(eng.local, t.n, N))
    halo = None
    if world > 1:
        from pylrbms_amd.grid import DDSubdomainsGrid
        plan = HaloPlan(lambda r: DDSubdomainsGrid(grid.lower_left, grid.upper_right, grid.K, grid.P, rank=r,
                                                   world_size=world), world, rank)
        halo = HaloExchange(plan, N, V.device)
    buf = eng.alloc_reduce_buffers(N)

    def step():
        # sharded: the halo exchange runs on the communication stream under the halo-independent half of the pass
        eng.project_and_estimate(V, buf, halo=halo)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Every kernel of the pass on its own: HIP event pairs on the stream each kernel is launched on (lrbms_kernel_timing), in a
    # separate untimed loop of the same steps -- and it runs FIRST: after an idle period (the host builds the bases above) the
    # device needs ~25 passes (18 ms) to reach its steady clocks, the first ones run up to 10 % slower (tools/ramp_time.py), and W
    # warm-up passes of 0.7 ms do not cover that.  The timed region below is W warm-up + exactly K timed passes, at steady clocks.
    kernel_ms = None
    if eng.ctx.fused_supported(eng.Q, N, factored=True):
        for _ in range(max(0, 30 - args.steps)):
            step()
        eng.ctx.kernel_timing(True)
        for _ in range(args.steps):
            step()
        rows = eng.ctx.kernel_timing_read()
        eng.ctx.kernel_timing(False)
        if world == 1:
            kernel_ms = {}
            for name, ms in rows:
                kernel_ms.setdefault({'k_f1w': 'k_f1', 'k_f1v': 'k_f1', 'k_f1u': 'k_f1'}.get(name, name), []).append(ms)   # the forms of one step
            kernel_ms = {k: float(np.mean(v)) for k, v in kernel_ms.items()}
    fence()

    for _ in range(args.warmup):
        step()
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    fence()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    tt = torch.tensor([elapsed], dtype=torch.float64, device=V.device)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = S_total * args.steps / elapsed
    # the timed passes left their results in `buf`: every output finite, one checksum on record (same seed => same value)
    outs = {'{}[{}]'.format(k, i): v for k in ('sys', 'grams') for i, v in enumerate(buf[k])}
    bad = [k for k, v in outs.items() if not bool(torch.isfinite(v).all())]
    if bad:
        raise RuntimeError('non-finite outputs of the timed pass: ' + ', '.join(bad))
    # (sum and sum of absolute values over ALL ranks: a sharded run must reproduce the single-rank figures, tests/test_sharded_gpu.py)
    cs = torch.stack([sum(v.sum() for v in outs.values()), sum(v.abs().sum() for v in outs.values())])
    if world > 1:
        dist.all_reduce(cs, op=dist.ReduceOp.SUM)
    output_checksum, output_abs_checksum = float(cs[0]), float(cs[1])

    # the dense (fp64-MFMA) kernels of the pass on their own: phase 4 = k_f1, k_f2, k_f3 (HIP events on the launch stream)
    dense_ms = None
    if world == 1 and eng.ctx.fused_supported(eng.Q, N, factored=True):
        pargs = (V, eng.F, eng.A_diag, eng.A_cpl, eng.P_diag, eng.b, eng.ebar, eng.caa, eng.Aab, eng.Bbb, buf['work'],
                 buf['sys'], buf['grams'])
        eng.ctx.project_estimate_fused(*pargs, phase=4)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            eng.ctx.project_estimate_fused(*pargs, phase=4)
        e1.record()
        torch.cuda.synchronize()
        dense_ms = e0.elapsed_time(e1) / args.steps
        eng.project_and_estimate(V, buf)          # leave complete results in the buffers for the online section

    # the same pass writing the DENSE block-compact layout (G_rdd / G_bb [S][9][QN][QN], G_nc [S][5N][5N]: what rd.operators hands a
    # caller of the reference's API; the timed region above writes the factored layout the reduced estimate consumes)
    dense_layout_ms = None
    if world == 1 and eng.ctx.fused_supported(eng.Q, N, factored=False):
        bufd = eng.alloc_reduce_buffers(N, factored=False)
        eng.__dict__.pop('_bound_pass', None)
        eng.project_and_estimate(V, bufd)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            eng.project_and_estimate(V, bufd)
        e1.record()
        torch.cuda.synchronize()
        dense_layout_ms = e0.elapsed_time(e1) / args.steps
        del bufd
        eng.__dict__.pop('_bound_pass', None)
        torch.cuda.empty_cache()

    online = None
    if world == 1 and not args.no_online:
        # online phase (O1) on the same problem: energy-orthonormalise the local bases (B1) with the projected energy
        # product of the last pass, re-project, then solve 256 parameters in batches of 16
        Lh = np.linalg.cholesky(buf['sys'][2].cpu().numpy())                 # [S, N, N], small: done on the host
        LinvT = eng.ctx.from_numpy(np.linalg.inv(Lh).transpose(0, 2, 1))
        Vo = torch.bmm(V[:eng.S], LinvT).contiguous()                        # columns orthonormal w.r.t. the energy product
        bufo = eng.project_and_estimate(Vo, buf)
        mus = np.random.default_rng(7).uniform(0.1, 1.0, size=256)
        coeffs = lam['coefficients']
        thetas = np.array([[c.evaluate(float(m)) for c in coeffs] for m in mus])
        nb = 64                            # parameters per lrbms_reduced_solve_batch call: four groups of 16 on four streams, inside the library
        eng.ctx.reduced_solve_batch(thetas[:16], bufo['sys'][0], bufo['sys'][1])      # warm-up (also creates the rocBLAS handle)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        # the two-level preconditioner of the reduced model, built ONCE at the middle of the parameter range and used by
        # every batch (inside the timed region: its dense S x S factorisation costs about as much as one batched solve)
        pc = eng.ctx.reduced_precond_build(np.array([c.evaluate(0.55) for c in coeffs]), bufo['sys'][0])
        eng.ctx.reduced_precond_use(pc)
        torch.cuda.synchronize()
        t_pc = time.perf_counter() - t1
        eng.ctx.reduced_solve_batches(thetas[:nb], bufo['sys'][0], bufo['sys'][1], per_call=nb)   # untimed: first use of the side streams
        torch.cuda.synchronize()
        t_warm = time.perf_counter() - t1 - t_pc
        ulist, info = eng.ctx.reduced_solve_batches(thetas, bufo['sys'][0], bufo['sys'][1], per_call=nb, rtol=1e-12, concat=False)
        iters, worst = info['iterations'], info['relative_residual']
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for b, ub in enumerate(ulist):                                        # E1: the arrays of a solve call as they are (passes of 16 inside)
            eng.ctx.reduced_estimate_batch(thetas[b * nb:b * nb + ub.shape[2]], ub, bufo['grams'], eng.f2, eng.ceps, eng.hdiam)
        torch.cuda.synchronize()
        t_est = time.perf_counter() - t2
        dt = time.perf_counter() - t1 - t_warm
        eng.ctx.reduced_precond_use(None)
        # roofline of the solve loop: what one CG iteration of a 64-parameter call moves and multiplies.  Panel matvec: every
        # projected block of B_sys once (existing neighbour slots x Q x N^2 doubles), the direction rows of every slot (z and p_old,
        # N x 64 doubles each), the results y and p; update + preconditioner: x, r, p, y read, x, r, z written, the inverse diagonal
        # blocks; coarse level: the dense S x S inverse.  Flops: the MFMA products of the panel matvec and of Dinv r (padding excluded).
        nslots = int((eng.nbr >= 0).sum())
        Qn = eng.Q
        it_bytes = 8 * (nslots * Qn * N * N + 2 * nslots * N * nb + 2 * eng.S * N * nb      # matvec: blocks, direction rows, y + p
                        + 7 * eng.S * N * nb + eng.S * N * N                                # update: x r p y in, x r z out, Dinv
                        + eng.S * eng.S + 2 * eng.S * nb)                                   # coarse apply
        it_flops = 2.0 * nb * N * N * (nslots * Qn + eng.S) + 2.0 * nb * eng.S * eng.S
        ncalls = (len(mus) + nb - 1) // nb
        t_solve = dt - t_est - t_pc
        online_roof = {'bound': 'hbm', 'unit': 'GB/s', 'peak': 8000.0,
                       'bytes_per_cg_iteration': it_bytes, 'bytes_per_solve_and_iteration': it_bytes / nb,
                       'achieved': 1e-9 * ncalls * iters * it_bytes / t_solve, 'frac': 1e-9 * ncalls * iters * it_bytes / t_solve / 8000.0,
                       'mfma_flops_per_cg_iteration': it_flops, 'mfma_TFLOPs': 1e-12 * ncalls * iters * it_flops / t_solve,
                       'mfma_frac': 1e-12 * ncalls * iters * it_flops / t_solve / 78.6,
                       'basis': 'algorithmic bytes of one CG iteration of a {}-parameter call (B_sys blocks once per call and iteration, direction '
                                'rows, solver vectors, coarse inverse) x {} iterations x {} calls / the solve time without the preconditioner '
                                'build ({:.2f} ms); iterations counted as the slowest parameter of a call needs'.format(nb, iters, ncalls, 1e3 * t_solve)}
        online = {'metric': 'online reduced solves (O1)', 'value': len(mus) / (dt - t_est), 'unit': 'mu-solves/s',
                  'solve_plus_estimate_per_s': len(mus) / dt, 'estimates_per_s': len(mus) / t_est, 'parameters': len(mus),
                  'batch': nb, 'reduced_dim': S_total * N, 'cg_iterations_max': iters, 'relative_residual_max': worst,
                  'preconditioner_build_ms': 1e3 * t_pc, 'groups_in_flight': 1, 'roofline': online_roof,
                  'solver': 'PCG on the block-sparse reduced system, rtol 1e-12, 64 parameters per call as ONE panel of 64 columns (every '
                            'projected block read once per iteration for all 64), preconditioner = inverse diagonal blocks + coarse '
                            'level on the first local basis vectors, built once at mu = 0.55 (time included in value); '
                            'estimates: lrbms_reduced_estimate_batch (local nc / r / df terms of every subdomain)'}

    enrichment = None
    if world == 1 and not args.no_online:
        # online enrichment: the neighbourhood corrector problems of ALL subdomains in one launch
        # (5 * 384 = 1920 unknowns each at config 3), block-Jacobi PCG to rtol 1e-12 inside one workgroup per problem
        th = np.array([c.evaluate(0.3) for c in lam['coefficients']])
        marked = list(range(eng.S))
        eng.local_corrections(th, marked[:8])                                        # warm-up (also assembles D_corr)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        _, cinfo = eng.local_corrections(th, marked)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        enrichment = {'metric': 'neighbourhood corrector solves (solve_for_local_correction)', 'value': len(marked) / dt,
                      'unit': 'local solves/s', 'problems': len(marked), 'unknowns_per_problem': 5 * t.n, 'ms': 1e3 * dt,
                      'cg_iterations_max': int(cinfo[:, 0].max()), 'cg_iterations_mean': float(cinfo[:, 0].mean()),
                      'relative_residual_max': float(cinfo[:, 1].max())}
        # incremental re-projection after a round that marked m subdomains (lrbms_fused_set_subset): the fused pass over the marked
        # subdomains and their neighbours only, into the buffers of the whole pass; 1 024 marked = the whole pass through the same path
        rng_m = np.random.default_rng(11)
        reproj = {}
        for m in (16, 128, eng.S):
            mk = sorted(rng_m.choice(eng.S, size=m, replace=False).tolist())
            subset = eng.touched_targets([eng.local[i] for i in mk])
            eng.project_and_estimate(Vo, buf, subset=subset)          # (Vo: the bases `buf` was projected from -- its rows stay valid)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                eng.project_and_estimate(Vo, buf, subset=subset)
            torch.cuda.synchronize()
            reproj[str(m)] = {'marked': m, 'subdomains_projected': len(subset), 'ms': 1e3 * (time.perf_counter() - t1) / args.steps}
        enrichment['incremental_reprojection'] = reproj

    parabolic = None
    if world == 1 and not args.no_online:
        # parabolic LRBMS: implicit Euler trajectories, full order and reduced, each ONE native call
        # (T = 0.05, 10 steps, mu = 0.5, zero initial data; the reduced model is the online one built above)
        th = np.array([c.evaluate(0.5) for c in lam['coefficients']])
        nt_p, T_p = 10, 0.05
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        _, finfo = eng.ctx.fom_implicit_euler(th, T_p / nt_p, nt_p, eng.A_diag, eng.A_cpl, eng.b)
        torch.cuda.synchronize()
        t_f = time.perf_counter() - t1
        t1 = time.perf_counter()
        _, rinfo = eng.ctx.reduced_implicit_euler(th, T_p / nt_p, nt_p, bufo['sys'][0], bufo['sys'][3], bufo['sys'][1])
        torch.cuda.synchronize()
        t_r = time.perf_counter() - t1
        parabolic = {'metric': 'implicit Euler time steps', 'unit': 'steps/s', 'steps': nt_p, 'T': T_p,
                     'full_order_steps_per_s': nt_p / t_f, 'full_order_dofs': S_total * t.n,
                     'full_order_cg_iterations': finfo['iterations'], 'reduced_steps_per_s': nt_p / t_r,
                     'reduced_dim': S_total * N, 'reduced_cg_iterations': rinfo['iterations'],
                     'relative_residual_max': max(finfo['relative_residual'], rinfo['relative_residual'])}

    # what RCCL (or gloo) actually saw: world size, per-rank subdomain counts and halo bytes, gathered from every rank
    dist_info = {'backend': backend if world > 1 else None, 'world_size': dist.get_world_size() if world > 1 else 1}
    mine = [float(eng.S), float(len(eng.halo)), float(halo.send_bytes if halo is not None else 0),
            float(halo.recv_bytes if halo is not None else 0)]
    if world > 1:
        gathered = [torch.zeros(4, dtype=torch.float64, device=V.device) for _ in range(world)]
        dist.all_gather(gathered, torch.tensor(mine, dtype=torch.float64, device=V.device))
        rows = [g.cpu().tolist() for g in gathered]
    else:
        rows = [mine]
    dist_info['subdomains_per_rank'] = [int(r[0]) for r in rows]
    dist_info['halo_subdomains_per_rank'] = [int(r[1]) for r in rows]
    dist_info['halo_bytes_sent_per_rank_per_step'] = [int(r[2]) for r in rows]
    dist_info['halo_bytes_received_per_rank_per_step'] = [int(r[3]) for r in rows]

    if rank == 0:
        Q = eng.Q
        flops = algorithmic_flops_per_subdomain(t.n, t.n_rt, t.n_T, N, Q)
        byts = algorithmic_bytes_per_subdomain(t.n, t.n_rt, t.n_T, N, Q)
        # device time of one pass on this rank's stream (HIP events on the launch stream)
        dev_s_per_step = 1e-3 * dev_ms / args.steps
        s_rank = eng.S
        model = kernel_model(t, N, Q, s_rank)
        f1_flops = eng.ctx.fused_mfma_per_subdomain(Q, N) * 2048 * s_rank     # what the launcher's form of k_f1 executes (k_f1v: dead
        model['k_f1'] = (model['k_f1'][0], model['k_f1'][1], f1_flops)        # tiles of the symmetric groups skipped, two applies)
        # COMPULSORY bytes of the pass: every input once (basis slabs + the neighbour rows a subdomain reads, assembled
        # operators) and every output once, in the layouts the pass actually writes (G_rdd / G_bb block-compact).  This is
        # what the HBM roofline is priced against; the SURVEY 8(d) "algorithmic" count also charges the intermediates W, R,
        # D and the dense form of the two big Gram operators, which the fused pass never moves -- reported separately below,
        # it is NOT a utilisation figure (it exceeds what HBM can stream).
        inputs = 8 * (t.n * N + 4 * 3 * t.ntouch * N + t.n_T * (36 * Q + 36 + 1 + Q * Q + 9 * Q + 9) + Q * 4 * t.ncf * 9 +
                      Q * t.n_rt * 6 + t.n) * s_rank
        outputs = sum(w for k, (r, w, f) in model.items() if k not in ('k_flux_compact', 'k_vertex_avg', 'k_thin3', 'k_prep_lds'))
        compulsory = inputs + outputs
        ach_gbs = compulsory / dev_s_per_step / 1e9
        pmc = load_pmc_traffic(args.config) if world == 1 else None
        pmc_note = None
        if pmc and pmc['stale']:
            pmc_note = '{} (withheld: the kernel sources changed since that profile was taken)'.format(pmc['file'])
            pmc = None
        traffic = pmc['per_pass_bytes'] if pmc else None
        # The pass is bound by the fp64 matrix pipe, not by HBM: its dominant kernel (k_f1, ~45 % of the pass) and the dense
        # kernels together run at > 0.5 of the MFMA peak while the compulsory bytes need 0.12 of the HBM peak.  Top level = the
        # time-weighted bound: executed fp64-MFMA flops of the dense kernels (k_f1 + k_f2 + k_f3, padding included, as the
        # library counts them) over their measured time; the HBM view of the whole pass stays beside it under `hbm`.
        hbm = {'bound': 'hbm', 'achieved': ach_gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': ach_gbs / PEAK_HBM_GBS,
               'basis': 'compulsory bytes per pass (inputs once + outputs once, compact layouts) / device time of the pass',
               'compulsory_bytes': compulsory, 'compulsory_input_bytes': inputs, 'compulsory_output_bytes': outputs,
               'traffic': traffic, 'traffic_over_compulsory': (traffic / compulsory) if traffic else None,
               'traffic_frac_of_peak': (traffic / dev_s_per_step / 1e9 / PEAK_HBM_GBS) if traffic else None,
               'traffic_source': pmc['file'] if pmc else pmc_note}
        roofline = {'bound': 'mfma', 'achieved': None, 'peak': PEAK_FP64_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': None,
                    'traffic': traffic, 'traffic_source': pmc['file'] if pmc else pmc_note, 'hbm': hbm,
                    'basis': 'executed fp64-MFMA flops (padding included) of the dense kernels k_f1 + k_f2 (+ k_f3 when it runs) / their '
                             'measured time (phase 4 of the pass, HIP events on the launch stream); the dominant kernel alone: dominant_kernel',
                    'kernel': 'fused project+estimate pass: k_prep_lds (flux image, vertex averages and G_nc[self,self] from one copy of '
                              'the basis slab in LDS; without LRBMS_OPT_PREP_LDS: k_flux_compact, k_vertex_avg, k_f3), k_f1, k_f2, k_thin3 '
                              '(= k_coupling, k_thin_rt, k_thin_ncf in one launch)',
                    'device_ms_per_step': 1e3 * dev_s_per_step,
                    'dense_layout_ms_per_step': dense_layout_ms,
                    'survey_8d_algorithmic': {'bytes_per_subdomain': byts, 'flops_per_subdomain': flops,
                                              'note': 'canonical counts of SURVEY 8(d); ~90 % of the flops are structural zeros the pass '
                                                      'skips and the bytes include intermediates it keeps on chip: not executed, not moved',
                                              'GBps': byts * s_rank / dev_s_per_step / 1e9,
                                              'TFLOPs': flops * s_rank / dev_s_per_step / 1e12}}
        if kernel_ms:
            table = []
            for name, ms in sorted(kernel_ms.items(), key=lambda kv: -kv[1]):
                r, w, f = model.get(name, (None, None, None))
                row = {'name': name, 'us': 1e3 * ms, 'bytes_read': r, 'bytes_written': w, 'mfma_flops': f}
                if r is not None:
                    row['hbm_GBps'] = (r + w) / (1e-3 * ms) / 1e9
                    row['hbm_frac'] = row['hbm_GBps'] / PEAK_HBM_GBS
                if f:
                    row['mfma_TFLOPs'] = f / (1e-3 * ms) / 1e12
                    row['mfma_frac'] = row['mfma_TFLOPs'] / PEAK_FP64_MFMA_TFLOPS
                if pmc and name in pmc.get('per_kernel', {}):
                    row['pmc_bytes'] = pmc['per_kernel'][name]['bytes']
                table.append(row)
            roofline['kernels'] = table
            roofline['kernels_sum_us'] = sum(r['us'] for r in table)
            dom = table[0]
            roofline['dominant_kernel'] = dict(dom, bound='mfma' if dom.get('mfma_frac', 0) >= dom.get('hbm_frac', 0) else 'hbm',
                                               frac=max(dom.get('mfma_frac', 0), dom.get('hbm_frac', 0)))
        if dense_ms is not None:
            # (with LRBMS_OPT_PREP_LDS, the default, G_nc[self, self] is produced by k_prep_lds and k_f3 is not launched)
            dense_names = [k for k in ('k_f1', 'k_f2', 'k_f3') if kernel_ms is None or k in kernel_ms]
            mf = sum(model[k][2] for k in dense_names)
            roofline['dense_kernels'] = {'bound': 'mfma', 'achieved': mf / (1e-3 * dense_ms) / 1e12, 'peak': PEAK_FP64_MFMA_TFLOPS,
                                         'unit': 'TFLOP/s', 'frac': mf / (1e-3 * dense_ms) / 1e12 / PEAK_FP64_MFMA_TFLOPS,
                                         'ms': dense_ms, 'executed_mfma_flops': mf,
                                         'kernel': ' + '.join(dense_names) + ' (phase 4 of the pass; executed fp64 MFMA flops, padding '
                                                   'included, over their measured time)'}
            roofline['achieved'], roofline['frac'] = roofline['dense_kernels']['achieved'], roofline['dense_kernels']['frac']
        elif kernel_ms and roofline.get('dominant_kernel', {}).get('mfma_TFLOPs'):
            roofline['achieved'] = roofline['dominant_kernel']['mfma_TFLOPs']
            roofline['frac'] = roofline['dominant_kernel']['mfma_frac']
        # the pass as a whole against the matrix peak: every executed fp64-MFMA flop (the dense kernels, the G_nc fold of the
        # preparation kernel; padding included, as the library counts them) over the time of the whole pass
        pass_flops = sum(f for k, (r, w, f) in model.items() if f and (kernel_ms is None or k in kernel_ms))
        roofline['pass_executed_mfma_flops'] = pass_flops
        roofline['pass_mfma_frac'] = pass_flops / dev_s_per_step / 1e12 / PEAK_FP64_MFMA_TFLOPS
        out = {'metric': 'offline project+estimate throughput', 'value': value, 'unit': 'subdomains/s',
               'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
               'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
               'config': {'workload': '{}: 2D multiscale diffusion, {}x{} subdomains, k_c={} '
                                      '(n={} DG DoFs, n_rt={} RT0 DoFs per subdomain), Q={}, local basis dim {}'
                                      .format({'cfg2': 'BASELINE.json config 2', 'cfg3': 'BASELINE.json config 3'}.get(
                                                  args.config, 'diagnostic variant {} of BASELINE.json config 3'.format(args.config)),
                                              cfg['num_subdomains'][0], cfg['num_subdomains'][1],
                                              cfg['coarse_per_subdomain'], t.n, t.n_rt, Q, N),
                          'subdomains': S_total, 'N': N, 'Q': Q, 'parallelism': 'subdomain tiles x{}'.format(world)},
               'roofline': roofline, 'distributed': dist_info, 'output_checksum': output_checksum, 'output_abs_checksum': output_abs_checksum,
               'timed_region': 'W warm-up + exactly K timed passes between barrier + synchronize, at steady device clocks: the untimed '
                               'per-kernel timing leg (>= 30 passes of the same step) runs in front of it -- after an idle period the '
                               'first ~25 passes run up to 10 % slower (tools/ramp_time.py)'}
        out['assemble'] = {'metric': 'offline assembly K1-K6, K9 (+ flux coefficients)', 'ms': assemble_ms,
                           'value': eng.S / (1e-3 * assemble_ms), 'unit': 'subdomains/s (this rank)'}
        if online is not None:
            out['online'] = online
        if enrichment is not None:
            out['enrichment'] = enrichment
        if parabolic is not None:
            out['parabolic'] = parabolic
        if cpu is not None:                              # reported baseline: rank 0 at N = 1 only (timed before GPU init)
            out['cpu_baseline'] = cpu
        if do_cfg5:
            # BASELINE.json config 5 (3D, P2 tetrahedra, N = 30) on the same GPU, after the config-3 buffers are released
            import bench3d
            del buf, V
            torch.cuda.empty_cache()
            try:
                out['config5'] = bench3d.run('cfg5', steps=args.steps, warmup=args.warmup, device_index=local_rank, cpu=False,
                                             online=not args.no_online, options=opts3)
                if cpu5 is not None:
                    out['config5']['cpu_baseline'] = cpu5
            except Exception as exc:           # the config-3 line must not depend on the second leg: report, do not fail
                out['config5'] = {'error': '{}: {}'.format(type(exc).__name__, exc)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
