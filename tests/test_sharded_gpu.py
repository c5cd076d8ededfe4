"""The N > 1 path on a real GPU: 2 (and 4) ranks share cuda:0, each owns a tile of subdomains, the halo rows travel by
the same HaloExchange (gloo here; RCCL on a multi-GPU node), and every rank's projected operators must equal the
single-rank result for its own subdomains.  This exercises the S_ext > S indexing of every kernel."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

PX, PY, KC, N = 4, 3, 2, 5
# (subdomains x, y, coarse squares per subdomain and direction, local basis size): 'small' exercises odd N (k_f1u, the streaming
# sweeps); 'cfg3' is the template of BASELINE.json config 3 / 4 (k_c = 4, N = 40) on a 6 x 4 grid -- the kernels a sharded run of
# config 4 launches (k_f1v<3,2,1,2,4>, k_prep_lds<3> with the G_nc fold, its slab-less 256-thread phase-2 instance, k_thin3<3>)
SHAPES = {'small': (PX, PY, KC, N), 'cfg3': (6, 4, 4, 40), 'vp': (4, 4, 2, 6)}


def _problem(comm=None, shape='small'):
    from pylrbms_amd import multiscale_problem
    px, py, kc, _ = SHAPES[shape]
    return multiscale_problem.init_grid_and_problem({'num_subdomains': [px, py], 'coarse_per_subdomain': kc}, mpi_comm=comm)


def _engine(p):
    from pylrbms_amd.engine import Engine
    lam = p['lambda']
    theta_bar = np.array([c.evaluate(p['mu_bar']) for c in lam['coefficients']])
    return Engine(p['grid'], lam['functions'], p['kappa'], p['f'], p['lambda_bar'], p['lambda_hat'], theta_bar).assemble()


def _bases(S, n, N=N):
    rng = np.random.default_rng(77)
    return rng.standard_normal((S, n, N))


def _put(outdir, rank, value):
    """Workers hand their results back through files: the parent has initialised the GPU, so it must not fork() a
    manager process (mp.Manager() does); mp.spawn starts the workers with the 'spawn' method."""
    import pickle
    with open(os.path.join(outdir, 'result_{}.pkl'.format(rank)), 'wb') as fh:
        pickle.dump(value, fh)


def _spawn_and_collect(fn, args, world, tmp_path):
    import pickle
    outdir = str(tmp_path / 'results')
    os.makedirs(outdir, exist_ok=True)
    mp.spawn(fn, args=tuple(args) + (outdir,), nprocs=world, join=True)
    out = {}
    for r in range(world):
        path = os.path.join(outdir, 'result_{}.pkl'.format(r))
        if os.path.exists(path):
            with open(path, 'rb') as fh:
                out[r] = pickle.load(fh)
    return out


def _worker(rank, world, port, ref_path, shape, results):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pylrbms_amd.engine import Engine
        from pylrbms_amd.grid import DDSubdomainsGrid
        from pylrbms_amd.parallel import Communicator, HaloExchange, HaloPlan
        N = SHAPES[shape][3]
        p = _problem(Communicator(rank, world), shape)
        grid = p['grid']
        eng = _engine(p)
        n = grid.template.n
        Vg = _bases(grid.num_subdomains, n, N)
        # halo exchange on CPU tensors (gloo), then upload
        Vh = torch.zeros(eng.S_ext, n, N, dtype=torch.float64)
        Vh[:eng.S] = torch.from_numpy(Vg[eng.local])
        plan = HaloPlan(lambda r: DDSubdomainsGrid(grid.lower_left, grid.upper_right, grid.K, grid.P, rank=r,
                                                   world_size=world), world, rank)
        HaloExchange(plan, N, Vh.device)(Vh)
        V = Vh.to(eng.ctx.device)
        ref = np.load(ref_path)
        ok = True
        worst = 0.0
        ran = {}
        modes = (False, True, 'phased', 'overlap', 'phased/unforked', 'phased/unforked/streaming', 'overlap/unforked',
                 'overlap/serial', 'overlap/unforked/serial')
        for mode in modes:
            # '/unforked': the launch policy of a rank with >= 192 subdomains (every kernel its own launch on one stream), forced
            # here at a small count; '/streaming': the preparation by the streaming sweeps (k_flux_side + k_vertex_side in phase 2);
            # '/serial': Engine.SERIAL_PHASES_FROM reached -- the two halves one after the other on ONE stream, the branch a rank
            # with >= 384 subdomains takes (2 GPUs at config 4), halo slabs poisoned until the exchange has finished
            eng.ctx.set_option('streams', 0 if 'unforked' in str(mode) else -1)
            eng.ctx.set_option('prep_lds', 0 if 'streaming' in str(mode) else 1)
            Engine.SERIAL_PHASES_FROM = 1 if 'serial' in str(mode) else 384
            fused = str(mode).split('/')[0] if isinstance(mode, str) else mode
            eng.ctx.kernel_timing(True)
            if fused == 'overlap':
                # the production form of a sharded pass: Engine drives the exchange (gloo here: staged through the host)
                # and the stream choreography -- dense kernels on the main stream, halo-dependent ones on a side stream
                Vo = torch.zeros_like(V)
                Vo[:eng.S] = V[:eng.S]
                Vo[eng.S:] = float('nan')
                buf = eng.project_and_estimate(Vo, eng.alloc_reduce_buffers(N), halo=HaloExchange(plan, N, Vo.device))
                torch.cuda.synchronize()
            elif fused == 'phased':
                # the overlapped form: phase 1 must not touch the halo slabs (poisoned with NaN while it runs), phase 2
                # runs after the exchange has filled them
                buf = eng.alloc_reduce_buffers(N)
                Vp = V.clone()
                Vp[eng.S:] = float('nan')
                args = (Vp, eng.F, eng.A_diag, eng.A_cpl, eng.P_diag, eng.b, eng.ebar, eng.caa, eng.Aab, eng.Bbb, buf['work'],
                        buf['sys'], buf['grams'])
                eng.ctx.project_estimate_fused(*args, phase=1)
                torch.cuda.synchronize()
                Vp[eng.S:] = V[eng.S:]
                eng.ctx.project_estimate_fused(*args, phase=2)
            else:
                buf = eng.project_and_estimate(V, eng.alloc_reduce_buffers(N, factored=bool(fused)), fused=fused)
            ran[str(mode)] = sorted({k for k, _ in eng.ctx.kernel_timing_read()})
            eng.ctx.kernel_timing(False)
            from pylrbms_amd.engine import expand_factored_grams
            names = ('B_sys', 'rhs_red', 'E_red', 'M_red', 'G_nc', 'r_fd', 'G_rdd', 'G_bb', 'G_ab', 'G_aa')
            for name, arr in zip(names, list(buf['sys']) + list(expand_factored_grams(buf['grams']))):
                a = arr.cpu().numpy()
                r = ref[name]
                if name in ('B_sys', 'G_ab'):
                    r = r[:, eng.local]
                elif name == 'G_aa':
                    r = r[:, :, eng.local]
                else:
                    r = r[eng.local]
                err = np.abs(a - r).max() / max(np.abs(r).max(), 1e-300)
                worst = max(worst, err)
                ok &= bool(err < 1e-12)
        Engine.SERIAL_PHASES_FROM = 384
        eng.ctx.set_option('streams', -1)
        if shape == 'cfg3':
            # the kernels config 4 launches did run, in every mode that is meant to take them: the lean projection kernel, the
            # preparation from the LDS copy of the slab, and in the phased forms its slab-less instance for the neighbours' shares
            for mode, names_ran in ran.items():
                if mode == 'False':
                    continue
                ok &= 'k_f1w' in names_ran and 'k_f1v' not in names_ran and 'k_f1u' not in names_ran and 'k_f1' not in names_ran
                if 'streaming' in mode:
                    ok &= 'k_prep_lds' not in names_ran and 'k_flux_side' in names_ran and 'k_vertex_side' in names_ran
                else:
                    ok &= 'k_prep_lds' in names_ran and 'k_f3' not in names_ran
                    if mode != 'True':
                        ok &= 'k_prep_lds<side>' in names_ran
                ok &= 'k_thin3' in names_ran and 'k_f2' in names_ran
        # sharded estimate: local indicators + fused norms
        theta = np.array([1.0, 0.4])
        u_g = np.random.default_rng(5).standard_normal((grid.num_subdomains, N))
        u = eng.ctx.from_numpy(u_g[eng.ext])
        eta = eng.reduced_estimate(theta, u, buf['grams']).cpu().numpy()
        ok &= bool(np.abs(eta - ref['eta'][:, eng.local]).max() < 1e-10 * np.abs(ref['eta']).max())
        # API level: sharded discretize -> reductor with the same bases -> reduce (halo exchange inside) -> rd.solve, which
        # gathers the reduced system on every rank and solves it through a second context with the global neighbour table
        from pylrbms_amd.discretize_elliptic_block_swipdg import discretize
        from pylrbms_amd.reductor import LRBMSReductor
        del eng, buf, V
        d, _ = discretize(_problem(Communicator(rank, world), shape), mpi_comm=Communicator(rank, world))
        red = LRBMSReductor(d, bases={'domain_{}'.format(ii): Vg[ii].T for ii in d.engine.local})
        rd = red.reduce()
        u_loc = rd.solve(0.4).tensor[:, :, 0].cpu().numpy()
        err_u = np.abs(u_loc - ref['u_solve'][d.engine.local]).max() / np.abs(ref['u_solve']).max()
        ok &= bool(err_u < 1e-9)
        # the batched sweep on the gathered system: column 1 is the same parameter as the single solve
        ub = rd.solve_batch([0.9, 0.4, 0.15]).tensor.cpu().numpy()
        ok &= bool(ub.shape == (d.engine.S, Vg.shape[2], 3))
        ok &= bool(np.abs(ub[:, :, 1] - u_loc).max() < 1e-9 * np.abs(u_loc).max())
        _put(results, rank, (ok, max(worst, err_u), d.engine.S, d.engine.S_ext, ran))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world, shape', [(2, 'small'), (4, 'small'), (2, 'cfg3'), (4, 'cfg3')])
def test_sharded_projection_matches_single_rank(world, shape, tmp_path):
    N = SHAPES[shape][3]
    p = _problem(shape=shape)
    eng = _engine(p)
    grid = p['grid']
    V = eng.ctx.from_numpy(_bases(grid.num_subdomains, grid.template.n, N))
    buf = eng.project_and_estimate(V, fused=False)
    names = ('B_sys', 'rhs_red', 'E_red', 'M_red', 'G_nc', 'r_fd', 'G_rdd', 'G_bb', 'G_ab', 'G_aa')
    out = {k: v.cpu().numpy() for k, v in zip(names, list(buf['sys']) + list(buf['grams']))}
    u_g = np.random.default_rng(5).standard_normal((grid.num_subdomains, N))
    out['eta'] = eng.reduced_estimate(np.array([1.0, 0.4]), eng.ctx.from_numpy(u_g), buf['grams']).cpu().numpy()
    out['u_solve'] = eng.reduced_solve(np.array([1.0, 0.4]), buf['sys'][0], buf['sys'][1])[0].cpu().numpy()
    ref_path = str(tmp_path / 'ref.npz')
    np.savez(ref_path, **out)
    del eng, buf, V
    torch.cuda.empty_cache()
    port = 29500 + 2 * (os.getpid() % 100) + world + (10 if shape == 'cfg3' else 0)        # disjoint port ranges per test of this file
    results = _spawn_and_collect(_worker, (world, port, ref_path, shape), world, tmp_path)
    assert len(results) == world
    for r in range(world):
        ok, worst, S, S_ext, ran = results[r]
        assert ok, (r, worst, ran)
        assert S_ext > S


def _vertex_patch_worker(rank, world, port, ref_path, results):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pylrbms_amd.engine import Engine
        from pylrbms_amd.grid import DDSubdomainsGrid
        from pylrbms_amd.parallel import Communicator, HaloExchange, HaloPlan
        N = SHAPES['vp'][3]
        comm = Communicator(rank, world)
        p = _problem(comm, 'vp')
        grid, lam = p['grid'], p['lambda']
        theta_bar = np.array([c.evaluate(p['mu_bar']) for c in lam['coefficients']])
        eng = Engine(grid, lam['functions'], p['kappa'], p['f'], p['lambda_bar'], p['lambda_hat'], theta_bar,
                     conventions={'oswald_vertex_patch': True}).assemble()
        ref = np.load(ref_path)
        Vg, U, thetas = ref['V'], ref['U'], ref['thetas']
        plan = HaloPlan(lambda r: DDSubdomainsGrid(grid.lower_left, grid.upper_right, grid.K, grid.P, rank=r, world_size=world),
                        world, rank, diagonal=True)
        assert plan.S_ext == eng.S_ext > eng.S
        ok, worst = True, 0.0
        for mode in ('overlap', 'overlap/serial', 'whole', 'whole/streaming'):
            eng.ctx.set_option('prep_lds', 0 if 'streaming' in mode else 1)
            Engine.SERIAL_PHASES_FROM = 1 if 'serial' in mode else 384
            V = torch.full((eng.S_ext, grid.template.n, N), float('nan'), dtype=torch.float64, device=eng.ctx.device)
            V[:eng.S] = eng.ctx.from_numpy(Vg[eng.local])
            hx = HaloExchange(plan, N, V.device)
            if mode.startswith('overlap'):       # the production choreography: exchange under the halo-independent half
                buf = eng.project_and_estimate(V, eng.alloc_reduce_buffers(N), halo=hx)
            else:
                # rows of the halo slabs nobody reads stay NaN; the pass must not touch them
                buf = eng.project_and_estimate(hx(V), eng.alloc_reduce_buffers(N))
            torch.cuda.synchronize()
            for m in range(U.shape[2]):
                u = eng.ctx.from_numpy(np.ascontiguousarray(U[eng.ext, :, m]))
                eta = eng.reduced_estimate(thetas[m], u, buf['grams']).cpu().numpy()
                for row in range(3):
                    want = ref['eta'][m, row][eng.local]
                    err = np.abs(eta[row] - want).max() / np.abs(ref['eta'][m, row]).max()
                    worst = max(worst, float(err))
                    ok &= bool(err < 1e-10)
            ub = eng.ctx.from_numpy(np.ascontiguousarray(U[eng.ext]))
            eta_b = eng.ctx.reduced_estimate_batch(thetas, ub, buf['grams'], eng.f2, eng.ceps, eng.hdiam).cpu().numpy()
            for m in range(U.shape[2]):
                for row in range(3):
                    ok &= bool(np.abs(eta_b[row, :, m] - ref['eta'][m, row][eng.local]).max() < 1e-10 * np.abs(ref['eta'][m, row]).max())
        Engine.SERIAL_PHASES_FROM = 384
        del eng, buf, V
        # API level: the same through discretize(conventions=) -> reductor -> rd.estimate (global norms all-reduced)
        from pylrbms_amd.discretize_elliptic_block_swipdg import discretize
        from pylrbms_amd.reductor import LRBMSReductor
        from pylrbms_amd.vectorarrays import ReducedVectorArray
        d, _ = discretize(_problem(comm, 'vp'), mpi_comm=comm, conventions={'oswald_vertex_patch': True})
        rd = LRBMSReductor(d, bases={'domain_{}'.format(ii): Vg[ii].T for ii in d.engine.local}).reduce()
        u0 = ReducedVectorArray(d.engine.ctx.from_numpy(np.ascontiguousarray(U[d.engine.local, :, :1])))
        est = float(rd.estimate(u0, mu=float(ref['mus'][0])))
        ok &= bool(abs(est - float(ref['est0'])) < 1e-9 * float(ref['est0']))
        _put(results, rank, (ok, worst, est))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_vertex_patch_on_a_sharded_grid_matches_the_oracle(world, tmp_path):
    """conventions={'oswald_vertex_patch': True} (the reading that reproduces the reference's printed nonconformity value,
    linearelliptic_block_swipdg_decomp.py:41) on 2 and 4 ranks: the diagonal neighbours are halo slabs of their own
    (lrbms_set_diagonal_neighbours), their corner rows travel with the one halo exchange (HaloPlan(diagonal=True); 4 ranks: the
    cross point of the four tiles needs corner items), and every rank's local estimator terms equal the ORACLE's with
    ``oswald_patch='vertex'`` (1e-10) -- whole pass, streaming preparation, and the overlapped / serial-phase choreography with
    the halo poisoned until the exchange has filled it; rd.estimate through the API equals the single-rank oracle value."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from common import energy_orthonormalize, make_bases, oracle_from_problem, theta_of
    from oracle.lrbms import OracleReductor
    N = SHAPES['vp'][3]
    p = _problem(shape='vp')
    d = oracle_from_problem(p, oswald_patch='vertex')
    V = energy_orthonormalize(make_bases(d.S, d.n, N, seed=31), d)
    rd = OracleReductor(d, [V[ii] for ii in range(d.S)]).reduce()
    mus = [0.2, 0.9]
    U = np.random.default_rng(9).standard_normal((d.S, N, len(mus)))
    eta = np.zeros((len(mus), 3, d.S))
    for m, mu in enumerate(mus):
        _, (nc, r, df), _ = rd.estimate([U[ii, :, m] for ii in range(d.S)], mu, decompose=True)
        eta[m] = np.stack([nc, r, df])
    est0 = rd.estimate([U[ii, :, 0] for ii in range(d.S)], mus[0])
    ref_path = str(tmp_path / 'ref.npz')
    np.savez(ref_path, V=V, U=U, thetas=np.stack([theta_of(p, mu) for mu in mus]), eta=eta, mus=np.array(mus), est0=float(est0))
    port = 29400 + (os.getpid() % 90) + world
    results = _spawn_and_collect(_vertex_patch_worker, (world, port, ref_path), world, tmp_path)
    assert len(results) == world
    for r in range(world):
        ok, worst, est = results[r]
        assert ok, (r, worst, est, float(est0))


def _bench_line(nranks, port):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LRBMS_BENCH_BACKEND='gloo', LRBMS_BENCH_DEVICE='0')
    tail = [os.path.join(root, 'bench.py'), '--gpus', str(nranks), '--steps', '3', '--warmup', '1', '--config', 'cfg2',
            '--no-cpu-baseline', '--no-online']
    if nranks > 1:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nranks), '--master-addr',
               '127.0.0.1', '--master-port', str(port)] + tail
    else:
        cmd = [sys.executable] + tail
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1                                   # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_two_rank_rehearsal():
    """bench.py's N > 1 code path (sharded engine, asynchronous halo exchange under phase 1, max-over-ranks timing) with
    two ranks sharing cuda:0 over gloo (halo rows staged through host memory); the driver's runs use RCCL.  The two ranks
    together must have computed what one rank computes: sum and sum of absolute values of every output of the timed pass,
    all-reduced over the ranks, against the single-rank run of the same config (same seeded bases by global subdomain index)."""
    out = _bench_line(2, 29900 + (os.getpid() % 90))
    assert out['n_gpus'] == 2 and out['value'] > 0 and out['scaling'] == 'strong' and 'cpu_baseline' not in out
    one = _bench_line(1, 0)
    assert one['n_gpus'] == 1 and one['output_abs_checksum'] > 0
    assert abs(out['output_abs_checksum'] - one['output_abs_checksum']) <= 1e-10 * one['output_abs_checksum']
    assert abs(out['output_checksum'] - one['output_checksum']) <= 1e-10 * one['output_abs_checksum']


def _enrichment_run(p, comm=None):
    """Two rounds of AdaptiveEnrichment.solve on the problem (sharded if ``comm`` is given): (eta, local sizes by subdomain)."""
    from pylrbms_amd.discretize_elliptic_block_swipdg import discretize
    from pylrbms_amd.online_enrichment import AdaptiveEnrichment
    from pylrbms_amd.reductor import LRBMSReductor
    d, data = discretize(p, mpi_comm=comm)
    reductor = LRBMSReductor(d, order=0)
    reductor.extend_basis(d.solve(0.7))      # a snapshot: on a sharded discretization the gathered operator, solved natively
    rd = reductor.reduce()
    ae = AdaptiveEnrichment(p, d, data['block_space'], reductor, rd, target_error=1e-12, marking_doerfler_theta=0.5,
                            marking_max_age=2)
    log = []
    U, rd, reductor = ae.solve(0.4, enrichment_steps=2, callback=lambda rd_, U_, mu_, info: log.append(info['eta']))
    eta = rd.estimate(U, mu=d.parse_parameter(0.4))
    return float(eta), dict(zip(d.engine.local, reductor.local_sizes())), [float(e) for e in log]


def _enrich_worker(rank, world, port, results):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pylrbms_amd.parallel import Communicator
        _put(results, rank, _enrichment_run(_problem(Communicator(rank, world)), Communicator(rank, world)))
    finally:
        dist.destroy_process_group()


def test_adaptive_enrichment_on_a_sharded_discretization(tmp_path):
    """Online enrichment across ranks: global Doerfler / age marking from all-gathered indicators, every rank solves the
    corrector problems of its own marked subdomains (operator blocks of the halo assembled on the rank), ragged bases
    padded to the global width for the halo exchange -- same estimates and local basis sizes as the single-rank run."""
    eta1, sizes1, log1 = _enrichment_run(_problem())
    # (the estimate need not fall: at the reference's HEAD the corrector ignores the current solution)
    assert len(log1) == 3 and max(sizes1.values()) > min(sizes1.values())          # ragged after marking
    torch.cuda.empty_cache()
    world = 2
    port = 29300 + (os.getpid() % 150)
    results = _spawn_and_collect(_enrich_worker, (world, port), world, tmp_path)
    sizes = {}
    for r in range(world):
        eta, loc, log = results[r]
        assert abs(eta - eta1) < 1e-8 * eta1 and np.allclose(log, log1, rtol=1e-8)
        sizes.update(loc)
    assert sizes == sizes1


def _parabolic_run(p, comm=None):
    from pylrbms_amd.discretize_parabolic_block_swipdg import discretize
    from pylrbms_amd.reductor import ParabolicLRBMSReductor
    d, _ = discretize(p, 0.2, 4, mpi_comm=comm)
    U = d.solve(0.6)
    reductor = ParabolicLRBMSReductor(d, order=0)
    reductor.extend_basis(U[[2, 4]])
    rd = reductor.reduce()
    u = rd.solve(0.6)
    est = (float(d.estimate(U, mu=0.6)[0]), float(rd.estimate(u, mu=0.6)[0]))
    return (dict(zip(d.engine.local, U.tensor.cpu().numpy())), dict(zip(d.engine.local, reductor.reconstruct(u).tensor.cpu().numpy())),
            est)


def _parabolic_worker(rank, world, port, results):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pylrbms_amd.parallel import Communicator
        _put(results, rank, _parabolic_run(_problem(Communicator(rank, world)), Communicator(rank, world)))
    finally:
        dist.destroy_process_group()


def test_parabolic_solves_on_a_sharded_discretization(tmp_path):
    """Full-order and reduced implicit Euler trajectories on two ranks (gathered operators, native solvers on every rank)
    and their parabolic estimates equal the single-rank ones."""
    U1, R1, est1 = _parabolic_run(_problem())
    torch.cuda.empty_cache()
    world = 2
    port = 29100 + (os.getpid() % 150)
    results = _spawn_and_collect(_parabolic_worker, (world, port), world, tmp_path)
    seen = 0
    for r in range(world):
        U, R, est = results[r]
        # the parabolic estimates (elliptic part sharded, time residual on the gathered operators, d_t nc all-reduced)
        assert abs(est[0] - est1[0]) < 1e-8 * est1[0] and abs(est[1] - est1[1]) < 1e-7 * est1[1], (est, est1)
        for g in U:
            assert np.abs(U[g] - U1[g]).max() < 1e-9 * np.abs(U1[g]).max()
            assert np.abs(R[g] - R1[g]).max() < 1e-8 * np.abs(R1[g]).max()
            seen += 1
    assert seen == PX * PY
