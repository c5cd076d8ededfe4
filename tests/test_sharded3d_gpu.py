"""The N > 1 path of the 3D / P2 configuration on a real GPU: 2 and 4 ranks share cuda:0, each owns a tile of subdomains, the
halo rows travel by HaloExchange (gloo here with host staging; RCCL on a multi-GPU node), and every rank's outputs of the pass
must equal the single-rank result for its own subdomains (exercises the S_ext > S indexing of every 3D kernel); the estimator
norms are the fused all-reduce of the 2D path."""
import os
import pickle

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

P, KC, N = (4, 2, 1), 2, 6


def _problem(rank=0, world=1):
    from pylrbms_amd import multiscale_problem3d
    return multiscale_problem3d.init_grid_and_problem({'num_subdomains': P, 'cubes_per_subdomain': KC}, rank=rank, world_size=world)


def _engine(p):
    from pylrbms_amd.engine3d import Engine3D
    lam = p['lambda']
    return Engine3D(p['grid'], lam['functions'], p['f'], p['lambda_bar'], p['lambda_hat']).assemble()


def _bases(S, n):
    return np.random.default_rng(77).standard_normal((S, n, N))


def _worker(rank, world, port, ref_path, outdir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pylrbms_amd.grid3d import DDSubdomainsGrid3D
        from pylrbms_amd.parallel import HaloExchange, HaloPlan, global_norms
        from pylrbms_amd.discretize_elliptic_block_swipdg_3d import BlockDiscretization3D
        p = _problem(rank, world)
        grid = p['grid']
        d = BlockDiscretization3D(p)
        eng = d.engine
        n = grid.template.n
        Vg = _bases(grid.num_subdomains, n)
        V = eng.ctx.zeros(eng.S_ext, n, N)
        V[eng.S:] = float('nan')                       # the halo slabs hold nothing until the exchange has run: phase 1 must not
                                                       # read them (rows the exchange does not fill are never read at all)
        V[:eng.S] = eng.ctx.from_numpy(Vg[eng.local])
        plan = HaloPlan(lambda r: DDSubdomainsGrid3D(grid.lower_left, grid.upper_right, grid.K, grid.P, rank=r, world_size=world),
                        world, rank)
        halo = HaloExchange(plan, N, V.device)
        out = eng.project_and_estimate(V, halo=halo)
        torch.cuda.synchronize()
        ref = pickle.load(open(ref_path, 'rb'))
        worst = 0.0
        for k, v in out.items():
            got = v.cpu().numpy()
            want = ref[k]
            axis = {'B_sys': 1, 'G_ab': 1, 'Xab': 1, 'G_aa': 2}.get(k, 0)
            want = np.take(want, eng.local, axis=axis)
            if k == 'B_sys':
                pass
            worst = max(worst, float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-300)))
        # estimate with global coefficients: needs u of the halo subdomains too
        u_g = ref['u']
        u = eng.ctx.from_numpy(u_g[eng.ext])
        eta = eng.reduced_estimate(np.array([1.0, 0.4]), u, out).cpu().numpy()
        worst_eta = float(np.abs(eta - ref['eta'][:, eng.local]).max() / np.abs(ref['eta']).max())
        norms = global_norms(torch.from_numpy(eta[0]), torch.from_numpy(eta[1] + eta[2]))
        want_n = np.array([np.linalg.norm(ref['eta'][0]), np.linalg.norm(ref['eta'][1] + ref['eta'][2])])
        # the combined estimate of the sharded discretization is the GLOBAL one (estimators.py:100-101 are mpi_norm), and the
        # decomposition keeps this rank's local indicators
        eta_c, _, ind = d.combine(eta, 0.4, decompose=True)
        err_c = abs(eta_c - ref['eta_combined']) / ref['eta_combined']
        err_i = float(np.abs(ind - ref['indicators'][eng.local]).max() / np.abs(ref['indicators']).max())
        with open(os.path.join(outdir, 'result_{}.pkl'.format(rank)), 'wb') as fh:
            pickle.dump((worst, worst_eta, float(np.abs(norms.numpy() - want_n).max() / want_n.max()), eng.S, eng.S_ext,
                         max(err_c, err_i)), fh)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_sharded_3d_pass_matches_the_single_rank_pass(world, tmp_path):
    from pylrbms_amd.discretize_elliptic_block_swipdg_3d import BlockDiscretization3D
    p = _problem()
    d = BlockDiscretization3D(p)
    eng = d.engine
    Vg = _bases(p['grid'].num_subdomains, p['grid'].template.n)
    out = eng.project_and_estimate(eng.ctx.from_numpy(Vg))
    u = np.random.default_rng(3).standard_normal((eng.S, N))
    eta = eng.reduced_estimate(np.array([1.0, 0.4]), eng.ctx.from_numpy(u), out).cpu().numpy()
    ref = {k: v.cpu().numpy() for k, v in out.items()}
    eta_c, _, ind = d.combine(eta, 0.4, decompose=True)
    ref.update(u=u, eta=eta, eta_combined=float(eta_c), indicators=ind)
    del d
    ref_path = str(tmp_path / 'ref.pkl')
    pickle.dump(ref, open(ref_path, 'wb'))
    del eng, out
    torch.cuda.empty_cache()
    outdir = str(tmp_path / 'results')
    os.makedirs(outdir, exist_ok=True)
    port = 29300 + (os.getpid() % 500) + world
    mp.spawn(_worker, args=(world, port, ref_path, outdir), nprocs=world, join=True)
    for r in range(world):
        worst, worst_eta, wn, S, S_ext, wc = pickle.load(open(os.path.join(outdir, 'result_{}.pkl'.format(r)), 'rb'))
        assert S_ext > S == 8 // world
        assert worst < 1e-12 and worst_eta < 1e-11 and wn < 1e-12 and wc < 1e-11, (r, worst, worst_eta, wn, wc)


def _solve_worker(rank, world, port, ref_path, outdir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pylrbms_amd.discretize_elliptic_block_swipdg_3d import BlockDiscretization3D
        d = BlockDiscretization3D(_problem(rank, world))
        U, (it, res) = d.solve(0.4, rtol=1e-12, return_info=True)
        want = pickle.load(open(ref_path, 'rb'))[d.engine.local]
        err = float(np.abs(U.cpu().numpy() - want).max() / np.abs(want).max())
        with open(os.path.join(outdir, 'solve_{}.pkl'.format(rank)), 'wb') as fh:
            pickle.dump((err, it, res, tuple(U.shape)), fh)
    finally:
        dist.destroy_process_group()


def test_sharded_3d_snapshot_solve_gathers_the_block_operator(tmp_path):
    """d.solve(mu) on a sharded 3D discretization: every rank gathers A_diag / A_cpl / b once and solves redundantly through a
    second context with the global neighbour table (as the 2D path does); each rank returns its own rows of the single-rank
    solution."""
    from pylrbms_amd.discretize_elliptic_block_swipdg_3d import BlockDiscretization3D
    d = BlockDiscretization3D(_problem())
    U = d.solve(0.4, rtol=1e-12).cpu().numpy()
    ref_path = str(tmp_path / 'U.pkl')
    pickle.dump(U, open(ref_path, 'wb'))
    del d
    torch.cuda.empty_cache()
    outdir = str(tmp_path / 'results')
    os.makedirs(outdir, exist_ok=True)
    world = 2
    mp.spawn(_solve_worker, args=(world, 29900 + (os.getpid() % 90), ref_path, outdir), nprocs=world, join=True)
    for r in range(world):
        err, it, res, shape = pickle.load(open(os.path.join(outdir, 'solve_{}.pkl'.format(r)), 'rb'))
        assert shape[0] == 8 // world and it > 0 and res <= 1e-12 and err < 1e-9, (r, err, it, res, shape)
