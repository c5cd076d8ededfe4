"""GPU parity of the 3D / P2 HIP path (BASELINE.json config 5) against the CPU oracle (oracle/lrbms3d.py), through the C ABI
of include/lrbms3d_hip.h.  PARITY UNPINNED beyond the oracle: the reference has no 3D / P2 counterpart
(discretize_elliptic_block_swipdg.py:22-23), the oracle itself is validated by properties (tests/test_oracle3d.py).

Tolerances: 1e-11 relative (max norm) for every assembled / projected array and the estimator terms, 1e-10 for the reduced
solve (the north_star bound)."""
import numpy as np
import pytest

import common3d as c3

pytestmark = pytest.mark.gpu

TOL = 1e-11


@pytest.fixture(scope='module', params=list(c3.PROBLEMS))
def case(request):
    import torch
    from pylrbms_amd.engine3d import Engine3D, expand_factored
    p = c3.make_problem(request.param)
    d = c3.oracle_of(p)
    eng = Engine3D(p['grid'], p['lambdas'], p['f'], p['lambda_bar'], p['lambda_hat'], theta_bar=c3.theta_of(p, p['mu_bar'])).assemble()
    V = c3.make_bases3d(d.S, d.n, p['N'], seed=3)
    Vd = eng.ctx.from_numpy(V)
    out = eng.project_and_estimate(Vd)
    torch.cuda.synchronize()
    rd = c3.reduce_with_oracle(p, d, V)
    return dict(p=p, d=d, eng=eng, V=V, Vd=Vd, out=out, rd=rd, dense=expand_factored(eng, out, d.Q, p['N']))


def test_assembled_operators_match_the_oracle(case):
    p, d, eng = case['p'], case['d'], case['eng']
    ref = c3.oracle_assembled(p, d)
    for k in ('A_diag', 'A_cpl', 'b', 'f2', 'ceps', 'bdiv', 'ebar', 'Aaa', 'Aab', 'Bbb', 'Cf', 'P_diag'):
        got = eng.ops[k].cpu().numpy().reshape(ref[k].shape)
        assert c3.rel(got, ref[k]) < TOL, (k, c3.rel(got, ref[k]))


def test_block_operator_is_the_global_operator(case):
    p, d, eng = case['p'], case['d'], case['eng']
    rng = np.random.default_rng(1)
    x = rng.standard_normal((d.S, d.n, 3))
    th = c3.theta_of(p, p['mu'])
    y = eng.ctx.fom_apply(d.Q, th, eng.ops['A_diag'], eng.ops['A_cpl'], eng.ctx.from_numpy(x)).cpu().numpy()
    ref = d.system_matrix(p['mu']) @ x.reshape(d.ndof, 3)
    assert c3.rel(y.reshape(d.ndof, 3), ref) < TOL


def test_projected_operators_match_the_oracle(case):
    p, d, rd, dense = case['p'], case['d'], case['rd'], case['dense']
    got = {k: v.cpu().numpy() for k, v in dense.items()}
    worst = {}
    for ii in range(d.S):
        ref = c3.oracle_dense_blocks(p, d, rd, ii)
        for k in ('G_nc', 'G_bb', 'G_rdd', 'r_fd'):
            worst[k] = max(worst.get(k, 0.0), c3.rel(got[k][ii], ref[k]))
        worst['G_ab'] = max(worst.get('G_ab', 0.0), c3.rel(got['G_ab'][:, ii], ref['G_ab']))
        worst['G_aa'] = max(worst.get('G_aa', 0.0), c3.rel(got['G_aa'][:, :, ii], ref['G_aa']))
        worst['B_sys'] = max(worst.get('B_sys', 0.0), c3.rel(got['B_sys'][:, ii], ref['B_sys']))
        worst['rhs_red'] = max(worst.get('rhs_red', 0.0), c3.rel(got['rhs_red'][ii], rd.rhs[ii]))
    assert all(v < TOL for v in worst.values()), worst


def test_every_k_split_of_the_projection_kernels_matches_the_oracle(case):
    """LRBMS3_OPT_KSPLIT: 1 = one workgroup per (subdomain, operator) with the in-kernel epilogues (H + H^T through the LDS, mirrored
    G_aa store, G_bb / G_rdd / r_fd epilogue, direct rhs_red) -- the path BASELINE.json config 5 runs at its 512 subdomains, which the
    automatic choice never takes at test sizes; 2, 5, 8 = partial tiles + k3_pg_combine.  Every setting against the oracle at the
    parity tolerance, against the automatic choice to summation-order rounding, and the estimate from the ksplit = 1 operators."""
    import torch
    p, d, eng, rd, out, Vd = case['p'], case['d'], case['eng'], case['rd'], case['out'], case['Vd']
    from pylrbms_amd.engine3d import expand_factored
    N = p['N']
    refs = [c3.oracle_dense_blocks(p, d, rd, ii) for ii in range(d.S)]
    work = eng.alloc_work(N)
    try:
        for ks in (1, 2, 5, 8):
            eng.ctx.set_option('ksplit', ks)
            o2 = eng.alloc_outputs(N)
            for v in o2.values():
                v.fill_(float('nan'))
            eng.project_and_estimate(Vd, o2, work)
            torch.cuda.synchronize()
            for k in out:
                scale = float(out[k].abs().max())
                assert float((out[k] - o2[k]).abs().max()) <= 1e-12 * max(scale, 1e-300), (ks, k)
            got = {k: v.cpu().numpy() for k, v in expand_factored(eng, o2, d.Q, N).items()}
            worst = 0.0
            for ii, ref in enumerate(refs):
                for k in ('G_nc', 'G_bb', 'G_rdd', 'r_fd'):
                    worst = max(worst, c3.rel(got[k][ii], ref[k]))
                worst = max(worst, c3.rel(got['G_ab'][:, ii], ref['G_ab']), c3.rel(got['G_aa'][:, :, ii], ref['G_aa']),
                            c3.rel(got['B_sys'][:, ii], ref['B_sys']), c3.rel(got['rhs_red'][ii], rd.rhs[ii]))
            assert worst < TOL, (ks, worst)
            if ks == 1:
                u = np.random.default_rng(5).standard_normal((d.S, N))
                th = c3.theta_of(p, p['mu'])
                eta = eng.reduced_estimate(th, eng.ctx.from_numpy(u), o2).cpu().numpy()
                for got_, ref_ in zip(eta, rd.local_terms([u[ii] for ii in range(d.S)], p['mu'])):
                    assert c3.rel(got_, ref_) < 1e-10
                us, (it, res) = eng.reduced_solve(th, o2, rtol=1e-13)
                assert c3.rel(us.cpu().numpy(), np.stack(rd.solve(p['mu']))) < 1e-10
                # repeated passes are bit-identical (fixed summation order)
                o3 = eng.alloc_outputs(N)
                eng.project_and_estimate(Vd, o3, work)
                torch.cuda.synchronize()
                for k in o2:
                    assert torch.equal(o2[k], o3[k]), k
    finally:
        eng.ctx.set_option('ksplit', 0)


def test_estimator_terms_match_the_oracle(case):
    p, d, eng, rd, out = case['p'], case['d'], case['eng'], case['rd'], case['out']
    rng = np.random.default_rng(5)
    u = rng.standard_normal((d.S, p['N']))
    th = c3.theta_of(p, p['mu'])
    eta = eng.reduced_estimate(th, eng.ctx.from_numpy(u), out).cpu().numpy()
    nc, r, df = rd.local_terms([u[ii] for ii in range(d.S)], p['mu'])
    for got, ref, name in ((eta[0], nc, 'nc'), (eta[1], r, 'r'), (eta[2], df, 'df')):
        assert c3.rel(got, ref) < 1e-10, (name, c3.rel(got, ref))
    # ... which is the full-order estimate of the reconstruction (the defining property of the projection)
    red = rd.reductor
    full = d.local_terms(red.reconstruct([u[ii] for ii in range(d.S)]), p['mu'])
    for got, ref in zip(eta, full):
        assert c3.rel(got, ref) < 1e-9


def test_reduced_solve_matches_the_oracle(case):
    p, d, eng, rd, out = case['p'], case['d'], case['eng'], case['rd'], case['out']
    th = c3.theta_of(p, p['mu'])
    u, (it, res) = eng.reduced_solve(th, out, rtol=1e-13)
    ref = np.stack(rd.solve(p['mu']))
    assert res <= 1e-13 and it > 0
    assert c3.rel(u.cpu().numpy(), ref) < 1e-10, c3.rel(u.cpu().numpy(), ref)


def test_product_reproduces_the_committed_3d_fixtures():
    """The HIP path on the inputs stored in tests/golden/cfg5_*.npz against the stored expected outputs (generated with the
    oracle by tests/golden/make_golden.py)."""
    import os
    from pylrbms_amd.engine3d import Engine3D, expand_factored
    here = os.path.dirname(os.path.abspath(__file__))
    for name in ('aniso_2x2x1', 'q3_2x1x2'):
        gold = np.load(os.path.join(here, 'golden', 'cfg5_' + name + '.npz'))
        p = c3.make_problem(name)
        eng = Engine3D(p['grid'], p['lambdas'], p['f'], p['lambda_bar'], p['lambda_hat']).assemble()
        Q, N = eng.Q, int(gold['N'])
        out = eng.project_and_estimate(eng.ctx.from_numpy(gold['V']))
        dense = {k: v.cpu().numpy() for k, v in expand_factored(eng, out, Q, N).items()}
        th = c3.theta_of(p, float(gold['mu'][0]))
        assert c3.rel(eng.ops['b'].cpu().numpy().ravel(), gold['b']) < TOL
        assert c3.rel(eng.ops['f2'].cpu().numpy(), gold['f2']) < TOL and c3.rel(eng.ops['ceps'].cpu().numpy(), gold['ceps']) < TOL
        assert c3.rel(dense['rhs_red'], gold['rhs_red']) < TOL
        for k in ('G_nc', 'G_bb', 'G_rdd', 'r_fd'):
            assert c3.rel(dense[k][0], gold[k + '_0']) < TOL, k
        assert c3.rel(dense['G_ab'][:, 0], gold['G_ab_0']) < TOL and c3.rel(dense['G_aa'][:, :, 0], gold['G_aa_0']) < TOL
        assert c3.rel(dense['B_sys'][:, 0], gold['B_sys_0']) < TOL
        eta = eng.reduced_estimate(th, eng.ctx.from_numpy(gold['u_random']), out).cpu().numpy()
        for got, key in zip(eta, ('eta_nc', 'eta_r', 'eta_df')):
            assert c3.rel(got, gold[key]) < 1e-10, key
        u, _ = eng.reduced_solve(th, out, rtol=1e-13)
        assert c3.rel(u.cpu().numpy(), gold['u']) < 1e-10


@pytest.mark.parametrize('name,N', [('q1_strip', 64), ('aniso_2x2x1', 32), ('q1_strip', 1), ('q3_2x1x2', 7)])
def test_limit_sizes_of_the_pass(name, N):
    """The largest basis sizes the pass takes (N = 64 with Q = 1: 4 x 4 tiles per wave; Q N = 64), the smallest (N = 1) and an odd
    size with three components (the 8-byte operand form of the kernels), against the oracle; one size beyond the limit is refused."""
    from pylrbms_amd._native import NativeError
    from pylrbms_amd.engine3d import Engine3D, expand_factored
    p = c3.make_problem(name)
    p['N'] = N
    d = c3.oracle_of(p)
    eng = Engine3D(p['grid'], p['lambdas'], p['f'], p['lambda_bar'], p['lambda_hat']).assemble()
    V = c3.make_bases3d(d.S, d.n, max(N, 2), seed=5)[:, :, -N:].copy()      # N = 1: a non-constant column (gradients do not vanish)
    out = eng.project_and_estimate(eng.ctx.from_numpy(V))
    dense = {k: v.cpu().numpy() for k, v in expand_factored(eng, out, d.Q, N).items()}
    rd = c3.reduce_with_oracle(p, d, V)
    worst = 0.0
    for ii in range(d.S):
        ref = c3.oracle_dense_blocks(p, d, rd, ii)
        for k in ('G_nc', 'G_bb', 'G_rdd', 'r_fd'):
            worst = max(worst, c3.rel(dense[k][ii], ref[k]))
        worst = max(worst, c3.rel(dense['G_ab'][:, ii], ref['G_ab']), c3.rel(dense['G_aa'][:, :, ii], ref['G_aa']),
                    c3.rel(dense['B_sys'][:, ii], ref['B_sys']), c3.rel(dense['rhs_red'][ii], rd.rhs[ii]))
    assert worst < TOL, worst
    u = np.random.default_rng(1).standard_normal((d.S, N))
    eta = eng.reduced_estimate(c3.theta_of(p, p['mu']), eng.ctx.from_numpy(u), out).cpu().numpy()
    for got, want in zip(eta, rd.local_terms([u[ii] for ii in range(d.S)], p['mu'])):
        assert c3.rel(got, want) < 1e-10
    if name == 'aniso_2x2x1':
        with pytest.raises(NativeError):
            eng.project_and_estimate(eng.ctx.from_numpy(c3.make_bases3d(d.S, d.n, 33, seed=5)))       # Q N = 66 > 64


def test_batched_reduced_solve_matches_single_solves_and_the_oracle(case):
    p, d, eng, rd, out = case['p'], case['d'], case['eng'], case['rd'], case['out']
    if p['N'] > 32:
        pytest.skip('batched solve takes N <= 32')
    mus = [0.15, p['mu'], 0.55, 0.9, 1.2][:5]
    thetas = np.stack([c3.theta_of(p, mu) for mu in mus])
    ub, (it, res) = eng.ctx.reduced_solve_batch(d.Q, thetas, out['B_sys'], out['rhs_red'], rtol=1e-13)
    assert res <= 1e-13 and it > 0
    for m, mu in enumerate(mus):
        ref = np.stack(rd.solve(mu))
        assert c3.rel(ub[:, :, m].cpu().numpy(), ref) < 1e-10, (m, c3.rel(ub[:, :, m].cpu().numpy(), ref))
        us, _ = eng.reduced_solve(thetas[m], out, rtol=1e-13)
        assert c3.rel(ub[:, :, m].cpu().numpy(), us.cpu().numpy()) < 1e-10


def test_batched_solve_of_several_groups_on_their_streams(case):
    """nmu > 16: groups of 16 parameters as independent CG runs on the caller's stream and the library's side streams, launches
    interleaved, each group writing its own columns of u [S, N, nmu] -- every column equals the oracle's solution, with and
    without the prebuilt preconditioner; 17 and 64 parameters cover a one-column last group and four full groups."""
    p, d, eng, rd, out = case['p'], case['d'], case['eng'], case['rd'], case['out']
    if p['N'] > 32:
        pytest.skip('batched solve takes N <= 32')
    for nmu in (17, 64):
        mus = list(np.linspace(0.15, 1.2, nmu))
        thetas = np.stack([c3.theta_of(p, mu) for mu in mus])
        ub, (it, res) = eng.ctx.reduced_solve_batch(d.Q, thetas, out['B_sys'], out['rhs_red'], rtol=1e-13)
        assert tuple(ub.shape) == (eng.S, p['N'], nmu) and res <= 1e-13 and it > 0
        ref = [np.stack(rd.solve(mu)) for mu in mus]
        for m in range(nmu):
            assert c3.rel(ub[:, :, m].cpu().numpy(), ref[m]) < 1e-10, (nmu, m)
        eng.ctx.reduced_precond_use(eng.ctx.reduced_precond_build(d.Q, c3.theta_of(p, 0.6), out['B_sys']))
        try:
            u2, (_, res2) = eng.ctx.reduced_solve_batch(d.Q, thetas, out['B_sys'], out['rhs_red'], rtol=1e-13)
        finally:
            eng.ctx.reduced_precond_use(None)
        assert res2 <= 1e-13
        for m in range(nmu):
            assert c3.rel(u2[:, :, m].cpu().numpy(), ref[m]) < 1e-10, (nmu, m, 'pc')


def test_batched_solve_with_the_prebuilt_coarse_level(case):
    """lrbms3_reduced_precond_build / _use: the Galerkin coarse problem on the first local basis vectors, inverted once at a
    reference parameter, added to the inverse diagonal blocks -- the same solutions as the oracle for every parameter of the
    batch (any SPD preconditioner is admissible); switched off again by use(None)."""
    p, d, eng, rd, out = case['p'], case['d'], case['eng'], case['rd'], case['out']
    if p['N'] > 32:
        pytest.skip('batched solve takes N <= 32')
    mus = [0.15, p['mu'], 0.55, 0.9, 1.2]
    thetas = np.stack([c3.theta_of(p, mu) for mu in mus])
    u0, (it0, _) = eng.ctx.reduced_solve_batch(d.Q, thetas, out['B_sys'], out['rhs_red'], rtol=1e-13)
    pc = eng.ctx.reduced_precond_build(d.Q, c3.theta_of(p, 0.6), out['B_sys'])
    A0 = np.zeros((eng.S, eng.S))                                   # the coarse matrix from the projected blocks, on the host
    Bm = np.einsum('q,qsabc->sabc', c3.theta_of(p, 0.6), out['B_sys'].cpu().numpy())
    for s_ in range(eng.S):
        for slot in range(7):
            t_ = s_ if slot == 3 else eng.nbr[s_, slot]
            if t_ >= 0:
                A0[s_, t_] = Bm[s_, slot, 0, 0]
    assert c3.rel(pc[:eng.S * eng.S].cpu().numpy().reshape(eng.S, eng.S), np.linalg.inv(A0)) < 1e-9
    eng.ctx.reduced_precond_use(pc)
    try:
        u1, (it1, res1) = eng.ctx.reduced_solve_batch(d.Q, thetas, out['B_sys'], out['rhs_red'], rtol=1e-13)
    finally:
        eng.ctx.reduced_precond_use(None)
    # a preconditioner frozen at one parameter may cost iterations on a tiny problem whose parameters spread far (theta = mu^2
    # here); it pays with the subdomain count (bench3d: 48 -> 24 at 8 x 8 x 8).  Counted in steps of 8 without, 12 with it.
    assert res1 <= 1e-13 and 0 < it1 <= 2 * it0
    for m, mu in enumerate(mus):
        assert c3.rel(u1[:, :, m].cpu().numpy(), np.stack(rd.solve(mu))) < 1e-10
    u2, (it2, _) = eng.ctx.reduced_solve_batch(d.Q, thetas, out['B_sys'], out['rhs_red'], rtol=1e-13)
    assert it2 == it0 and np.array_equal(u2.cpu().numpy(), u0.cpu().numpy())


def test_phased_pass_is_bit_identical_to_the_whole_pass(case):
    """lrbms3_project_estimate_phase: 1 (rank-local slabs only) followed by 2 (neighbour rows) == 0, bit for bit."""
    import torch
    eng, Vd, out = case['eng'], case['Vd'], case['out']
    N = Vd.shape[2]
    out2, work = eng.alloc_outputs(N), eng.alloc_work(N)
    for v in out2.values():
        v.fill_(float('nan'))
    work.fill_(float('nan'))      # a recycled allocation may still hold the intermediates of the whole pass
    eng.ctx.project_estimate(eng.Q, Vd, eng.ops, work, out2, phase=1)
    eng.ctx.project_estimate(eng.Q, Vd, eng.ops, work, out2, phase=2)
    torch.cuda.synchronize()
    for k in out:
        assert torch.equal(out[k], out2[k]), k


def test_batched_estimate_matches_single_estimates(case):
    p, d, eng, rd, out = case['p'], case['d'], case['eng'], case['rd'], case['out']
    rng = np.random.default_rng(8)
    nmu = 11                                            # two passes of the kernel: 8 + 3
    mus = list(rng.uniform(0.1, 1.3, size=nmu))
    U = rng.standard_normal((d.S, p['N'], nmu))
    thetas = np.stack([c3.theta_of(p, mu) for mu in mus])
    eta = eng.ctx.reduced_estimate_batch(d.Q, thetas, eng.ctx.from_numpy(U), out, eng.ops, eng.hdiam).cpu().numpy()
    for m, mu in enumerate(mus):
        ref = rd.local_terms([U[ii, :, m] for ii in range(d.S)], mu)
        one = eng.reduced_estimate(thetas[m], eng.ctx.from_numpy(np.ascontiguousarray(U[:, :, m])), out).cpu().numpy()
        for k in range(3):
            assert c3.rel(eta[k, :, m], ref[k]) < 1e-10 and c3.rel(eta[k, :, m], one[k]) < 1e-12, (m, k)
