"""Drop-in API on the GPU: the call sequence of the reference driver python/scripts/online_adaptive_lrbms.py:65-151
(init problem -> discretize -> d.solve / d.estimate -> ParallelLRBMSReductor(order=0) -> extend_basis(snapshots) ->
reduce -> rd.solve / rd.estimate -> reconstruct -> d.estimate(reconstruction)), checked against the oracle."""
import numpy as np
import pytest

from common import oracle_from_problem
from oracle.lrbms import OracleReductor

pytestmark = pytest.mark.gpu


def _run(problem_module, config, mus, mu_test):
    from pylrbms_amd.discretize_elliptic_block_swipdg import discretize
    from pylrbms_amd.reductor import ExtensionError, ParallelLRBMSReductor
    p = problem_module.init_grid_and_problem(config)
    solver_options = {'max_iter': '400', 'precision': '1e-6', 'type': 'bicgstab.ilut'}
    d, data = discretize(p, solver_options={'inverse': solver_options}, mpi_comm=None)
    block_space = data['block_space']
    o = oracle_from_problem(p)

    # Phase 3: FOM solve + estimate
    mu = d.parse_parameter(mu_test)
    U = d.solve(mu)
    U_ref = o.solve(mu_test)
    assert np.abs(U.data.reshape(o.S, o.n) - U_ref).max() < 1e-8 * np.abs(U_ref).max()
    eta, (nc, r, df), ind = d.estimate(U, mu=mu, decompose=True)
    Ut = U.data.reshape(o.S, o.n)
    eta_o, (nc_o, r_o, df_o), ind_o = o.estimate(Ut, mu_test, decompose=True)
    assert abs(eta - eta_o) < 1e-9 * eta_o
    for a, b in ((nc[:, 0], nc_o), (r[:, 0], r_o), (df[:, 0], df_o), (ind[:, 0], ind_o)):
        assert np.abs(a - b).max() < 1e-9 * np.abs(b).max()

    # reductor with the local energy products, constant shape functions, two snapshots
    reductor = ParallelLRBMSReductor(
        d, products=[d.operators['local_energy_dg_product_{}'.format(ii)] for ii in range(block_space.num_blocks)],
        order=0)
    snaps = []
    for m in mus:
        S_ = d.solve(m)
        snaps.append(S_.data.reshape(o.S, o.n))
        try:
            reductor.extend_basis(S_)
        except ExtensionError:
            pass
    N = reductor.basis_size()
    assert N == 1 + len(mus)
    rd = reductor.reduce()
    assert rd.solution_space.dim == o.S * N
    # the image bases of reductor.py:40-60 are not formed by the fused pass; they are available on demand
    ib = reductor.image_bases()
    assert tuple(ib['OI'].shape) == (o.S, o.n, 5 * N) and tuple(ib['RT'].shape)[0] == o.S and 'OI' in reductor.bases
    assert bool(ib['OI'].isfinite().all()) and bool(ib['RT'].isfinite().all())

    # local bases are energy-orthonormal
    E = rd.E_red.cpu().numpy()
    assert np.abs(E - np.eye(N)[None]).max() < 1e-9

    u = rd.solve(mu)
    ur = reductor.reconstruct(u)
    # oracle: Galerkin solution in the same span (basis independent)
    bases = [np.stack([np.ones(o.n)] + [s[ii] for s in snaps], axis=1) for ii in range(o.S)]
    ored = OracleReductor(o, bases)
    ord_ = ored.reduce()
    ur_o = np.stack(ored.reconstruct(ord_.solve(mu_test)))
    assert np.abs(ur.data.reshape(o.S, o.n) - ur_o).max() < 1e-8 * np.abs(ur_o).max()

    est_red = rd.estimate(u, mu=mu)
    est_rec = d.estimate(ur, mu=mu)
    est_o = o.estimate(ur_o, mu_test)
    assert abs(est_red - est_rec) < 1e-8 * est_rec         # online_adaptive_lrbms.py:145-149
    assert abs(est_red - est_o) < 1e-7 * est_o
    eta2, (nc2, r2, df2), ind2 = rd.estimate(u, mu=mu, decompose=True)
    assert nc2.shape == (o.S, 1) and ind2.shape == (o.S, 1)
    return d, rd, reductor


def test_reference_driver_sequence_os2015():
    from pylrbms_amd import OS2015_academic_problem
    config = {'num_subdomains': [2, 2], 'half_num_fine_elements_per_subdomain_and_dim': 4}   # online_adaptive_lrbms.py:56-61
    d, rd, reductor = _run(OS2015_academic_problem, config, mus=[(0.1,), (1.0,)], mu_test=0.1)
    # the online loop returns immediately when the target is met (online_enrichment.py:81-83)
    from pylrbms_amd.online_enrichment import AdaptiveEnrichment
    loop = AdaptiveEnrichment(None, d, d.data['block_space'], reductor, rd, target_error=1e6, marking_doerfler_theta=0.8,
                              marking_max_age=0)
    U, rd2, _ = loop.solve(0.5, enrichment_steps=1)
    assert rd2 is rd and len(U) == 1
    reductor.enrich_local(0, U, 0.5)                               # reductor.py:75-78 (tests/test_enrichment_gpu.py)
    assert reductor.local_sizes() == [4, 3, 3, 3]


def test_reference_driver_sequence_thermalblock():
    """BASELINE.json config 1: 2D thermal block, 2x2 subdomains."""
    from pylrbms_amd import thermalblock_problem
    config = {'num_subdomains': [2, 2], 'half_num_fine_elements_per_subdomain_and_dim': 4}
    _run(thermalblock_problem, config, mus=[(0.1, 1.0, 1.0, 1.0), (1.0, 0.1, 1.0, 1.0), (1.0, 0.1, 0.3, 0.7)],
         mu_test=(0.4, 0.9, 0.2, 0.6))


def test_fom_apply_matches_the_oracle_operator():
    from pylrbms_amd import multiscale_problem
    from pylrbms_amd.discretize_elliptic_block_swipdg import discretize
    p = multiscale_problem.init_grid_and_problem({'num_subdomains': [3, 4], 'coarse_per_subdomain': 2})
    d, _ = discretize(p)
    o = oracle_from_problem(p)
    rng = np.random.default_rng(8)
    x = rng.standard_normal((o.S, o.n, 3))
    eng = d.engine
    y = eng.ctx.fom_apply(d.theta(0.37), eng.A_diag, eng.A_cpl, eng.ctx.from_numpy(x)).cpu().numpy()
    A = o.assemble_global(0.37)
    ref = (A @ x.reshape(o.ndof, 3)).reshape(o.S, o.n, 3)
    assert np.abs(y - ref).max() < 1e-12 * np.abs(ref).max()


def test_native_fom_solve_matches_the_oracle_and_reports_non_convergence():
    """lrbms_fom_solve (DuneDiscretization._solve, block_swipdg.py:219-225): CG on the never-assembled block operator
    against the oracle's sparse direct solve; rtol 1e-12 on the residual gives <= 1e-8 on the solution here."""
    from pylrbms_amd import multiscale_problem
    from pylrbms_amd._native import NativeError
    from pylrbms_amd.discretize_elliptic_block_swipdg import discretize
    p = multiscale_problem.init_grid_and_problem({'num_subdomains': [3, 4], 'coarse_per_subdomain': 2})
    d, _ = discretize(p)
    o = oracle_from_problem(p)
    eng = d.engine
    x, info = eng.ctx.fom_solve(d.theta(0.37), eng.A_diag, eng.A_cpl, eng.b)
    ref = o.solve(0.37)
    assert info['iterations'] > 0 and info['relative_residual'] <= 1e-12
    assert np.abs(x.cpu().numpy() - ref).max() < 1e-8 * np.abs(ref).max()
    with pytest.raises(NativeError, match='did not reach rtol'):
        eng.ctx.fom_solve(d.theta(0.37), eng.A_diag, eng.A_cpl, eng.b, max_iter=3)


def test_reduced_model_parameter_sweep_and_preconditioner_reuse():
    """``rd.solve_batch(mus)`` (batched PCG, one prebuilt two-level preconditioner per reduced model) returns the same
    solutions as ``rd.solve(mu)`` one by one and as the oracle's dense solves; the preconditioner is built once."""
    from pylrbms_amd import multiscale_problem
    from pylrbms_amd.discretize_elliptic_block_swipdg import discretize
    from pylrbms_amd.reductor import LRBMSReductor
    p = multiscale_problem.init_grid_and_problem({'num_subdomains': [4, 4], 'coarse_per_subdomain': 2})
    d, _ = discretize(p)
    o = oracle_from_problem(p)
    reductor = LRBMSReductor(d, order=0)
    snaps = []
    for mu in (0.15, 0.5, 0.95):
        U = d.solve(mu)
        snaps.append(U.data.reshape(o.S, o.n))
        reductor.extend_basis(U)
    rd = reductor.reduce()
    mus = [0.1 + 0.045 * k for k in range(19)]                       # two batches (16 + 3)
    ub = rd.solve_batch(mus)
    assert len(ub) == len(mus) and len(rd._pc) == 1
    pc_before = next(iter(rd._pc.values()))
    bases = [np.stack([np.ones(o.n)] + [s[ii] for s in snaps], axis=1) for ii in range(o.S)]
    ored = OracleReductor(o, bases)
    ord_ = ored.reduce()
    rec_b = reductor.reconstruct(ub).data.reshape(len(mus), o.S, o.n)
    for k in (0, 7, 18):
        u1 = rd.solve(mus[k])
        assert rd.last_solve_info['relative_residual'] <= 1e-12
        rec_1 = reductor.reconstruct(u1).data.reshape(o.S, o.n)
        ref = np.stack(ored.reconstruct(ord_.solve(mus[k])))
        assert np.abs(rec_1 - ref).max() < 1e-8 * np.abs(ref).max()
        assert np.abs(rec_b[k] - ref).max() < 1e-8 * np.abs(ref).max()
    assert next(iter(rd._pc.values())) is pc_before                   # reused, not rebuilt


def test_product_reproduces_the_reference_scripts_known_answers():
    """python/scripts/linearelliptic_block_swipdg_decomp.py:19-43 through the PRODUCT (not the oracle): OS2015, 4x4
    subdomains, mu = 1; the script prints what its three indicators 'should be' -- 1.66e-01 / 1.45e-01 / 3.55e-01 (the sqrt
    variant of the local indicators).  Residual and diffusive flux are reproduced to the printed digits.  The
    nonconformity indicator follows HEAD's face-neighbour neighbourhoods (block_swipdg.py:78-113) and is 1.2 % above the
    printed value, which belongs to vertex patches over all elements (tests/test_oracle.py): PARITY UNPINNED for it --
    checked against the oracle of the same convention and bounded against the printed value."""
    from pylrbms_amd import OS2015_academic_problem
    from pylrbms_amd.discretize_elliptic_block_swipdg import discretize
    p = OS2015_academic_problem.init_grid_and_problem({'num_subdomains': [4, 4], 'half_num_fine_elements_per_subdomain_and_dim': 4})
    d, _ = discretize(p)
    d.estimator = d.estimator.with_(sqrt_local=True)
    mu = d.parse_parameter(1.)
    U = d.solve(mu)
    eta, (nc, r, df), _ = d.estimate(U, mu=mu, decompose=True)
    assert abs(np.linalg.norm(r) - 1.45e-01) < 0.5e-3
    assert abs(np.linalg.norm(df) - 3.55e-01) < 0.5e-3
    assert abs(np.linalg.norm(nc) / 1.66e-01 - 1.0) < 0.02
    o = oracle_from_problem(p)
    _, (onc, _, _), _ = o.estimate(o.solve(1.0), 1.0, decompose=True, sqrt_local=True)
    assert abs(np.linalg.norm(nc) - np.linalg.norm(onc)) < 1e-9


def test_data_entries_of_discretize_are_usable_operators():
    """``data['local_projections' | 'local_rt_projections' | 'local_oi_projections' | 'local_div_ops']``
    (reference block_swipdg.py:696-729): block picker, sums of the neighbour images on a subdomain, local divergence --
    checked against the oracle's image bases and divergence matrices."""
    from pylrbms_amd import OS2015_academic_problem
    from pylrbms_amd.discretize_elliptic_block_swipdg import discretize
    p = OS2015_academic_problem.init_grid_and_problem({'num_subdomains': [3, 2],
                                                       'half_num_fine_elements_per_subdomain_and_dim': 6})
    d, data = discretize(p, mpi_comm=None)
    o = oracle_from_problem(p)
    S, n = o.S, o.n
    assert [len(data[k]) for k in ('local_projections', 'local_rt_projections', 'local_oi_projections', 'local_div_ops')] == [S] * 4
    U = d.solve(d.parse_parameter(0.4))
    Ut = U.data.reshape(S, n)
    for ii in (0, 4):
        assert np.array_equal(data['local_projections'][ii].apply(U).data.reshape(n), Ut[ii])
    red = OracleReductor(o, [Ut[ii][:, None] for ii in range(S)])
    OI, RT = red.image_bases()
    W = d.estimator.oswald_interpolation_error.apply(U)             # [S, n, 5]
    R = d.estimator.flux_reconstruction.apply(U)                    # [S, n_rt, 5 Q]
    Q = o.Q
    for ii in range(S):
        w_ref = sum(OI[kk][o.mesh.neighborhood_of(kk).index(ii)][:, 0] for kk in o.mesh.neighborhood_of(ii))
        w = data['local_oi_projections'][ii].apply(W).cpu().numpy()[:, 0]
        assert np.abs(w - w_ref).max() < 1e-11 * max(np.abs(w_ref).max(), np.abs(Ut).max())
        r_ref = sum(RT[kk][o.mesh.neighborhood_of(kk).index(ii)] for kk in o.mesh.neighborhood_of(ii))   # [n_rt, Q]
        r = data['local_rt_projections'][ii].apply(R).cpu().numpy()
        assert r.shape == (o.n_rt[ii], Q)
        assert np.abs(r - r_ref).max() < 1e-11 * np.abs(r_ref).max()
        div = data['local_div_ops'][ii]
        Dm = o.Div[ii].toarray()
        assert np.abs(div.matrix() - Dm).max() < 1e-12 * np.abs(Dm).max()
        got = div.apply(d.engine.ctx.from_numpy(r_ref)).cpu().numpy()
        assert np.abs(got - Dm @ r_ref).max() < 1e-11 * np.abs(Dm @ r_ref).max()
