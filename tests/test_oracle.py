"""Oracle pinning (CPU): structural invariants of SURVEY.md section 4, the soft known-answer values printed by the
reference's python/scripts/linearelliptic_block_swipdg_decomp.py:41-43, and the committed golden fixtures."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))

from common import energy_orthonormalize, make_bases, oracle_from_problem
from oracle.lrbms import OracleDiscretization, OracleReductor
from oracle.mesh import OracleMesh
from oracle.quadrature import TRI_W
from pylrbms_amd import OS2015_academic_problem, multiscale_problem

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _os2015(K, P, **kw):
    cos = lambda x: np.cos(0.5 * np.pi * x[..., 0]) * np.cos(0.5 * np.pi * x[..., 1])  # noqa: E731
    mesh = OracleMesh([-1, -1], [1, 1], K, P)
    th = [lambda mu: 1.0, lambda mu: float(np.ravel(mu)[0])]
    one = lambda x, c, k: 1.0 + 0.0 * cos(x)  # noqa: E731
    return OracleDiscretization(mesh, [lambda x, c, k: 1 + cos(x), lambda x, c, k: -cos(x)], th, np.eye(2),
                                lambda x, c, k: 0.5 * np.pi ** 2 * cos(x), one, one, 1.0, 1.0, **kw), cos


@pytest.fixture(scope='module')
def os4():
    return _os2015([4, 4], [4, 4])


def test_known_answers_of_the_reference_script(os4):
    """linearelliptic_block_swipdg_decomp.py:41-43: OS2015, 4x4 subdomains, mu = 1, sqrt variant of the local
    indicators: 1.66e-01 / 1.45e-01 / 3.55e-01.  Residual and diffusive-flux values are reproduced to the printed
    digits.  The nonconformity value is reproduced by the oracle's ``oswald_patch='vertex'`` mode (vertex patch of the
    Oswald interpolation over ALL elements at a vertex, i.e. including the diagonal subdomain at a cross point -- the
    older global estimator those numbers were printed for).  With the neighbourhood structure of HEAD (face neighbours
    only, block_swipdg.py:78-113: ``grid.neighborhood_of``) it comes out 1.2 % higher: PARITY UNPINNED for that
    convention -- no reference number exists for it, so only its distance to the printed value is bounded here."""
    d, _ = os4
    U = d.solve(1.0)
    _, (nc, r, df), _ = d.estimate(U, 1.0, decompose=True, sqrt_local=True)
    assert abs(np.linalg.norm(r) - 1.45e-01) < 0.5e-3
    assert abs(np.linalg.norm(df) - 3.55e-01) < 0.5e-3
    assert abs(np.linalg.norm(nc) / 1.66e-01 - 1.0) < 0.02          # HEAD neighbourhoods: unpinned, 1.2 % off
    dv, _ = _os2015([4, 4], [4, 4], oswald_patch='vertex')
    _, (nc, r, df), _ = dv.estimate(U, 1.0, decompose=True, sqrt_local=True)
    assert abs(np.linalg.norm(nc) - 1.66e-01) < 0.5e-3
    assert abs(np.linalg.norm(r) - 1.45e-01) < 0.5e-3
    assert abs(np.linalg.norm(df) - 3.55e-01) < 0.5e-3


def test_system_is_symmetric_positive_definite_and_blocks_match_global(os4):
    d, _ = os4
    A = d.assemble_global(0.37)
    assert abs(A - A.T).max() < 1e-13
    assert np.linalg.eigvalsh(A.toarray()).min() > 0
    # block operator == global operator (block_swipdg.py:452-471 vs :475-497)
    x = np.random.default_rng(0).standard_normal(d.ndof)
    y = np.zeros_like(x)
    th = d.theta(0.37)
    for ii in range(d.S):
        for jj in d.mesh.neighborhood_of(ii):
            for q in range(d.Q):
                y[ii * d.n:(ii + 1) * d.n] += th[q] * (d.block(d.A[q], ii, jj) @ x[jj * d.n:(jj + 1) * d.n])
    assert np.abs(y - A @ x).max() < 1e-12 * np.abs(y).max()
    # local + coupling + boundary parts add up and hit only their own blocks
    for q in range(d.Q):
        assert abs(d.A[q] - (d.A_local[q] + d.A_coupling[q] + d.A_boundary[q])).max() < 1e-14
        assert d.block(d.A_local[q], 0, 1).nnz == 0


def test_discretization_converges_and_flux_is_locally_conservative():
    errs = []
    for K in ([4, 4], [8, 8], [16, 16]):
        d, cos = _os2015(K, [2, 2])
        U = d.solve(1.0)                       # lambda = 1: -lap u = f, u = cos cos
        uex = cos(d.mesh.points).reshape(d.S, d.n)
        diff = (U - uex).reshape(-1)
        errs.append(np.sqrt(diff @ (d.l2_product @ diff)))
        th, u = d.theta(1.0), U.reshape(-1)
        r = sum(th[q] * (d.F[q] @ u) for q in range(d.Q))
        m = d.mesh
        out = (m.elem_face_sign * m.face_length[m.elem_face] * r[m.elem_face]).sum(axis=1)
        assert np.abs(out - d.b.reshape(-1, 3).sum(axis=1)).max() < 1e-12
    rates = np.log2(np.array(errs[:-1]) / np.array(errs[1:]))
    assert np.all(rates > 1.7)                 # P1: L2 rate 2


def test_nonconformity_vanishes_for_continuous_functions_with_zero_trace(os4):
    d, cos = os4
    U = cos(d.mesh.points).reshape(d.S, d.n)   # continuous P1 interpolant, zero on the boundary of [-1,1]^2
    _, (nc, _, _), _ = d.estimate(U, 1.0, decompose=True)
    # the quadratic form is evaluated through per-source Gram blocks (as the reduced estimator does), so the
    # cancellation leaves O(eps) relative to the O(0.1) block entries
    assert np.abs(nc).max() < 1e-14


def test_reduced_estimate_equals_full_estimate_of_the_reconstruction():
    """rd.estimate(u) == d.estimate(reconstruct(u)) (online_adaptive_lrbms.py:145-149)."""
    p = multiscale_problem.init_grid_and_problem({'num_subdomains': [3, 2], 'coarse_per_subdomain': 2})
    d = oracle_from_problem(p)
    V = energy_orthonormalize(make_bases(d.S, d.n, 4, seed=1), d)
    red = OracleReductor(d, [V[ii] for ii in range(d.S)])
    rd = red.reduce()
    u = rd.solve(0.6)
    eta, (nc, r, df), ind = rd.estimate(u, 0.6, decompose=True)
    U = np.stack(red.reconstruct(u))
    eta2, (nc2, r2, df2), ind2 = d.estimate(U, 0.6, decompose=True)
    for a, b in ((eta, eta2), (nc, nc2), (r, r2), (df, df2), (ind, ind2)):
        assert np.allclose(a, b, rtol=1e-9, atol=1e-14)
    # projection identity (item 3) and Galerkin orthogonality of the reduced solution
    A = d.assemble_global(0.6).toarray()
    Vg = np.zeros((d.ndof, sum(rd.sizes)))
    off = np.concatenate(([0], np.cumsum(rd.sizes)))
    for ii in range(d.S):
        Vg[ii * d.n:(ii + 1) * d.n, off[ii]:off[ii + 1]] = V[ii]
    Ared, bred, _ = rd.assemble(0.6)
    assert np.abs(Ared - Vg.T @ A @ Vg).max() < 1e-11 * np.abs(Ared).max()
    assert np.abs(bred - Vg.T @ d.b).max() < 1e-13


def test_estimator_quirks_are_switchable(os4):
    d, _ = os4
    assert d.alpha(0.5, 1.0, first_only=True) == 1.0          # estimators.py:121 returns inside the loop
    assert d.alpha(0.5, 1.0, first_only=False) == 0.5
    assert d.gamma(0.5, 1.0) == 1.0 and d.gamma(2.0, 1.0) == 2.0


@pytest.mark.parametrize('name', ['os2015_2x2', 'thermalblock_2x2', 'multiscale_3x3'])
def test_oracle_reproduces_golden_fixtures(name):
    from make_golden import build
    _, out = build(name)
    ref = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    for key in ref.files:
        a, b = np.asarray(out[key], dtype=np.float64), ref[key]
        assert a.shape == b.shape, key
        assert np.abs(a - b).max() <= 1e-9 * max(np.abs(b).max(), 1e-300), key


def test_local_correction_on_a_whole_domain_neighbourhood_is_the_global_solution():
    """Pins the from-scratch neighbourhood assembly (block_swipdg.py:227-316) against the global one: for 3 x 1
    subdomains N(1) is the whole domain, its Dirichlet boundary the physical one."""
    from pylrbms_amd import OS2015_academic_problem
    p = OS2015_academic_problem.init_grid_and_problem({'num_subdomains': [3, 1],
                                                       'half_num_fine_elements_per_subdomain_and_dim': 6})
    d = oracle_from_problem(p)
    u = d.solve(0.3)
    x = d.solve_for_local_correction(1, 0.3)
    assert np.abs(u[1] - x).max() < 1e-12 * np.abs(u[1]).max()
    # a proper neighbourhood: symmetric positive definite, and NOT the global solution (Dirichlet cut-off)
    A, b, hood, dofs = d.local_correction_system(0, 0.3)
    assert hood == [0, 1] and A.shape == (2 * d.n, 2 * d.n)
    assert abs(A - A.T).max() < 1e-13 and np.linalg.eigvalsh(A.toarray()).min() > 0.0
    assert np.abs(d.solve_for_local_correction(0, 0.3) - u[0]).max() > 1e-3 * np.abs(u[0]).max()


def test_parabolic_oracle_invariants():
    """oracle/parabolic.py (parity unpinned: the reference's parabolic path does not run at HEAD): the implicit Euler
    trajectory tends to the stationary solution; the reduced estimate equals the full-order estimate of the
    reconstruction in every part that does not involve the inverse mass; the elliptic-reconstruction terms of
    estimators.py:80-83 add up to ||M^-1 A u - div r||^2 - ||Pi f||^2 (checked against an independent evaluation)."""
    import scipy.sparse.linalg as spla
    from oracle.parabolic import OracleParabolic, OracleParabolicReduced
    d, _ = _os2015([2, 2], [2, 2])
    mu = 0.5
    U_inf = OracleParabolic(d, 200.0, 20).solve(mu)[-1]
    Ust = d.solve(mu)
    assert np.abs(U_inf - Ust).max() < 1e-10 * np.abs(Ust).max()

    par = OracleParabolic(d, 1.0, 6)
    U = par.solve(mu)
    assert np.abs(U[0]).max() == 0.0 and U.shape == (7, d.S, d.n)
    rng = np.random.default_rng(0)
    N = 4
    bases = [np.linalg.qr(np.hstack([U[1:, ii].T, rng.standard_normal((d.n, 1))]))[0][:, :N] for ii in range(d.S)]
    red = OracleReductor(d, bases)
    pr = OracleParabolicReduced(red, red.reduce(), 1.0, 6)
    u = pr.solve(mu)
    off = np.arange(d.S + 1) * N
    Urec = np.stack([np.stack([bases[ii] @ u[k, off[ii]:off[ii + 1]] for ii in range(d.S)]) for k in range(u.shape[0])])
    for rec in (False, True):
        _, parts_r = pr.estimate(u, mu, elliptic_reconstruction=rec)
        _, parts_f = par.estimate(Urec, mu, elliptic_reconstruction=rec)
        for i in ((0, 2, 4) if rec else (0, 1, 2, 4)):      # with the reconstruction, r involves M_red^-1 as well
            assert np.abs(parts_r[i] - parts_f[i]).max() < 1e-9 * np.abs(parts_f[i]).max()

    _, r1, _ = par._elliptic_local(U, mu, True)
    A, Minv, th, k = d.assemble_global(mu), spla.splu(d.l2_product.tocsc()), d.theta(mu), 3
    BU_R, F_R = Minv.solve(A @ U[k].reshape(-1)), Minv.solve(d.b)
    _, RT = OracleReductor(d, [U[k, ii][:, None] for ii in range(d.S)]).image_bases()
    for ii in range(d.S):
        ur = sum(RT[kk][d.mesh.neighborhood_of(kk).index(ii)] @ th for kk in d.mesh.neighborhood_of(ii))
        M, sl = d.block(d.l2_product, ii, ii), slice(ii * d.n, (ii + 1) * d.n)
        w = BU_R[sl] - d.Div[ii] @ ur
        val = w @ (M @ w) + d.local_eta_rf_squared[ii] - F_R[sl] @ (M @ F_R[sl])
        val *= (1.0 / np.pi ** 2) / d.min_diffusion_evs[ii] * d.subdomain_diameters[ii] ** 2
        assert abs(val - r1[ii, k]) < 1e-10 * abs(val)
