"""The 3D / P2 path through its API shim (discretize -> LRBMSReductor3D.reduce -> rd.solve / rd.estimate, d.estimate) against the
CPU oracle: the call sequence of the reference's driver (python/scripts/online_adaptive_lrbms.py:56-141) on BASELINE.json
config 5's problem family.  PARITY UNPINNED beyond the oracle (no 3D reference counterpart)."""
import numpy as np
import pytest

import common3d as c3

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', ['aniso_2x2x1', 'interior_3x3x3'])
def test_driver_sequence_matches_the_oracle(name):
    from pylrbms_amd.discretize_elliptic_block_swipdg_3d import LRBMSReductor3D, discretize
    p = c3.make_problem(name)
    o = c3.oracle_of(p)
    pd = {'grid': p['grid'], 'lambda': {'functions': p['lambdas'], 'coefficients': p['thetas']}, 'lambda_bar': p['lambda_bar'],
          'lambda_hat': p['lambda_hat'], 'f': p['f'], 'mu_bar': p['mu_bar'], 'mu_hat': p['mu_hat']}
    d, data = discretize(pd)
    V = c3.make_bases3d(o.S, o.n, p['N'], seed=11)
    red = LRBMSReductor3D(d, V)
    rd = red.reduce()
    mu = p['mu']
    u, (it, res) = rd.solve(mu, rtol=1e-13, return_info=True)
    ord_ = c3.reduce_with_oracle(p, o, V)
    uo = ord_.solve(mu)
    assert c3.rel(u.cpu().numpy(), np.stack(uo)) < 1e-10
    eta, (nc, r, df), ind = rd.estimate(u, mu, decompose=True)
    eta_o, (nco, ro, dfo), _ = ord_.estimate(uo, mu, decompose=True)
    assert abs(eta - eta_o) < 1e-9 * eta_o
    for a, b in ((nc, nco), (r, ro), (df, dfo)):
        assert c3.rel(a, b) < 1e-8
    # the reduced estimate is the full-order estimate of the reconstruction ...
    U = red.reconstruct(u)
    eta_f = d.estimate(U, mu)
    assert abs(eta_f - eta) < 1e-8 * eta
    # ... and the full-order estimate of an arbitrary DG function matches the oracle's
    W = np.random.default_rng(2).standard_normal((o.S, o.n))
    assert abs(d.estimate(W, mu) - o.estimate(W.ravel(), mu)) < 1e-9 * o.estimate(W.ravel(), mu)
    # block operator = global operator
    y = d.apply(d.engine.ctx.from_numpy(W[:, :, None]), mu).cpu().numpy()
    assert c3.rel(y.ravel(), o.system_matrix(mu) @ W.ravel()) < 1e-11
    # Galerkin orthogonality of the reduced solution
    res_ = o.b - o.system_matrix(mu) @ U.cpu().numpy().ravel()
    assert max(np.abs(V[ii].T @ res_[o.dofs_of(ii)]).max() for ii in range(o.S)) < 1e-9


def test_snapshots_bases_reduced_model_workflow():
    """The offline / online sequence of the reference's driver (online_adaptive_lrbms.py:56-141) in 3D: full-order snapshots
    (d.solve, checked against the oracle's sparse LU) -> local bases (constant + restricted snapshots) -> reduce -> reduced
    solve at a new parameter -> the reduced solution is the Galerkin projection: it reproduces a snapshot parameter exactly and
    its estimate is the full-order estimate of its reconstruction."""
    import torch
    from pylrbms_amd.discretize_elliptic_block_swipdg_3d import LRBMSReductor3D, discretize
    p = c3.make_problem('aniso_2x2x1')
    o = c3.oracle_of(p)
    pd = {'grid': p['grid'], 'lambda': {'functions': p['lambdas'], 'coefficients': p['thetas']}, 'lambda_bar': p['lambda_bar'],
          'lambda_hat': p['lambda_hat'], 'f': p['f'], 'mu_bar': p['mu_bar'], 'mu_hat': p['mu_hat']}
    d, _ = discretize(pd)
    snaps = []
    for mu in (0.2, 0.9):
        U, (it, res) = d.solve(mu, rtol=1e-12, return_info=True)
        assert it > 0 and res <= 1e-12
        assert c3.rel(U.cpu().numpy().ravel(), o.solve(mu)) < 1e-9
        snaps.append(U)
    V = torch.stack([torch.ones_like(snaps[0])] + snaps, dim=2)                # [S, n, 3]
    V = torch.linalg.qr(V)[0].contiguous()
    red = LRBMSReductor3D(d, V)
    rd = red.reduce()
    u = rd.solve(0.9, rtol=1e-13)
    assert c3.rel(red.reconstruct(u).cpu().numpy().ravel(), o.solve(0.9)) < 1e-8          # a snapshot parameter is reproduced
    mu = 0.5
    u = rd.solve(mu, rtol=1e-13)
    Ur = red.reconstruct(u)
    err = Ur.cpu().numpy().ravel() - o.solve(mu)
    assert np.sqrt(o.energy_norm2(err, mu)) < 0.05 * np.sqrt(o.energy_norm2(o.solve(mu), mu))
    assert abs(rd.estimate(u, mu) - d.estimate(Ur, mu)) < 1e-8 * rd.estimate(u, mu)


@pytest.mark.parametrize('name', ['aniso_2x2x1', 'kc_3x1x2'])
def test_reference_reductor_surface_in_3d(name):
    """``LRBMSReductor(d, products=..., order=0)`` + ``extend_basis(d.solve(mu))`` + ``reduce()`` (reference reductor.py:17-73, driver
    online_adaptive_lrbms.py:105-123) in 3D: the local energy product is assembled on the device (lrbms3_assemble_energy_product,
    checked entry by entry against the oracle in tests/test_parity3d_gpu.py), Gram-Schmidt runs on the device, the bases are
    orthonormal in that product (V^T P V = I to 1e-9 against the ORACLE's product matrix), and the reduced model is the Galerkin
    model on the same span as the oracle's reductor fed with the same snapshots (same reconstruction, same estimate)."""
    import torch
    from pylrbms_amd.discretize_elliptic_block_swipdg_3d import ExtensionError3D, LRBMSReductor3D, discretize
    p = c3.make_problem(name)
    o = c3.oracle_of(p)
    pd = {'grid': p['grid'], 'lambda': {'functions': p['lambdas'], 'coefficients': p['thetas']}, 'lambda_bar': p['lambda_bar'],
          'lambda_hat': p['lambda_hat'], 'f': p['f'], 'mu_bar': p['mu_bar'], 'mu_hat': p['mu_hat']}
    d, _ = discretize(pd)
    red = LRBMSReductor3D(d, products=None, order=0)                      # reductor.py:29-31: the constant starts every basis
    assert red.local_sizes() == [1] * o.S
    dflt = LRBMSReductor3D(d)                                              # reductor.py:23-24: neither bases nor order => order 0
    assert dflt.local_sizes() == [1] * o.S and torch.equal(dflt.bases, red.bases)
    snaps = []
    for mu in (0.2, 0.9):
        U = d.solve(mu, rtol=1e-12)
        snaps.append(U.cpu().numpy())
        red.extend_basis(U)                                                # online_adaptive_lrbms.py:117-121
    assert red.local_sizes() == [3] * o.S and red.basis_size() == 3
    with pytest.raises(ExtensionError3D):
        red.extend_basis(snaps[0])                                         # already in the span
    assert red.local_sizes() == [3] * o.S
    # orthonormal in the local energy product -- of the oracle
    Vh = red.bases.cpu().numpy()
    Pm = o.P.tocsr()
    for ii in range(o.S):
        dofs = o.dofs_of(ii)
        G = Vh[ii].T @ (Pm[dofs][:, dofs] @ Vh[ii])
        assert np.abs(G - np.eye(3)).max() < 1e-9
        span = np.stack([np.ones(o.n)] + [sn[ii] for sn in snaps], axis=1)
        coef = np.linalg.lstsq(span, Vh[ii], rcond=None)[0]
        assert np.abs(span @ coef - Vh[ii]).max() < 1e-9 * np.abs(Vh[ii]).max()       # the same span
    assert float((red.gram() - torch.eye(3, dtype=torch.float64, device='cuda')[None]).abs().max()) < 1e-9
    rd = red.reduce()
    # the oracle's reductor on the un-orthonormalised span: same Galerkin solution (reconstruction) and the same estimate
    bases_o = [np.stack([np.ones(o.n)] + [sn[ii] for sn in snaps], axis=1) for ii in range(o.S)]
    from oracle.lrbms3d import Reductor3D
    ored = Reductor3D(o, bases_o)
    ord_ = ored.reduce()
    for mu in (0.5, 0.9):
        u = rd.solve(mu, rtol=1e-13)
        rec = red.reconstruct(u).cpu().numpy().ravel()
        uo = ord_.solve(mu)
        ref = ored.reconstruct(uo)
        assert c3.rel(rec, ref) < 1e-8
        assert abs(rd.estimate(u, mu) - ord_.estimate(uo, mu)) < 1e-7 * ord_.estimate(uo, mu)
        assert c3.rel(red.reconstruct_local(u, 1).cpu().numpy(), ref[o.dofs_of(1)]) < 1e-8
    # one subdomain alone (reductor.py:31,78): a new local vector, the others keep their size
    w = np.random.default_rng(4).standard_normal(o.n)
    red.extend_basis_local(1, w)
    assert red.local_sizes() == [3, 4] + [3] * (o.S - 2) and red.basis_size() == 4
    G = red.gram().cpu().numpy()
    want = np.stack([np.diag([1.0] * nl + [0.0] * (4 - nl)) for nl in red.local_sizes()])
    assert np.abs(G - want).max() < 1e-9                                   # zero-padded columns stay zero
    rd2 = red.reduce()                                                     # ragged bases through the uniform kernels
    u2 = rd2.solve(0.5, rtol=1e-13)
    assert float(u2[[i for i in range(o.S) if i != 1], 3].abs().max()) == 0.0
    bases_o[1] = np.concatenate([bases_o[1], w[:, None]], axis=1)
    ored2 = Reductor3D(o, bases_o)
    assert c3.rel(red.reconstruct(u2).cpu().numpy().ravel(), ored2.reconstruct(ored2.reduce().solve(0.5))) < 1e-8
    # order = 1: the constant and the three coordinate functions, orthonormalised
    red1 = LRBMSReductor3D(d, order=1)
    assert red1.local_sizes() == [4] * o.S
    assert float((red1.gram() - torch.eye(4, dtype=torch.float64, device='cuda')[None]).abs().max()) < 1e-9
    with pytest.raises(NotImplementedError):
        red.enrich_local(0, None, mu=0.5)


@pytest.mark.parametrize('name', ['interior_3x3x3', 'kc_3x1x2'])
def test_full_order_solver_with_every_coarse_space_matches_the_sparse_lu(name):
    """lrbms3_fom_solve with its two-level preconditioner: no coarse level, subdomain constants (nc = 1), P1 per subdomain
    (nc = 4, the default after mesh upload) -- the same solution as the oracle's sparse LU each time, fewer iterations with
    the richer space; a rank-deficient coarse space (a zero column) falls back to the element blocks alone."""
    from pylrbms_amd.engine3d import Engine3D
    p = c3.make_problem(name)
    o = c3.oracle_of(p)
    eng = Engine3D(p['grid'], p['lambdas'], p['f'], p['lambda_bar'], p['lambda_hat']).assemble()
    th = c3.theta_of(p, p['mu'])
    want = o.solve(p['mu'])
    n = eng.t.n
    its = {}
    U, info = eng.ctx.fom_solve(eng.Q, th, eng.ops['A_diag'], eng.ops['A_cpl'], eng.ops['b'], rtol=1e-12)     # default: P1
    assert c3.rel(U.cpu().numpy().ravel(), want) < 1e-9
    its['P1'] = info[0]
    for key, Phi in (('none', None), ('constants', np.ones((n, 1))), ('deficient', np.zeros((n, 2)))):
        eng.ctx.fom_coarse_space(Phi)
        U, info = eng.ctx.fom_solve(eng.Q, th, eng.ops['A_diag'], eng.ops['A_cpl'], eng.ops['b'], rtol=1e-12)
        assert c3.rel(U.cpu().numpy().ravel(), want) < 1e-9, key
        its[key] = info[0]
    assert its['P1'] <= its['constants'] <= its['none'] == its['deficient'] and its['P1'] < its['none'], its   # (counted in steps of 16)
    # a kept coarse inverse (built at another parameter) is still a valid preconditioner: same solution
    x = np.asarray(eng.t.node_coordinates())
    ext = x.max(axis=0) - x.min(axis=0)
    eng.ctx.fom_coarse_space(np.concatenate([np.ones((n, 1)), (x - 0.5 * (x.max(axis=0) + x.min(axis=0))) / ext], axis=1))
    eng.ctx.fom_precond_keep(True)
    try:
        eng.ctx.fom_solve(eng.Q, c3.theta_of(p, 0.2), eng.ops['A_diag'], eng.ops['A_cpl'], eng.ops['b'], rtol=1e-12)      # builds and keeps
        U, info = eng.ctx.fom_solve(eng.Q, th, eng.ops['A_diag'], eng.ops['A_cpl'], eng.ops['b'], rtol=1e-12)             # reuses
    finally:
        eng.ctx.fom_precond_keep(False)
    assert c3.rel(U.cpu().numpy().ravel(), want) < 1e-9 and info[0] <= its['none']


def test_batched_online_phase_through_the_api():
    from pylrbms_amd.discretize_elliptic_block_swipdg_3d import LRBMSReductor3D, discretize
    p = c3.make_problem('aniso_2x2x1')
    pd = {'grid': p['grid'], 'lambda': {'functions': p['lambdas'], 'coefficients': p['thetas']}, 'lambda_bar': p['lambda_bar'],
          'lambda_hat': p['lambda_hat'], 'f': p['f'], 'mu_bar': p['mu_bar'], 'mu_hat': p['mu_hat']}
    d, _ = discretize(pd)
    rd = LRBMSReductor3D(d, c3.make_bases3d(d.engine.S, d.engine.t.n, 6, seed=1)).reduce()
    mus = [0.1 + 0.05 * k for k in range(19)]                          # two native batches (16 + 3)
    U = rd.solve_batch(mus, rtol=1e-13)
    etas = rd.estimate_batch(U, mus)
    for k in (0, 7, 18):
        u = rd.solve(mus[k], rtol=1e-13)
        assert c3.rel(U[k].cpu().numpy(), u.cpu().numpy()) < 1e-10
        assert abs(etas[k] - rd.estimate(u, mus[k])) < 1e-10 * etas[k]


def test_experimental_orders_of_convergence_of_the_product():
    """Convergence study of the 3D / P2 PRODUCT (the validation the reference's OS2015_convergence_study.py performs in 2D):
    full-order solves on 2^3, 3^3, 4^3 cubes of one subdomain for u = sin(pi x) sin(pi y) sin(pi z); L2 / H1 errors (evaluated with
    the oracle's quadrature on the host) converge with orders 3 / 2, the estimate decreases and equals the oracle's."""
    from oracle.lrbms3d import Discretization3D
    from oracle.mesh3d import KuhnMesh3D
    from pylrbms_amd.discretize_elliptic_block_swipdg_3d import discretize
    from pylrbms_amd.grid3d import make_grid3d
    pi = np.pi
    one = lambda x: 1.0 + 0.0 * x[..., 0]                                                           # noqa: E731
    uex = lambda x: np.sin(pi * x[..., 0]) * np.sin(pi * x[..., 1]) * np.sin(pi * x[..., 2])        # noqa: E731
    f = lambda x: 3 * pi * pi * uex(x)                                                              # noqa: E731
    rows = []
    for K in (2, 3, 4):
        grid = make_grid3d(num_subdomains=(1, 1, 1), cubes_per_subdomain_and_dim=K)
        pd = {'grid': grid, 'lambda': {'functions': [one], 'coefficients': [lambda mu: 1.0]}, 'lambda_bar': one, 'lambda_hat': one,
              'f': f, 'mu_bar': 1.0, 'mu_hat': 1.0, 'data_degree': 4}
        d, _ = discretize(pd)
        U = d.solve(1.0, rtol=1e-12)
        m = KuhnMesh3D([K] * 3, [1, 1, 1])
        o = Discretization3D(m, [one], [lambda mu: 1.0], np.eye(3), f, one, one, 1.0, 1.0, data_degree=4)
        u = U.cpu().numpy().ravel()
        assert c3.rel(u, o.solve(1.0)) < 1e-8
        x, w, phi, grad = o._vol_points(8)
        Ue = u.reshape(m.num_elements, 10)
        uh, gh = np.einsum('ki,ei->ek', phi, Ue), np.einsum('ekia,ei->eka', grad, Ue)
        c, s_ = np.cos(pi * x), np.sin(pi * x)
        ge = pi * np.stack([c[..., 0] * s_[..., 1] * s_[..., 2], s_[..., 0] * c[..., 1] * s_[..., 2], s_[..., 0] * s_[..., 1] * c[..., 2]], -1)
        l2 = np.sqrt(np.einsum('k,e,ek->', w, m.volume, (uh - uex(x)) ** 2))
        h1 = np.sqrt(np.einsum('k,e,eka->', w, m.volume, (gh - ge) ** 2))
        eta = d.estimate(U, 1.0)
        assert abs(eta - o.estimate(u, 1.0)) < 1e-8 * eta
        rows.append((K, l2, h1, eta))
    (k0, a0, b0, e0), (k1, a1, b1, e1) = rows[1], rows[2]
    rate = lambda x0, x1: np.log(x0 / x1) / np.log(k1 / k0)                                         # noqa: E731
    assert rate(a0, a1) > 2.8 and rate(b0, b1) > 1.8
    assert e1 < e0 < rows[0][3]


def test_contexts_of_one_process_share_the_side_streams():
    """HIP maps streams onto few hardware queues, so every context of the process (2D and 3D) uses the same library-owned side
    streams (csrc/lrbms_dev.h): two 2D contexts report the same handles, and a 3D pass gives bit-identical results whether or not
    other contexts are alive and after they are gone (the pool is reference-counted)."""
    import torch
    from pylrbms_amd._native import NativeContext
    from pylrbms_amd.engine3d import Engine3D
    p = c3.make_problem('aniso_2x2x1')
    o = c3.oracle_of(p)
    V = c3.make_bases3d(o.S, o.n, p['N'], seed=5)

    def one_pass():
        eng = Engine3D(p['grid'], p['lambdas'], p['f'], p['lambda_bar'], p['lambda_hat']).assemble()
        out = eng.project_and_estimate(eng.ctx.from_numpy(V))
        torch.cuda.synchronize()
        res = {k: v.cpu().numpy().copy() for k, v in out.items()}
        eng.ctx.close()
        return res
    alone = one_pass()
    a, b = NativeContext(0), NativeContext(0)
    try:
        for i in range(3):
            ha, hb = a.lib.lrbms_ctx_aux_stream(a.handle, i), b.lib.lrbms_ctx_aux_stream(b.handle, i)
            assert ha and ha == hb
        beside = one_pass()
    finally:
        a.close()
        b.close()
    after = one_pass()
    for k in alone:
        assert np.array_equal(alone[k], beside[k]) and np.array_equal(alone[k], after[k]), k
