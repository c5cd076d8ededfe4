"""The 3D / P2 oracle (BASELINE.json config 5: 3D diffusion, SWIPDG p = 2) -- CPU, no GPU.

There is no reference counterpart for this configuration (the reference binds the 2D P1 operators only), hence no golden
values: PARITY UNPINNED.  The oracle is validated by properties a correct discretization must have."""
import numpy as np
import pytest
import scipy.sparse.linalg as spla

from oracle.lrbms3d import Discretization3D, Reductor3D, p2_basis, tet_rule, tri_rule
from oracle.mesh3d import KuhnMesh3D


def _one(x):
    return 1.0 + 0.0 * x[..., 0]


def _lam1(x):
    return x[..., 0] * x[..., 1] + 0.5


def _problem(K, P, f=None, **kw):
    m = KuhnMesh3D(K, P)
    lb = lambda x: 1.0 + 0.5 * _lam1(x)                      # noqa: E731   lambda at mu_bar = mu_hat = 0.5
    return Discretization3D(m, [_one, _lam1], [lambda mu: 1.0, lambda mu: mu], np.eye(3),
                            f if f is not None else (lambda x: 1.0 + x[..., 2]), lb, lb, 0.5, 0.5, **kw)


def test_config5_template_sizes_and_neighbourhoods():
    m = KuhnMesh3D([4, 4, 4], [1, 1, 1])                     # one subdomain of config 5: k_c = 4
    assert m.elements_per_subdomain == 384 and 10 * m.elements_per_subdomain == 3840
    assert abs(m.volume.sum() - 1.0) < 1e-14
    assert abs(m.face_area[m.face_is_boundary].sum() - 6.0) < 1e-13
    assert np.all(np.bincount(m.elem_face.ravel())[~m.face_is_boundary] == 2)          # conforming
    m = KuhnMesh3D([3, 3, 3], [3, 3, 3])
    assert len(m.neighborhood_of(13)) == 7 and len(m.neighborhood_of(0)) == 4
    assert m.neighborhood_of(13) == [4, 10, 12, 13, 14, 16, 22]


def test_quadrature_and_basis():
    for deg in range(1, 9):
        b, w = tet_rule(deg)
        assert abs(w.sum() - 1.0) < 1e-14
        # int_T l1^a l2^b l3^c = 6 a! b! c! / (a + b + c + 3)!  (normalised by the volume 1/6)
        from math import factorial as fa
        for a, bb, c in ((deg, 0, 0), (deg // 2, deg - deg // 2, 0), (deg // 3, deg // 3, deg - 2 * (deg // 3))):
            exact = 6.0 * fa(a) * fa(bb) * fa(c) / fa(a + bb + c + 3)
            assert abs((w * b[:, 1] ** a * b[:, 2] ** bb * b[:, 3] ** c).sum() - exact) < 1e-14
        t, wt = tri_rule(deg)
        exact = 2.0 * fa(deg) / fa(deg + 2)
        assert abs((wt * t[:, 1] ** deg).sum() - exact) < 1e-14
    lam = np.random.default_rng(0).dirichlet(np.ones(4), size=7)
    phi, dphi = p2_basis(lam)
    assert np.abs(phi.sum(axis=1) - 1.0).max() < 1e-14                                   # partition of unity
    nodes = np.vstack([np.eye(4), 0.5 * (np.eye(4)[[0, 0, 0, 1, 1, 2]] + np.eye(4)[[1, 2, 3, 2, 3, 3]])])
    assert np.abs(p2_basis(nodes)[0] - np.eye(10)).max() < 1e-14                         # Lagrange property


def test_system_is_symmetric_positive_definite():
    d = _problem([2, 2, 2], [2, 1, 1])
    A = d.system_matrix(0.7)
    assert abs(A - A.T).max() < 1e-13
    assert np.linalg.eigvalsh(A.toarray()).min() > 0
    assert abs(d.P - d.P.T).max() < 1e-13
    # the local energy products do not couple subdomains
    Pd = d.P.toarray()
    assert np.abs(Pd[:d.n, d.n:]).max() == 0.0


def test_quadratic_solutions_are_reproduced_and_the_flux_is_locally_conservative():
    mu = 0.7
    uex = lambda x: x[..., 0] ** 2 + 2 * x[..., 1] * x[..., 2] - x[..., 2] ** 2 + x[..., 0] + 1.0      # noqa: E731

    def fex(x):            # -div((1 + mu (x y + 1/2)) grad u), laplace u = 0
        gl = np.stack([mu * x[..., 1], mu * x[..., 0], 0 * x[..., 0]], -1)
        gu = np.stack([2 * x[..., 0] + 1, 2 * x[..., 2], 2 * x[..., 1] - 2 * x[..., 2]], -1)
        return -(gl * gu).sum(-1)
    d = _problem([2, 2, 2], [2, 2, 2], f=fex)
    u = spla.spsolve(d.system_matrix(mu).tocsc(), d.b + d.dirichlet_rhs(uex, mu))
    assert np.abs(u - d.interpolate(uex)).max() < 1e-12
    m = d.mesh
    r = d.flux_reconstruction(u, mu)
    div_int = (d.div * r[m.elem_face]).sum(axis=1) * m.volume                            # int_T div r
    interior = ~np.any(m.face_is_boundary[m.elem_face], axis=1)
    assert np.abs(div_int - d.bdiv)[interior].max() < 1e-12                              # = int_T f
    # RT0 basis: unit normal flux through its own face
    e = 5
    X = m.vertices[m.elements[e]]
    for f in range(4):
        rr = np.zeros(m.num_faces)
        rr[m.elem_face[e, f]] = 1.0
        lam = np.zeros((1, 4))
        lam[0, [k for k in range(4) if k != f]] = 1.0 / 3.0
        val = d.rt0_values(e, lam, rr)[0]
        nrm = m.face_normal[m.elem_face[e, f]]
        assert abs(val @ nrm - 1.0) < 1e-13


def test_oswald_interpolation_of_a_conforming_function_is_the_function():
    for patch in ('neighborhood', 'vertex'):
        d = _problem([2, 2, 2], [2, 2, 1], oswald_patch=patch)
        bubble = lambda x: np.prod(x * (1 - x), axis=-1)                                  # noqa: E731
        v = d.interpolate(bubble)
        assert np.abs(d.oswald_error(v)).max() < 1e-15
        nc, _, _ = d.local_terms(v, 0.5)
        assert np.abs(nc).max() < 1e-28


def test_experimental_orders_of_convergence():
    pi = np.pi
    uex = lambda x: np.sin(pi * x[..., 0]) * np.sin(pi * x[..., 1]) * np.sin(pi * x[..., 2])    # noqa: E731
    errs = []
    for K in (2, 3, 4):
        m = KuhnMesh3D([K] * 3, [1, 1, 1])
        d = Discretization3D(m, [_one], [lambda mu: 1.0], np.eye(3), lambda x: 3 * pi * pi * uex(x), _one, _one, 1.0, 1.0,
                             data_degree=4)
        u = d.solve(1.0)
        x, w, phi, grad = d._vol_points(8)
        U = u.reshape(m.num_elements, 10)
        uh, gh = np.einsum('ki,ei->ek', phi, U), np.einsum('ekia,ei->eka', grad, U)
        c, s_ = np.cos(pi * x), np.sin(pi * x)
        ge = pi * np.stack([c[..., 0] * s_[..., 1] * s_[..., 2], s_[..., 0] * c[..., 1] * s_[..., 2],
                            s_[..., 0] * s_[..., 1] * c[..., 2]], -1)
        l2 = np.sqrt(np.einsum('k,e,ek->', w, m.volume, (uh - uex(x)) ** 2))
        h1 = np.sqrt(np.einsum('k,e,eka->', w, m.volume, (gh - ge) ** 2))
        errs.append((K, l2, h1, d.estimate(u, 1.0)))
    (k0, a0, b0, e0), (k1, a1, b1, e1) = errs[1], errs[2]
    rate = lambda x0, x1: np.log(x0 / x1) / np.log(k1 / k0)                              # noqa: E731
    assert rate(a0, a1) > 2.8 and rate(b0, b1) > 1.8                                     # P2: L2 order 3, energy order 2
    assert e1 < e0 < errs[0][3]                                                          # the estimate decreases


def test_reduced_model_is_consistent_with_the_full_order_model():
    """Galerkin orthogonality of the reduced solution, and: the estimate from the projected operators equals the
    full-order estimate of the reconstruction (the defining property of the projection)."""
    d = _problem([2, 2, 2], [2, 2, 1])
    rng = np.random.default_rng(0)
    N = 4
    bases = [np.hstack([np.ones((d.n, 1)), rng.standard_normal((d.n, N - 1))]) for _ in range(d.S)]
    red = Reductor3D(d, bases)
    rd = red.reduce()
    mu = 0.3
    u = [rng.standard_normal(N) for _ in range(d.S)]
    for a, b in zip(rd.local_terms(u, mu), d.local_terms(red.reconstruct(u), mu)):
        assert np.abs(a - b).max() < 1e-12 * np.abs(b).max()
    assert abs(rd.estimate(u, mu) - d.estimate(red.reconstruct(u), mu)) < 1e-12 * d.estimate(red.reconstruct(u), mu)
    ur = rd.solve(mu)
    res = d.b - d.system_matrix(mu) @ red.reconstruct(ur)
    assert max(np.abs(bases[ii].T @ res[d.dofs_of(ii)]).max() for ii in range(d.S)) < 1e-12
    # with the full local spaces as bases the reduced solution is the full-order one
    full = Reductor3D(d, [np.eye(d.n) for _ in range(d.S)])
    uf = full.reduce().solve(mu)
    assert np.abs(np.concatenate(uf) - d.solve(mu)).max() < 1e-10


@pytest.mark.parametrize('name', ['aniso_2x2x1', 'q3_2x1x2'])
def test_oracle_reproduces_the_committed_3d_fixtures(name):
    """tests/golden/cfg5_*.npz (self-generated, tests/golden/make_golden.py: the reference has no 3D counterpart) pin the
    oracle against silent drift; the HIP path is checked against the same files in tests/test_parity3d_gpu.py."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    from make_golden import build3d
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'cfg5_' + name + '.npz'))
    out = build3d(name)
    for k in gold.files:
        ref = gold[k]
        assert np.abs(np.asarray(out[k]) - ref).max() <= 1e-10 * max(np.abs(ref).max(), 1e-300), k
