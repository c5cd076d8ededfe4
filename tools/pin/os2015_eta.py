"""A/B table of the conventions the reference tree leaves open, against the 12-digit estimates the reference
holds (python/scripts/online_adaptive_lrbms.py:49-53, repeated in mpi_elliptic.py:30-34).

The comment block there reads ``[4, 4], 2, [2, 2], 4: 0.815510144764`` -- the tuple is the older config schema still
used by python/scripts/OS2015_convergence_study.py:29-32: num_coarse_grid_elements, num_grid_refinements,
num_grid_subdomains, num_grid_oversampling_layers.  ``[4, 4], 2, [2, 2]`` is exactly what ``make_grid`` builds today
for ``num_subdomains=[2, 2], half_num_fine_elements_per_subdomain_and_dim=4`` (grid.py:24-27); ``[6, 6], 4, [6, 6]``
is a 6 x 6 cube grid refined four times = the 8-triangle pattern on a 12 x 12 cube grid (K = [12, 12], P = [6, 6]).

CPU only (oracle).  ``python tools/pin/os2015_eta.py > profiles/r02_pin_table.txt``."""
import os
import sys

import numpy as np
import scipy.sparse.linalg as spla

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from oracle.lrbms import OracleDiscretization  # noqa: E402
from oracle.mesh import OracleMesh  # noqa: E402
from oracle.quadrature import QuadratureSpec  # noqa: E402

REF = {'os2015_4_2_2': 0.815510144764, 'os2015_6_4_6': 3.03372753518, 'local_thermalblock_6_4_6': 0.585792065793}


def _cos(x):
    return np.cos(0.5 * np.pi * x[..., 0]) * np.cos(0.5 * np.pi * x[..., 1])


def os2015(K, P, quad, **kw):
    """OS2015_academic_problem.py:19-67 with mu_bar = mu_hat = 1."""
    mesh = OracleMesh([-1, -1], [1, 1], K, P)
    th = [lambda mu: 1.0, lambda mu: float(np.ravel(mu)[0])]
    one = lambda x, c, k: 1.0 + 0.0 * _cos(x)  # noqa: E731
    return OracleDiscretization(mesh, [lambda x, c, k: 1 + _cos(x), lambda x, c, k: -_cos(x)], th, np.eye(2),
                                lambda x, c, k: 0.5 * np.pi ** 2 * _cos(x), one, one, 1.0, 1.0, quad=quad, **kw)


def _checkerboard(values):
    values = np.asarray(values, dtype=np.float64)

    def fn(x, c, k):
        c = np.asarray(c)
        ix = np.clip(np.floor(6 * (c[..., 0] + 1) / 2).astype(int), 0, 5)
        iy = np.clip(np.floor(6 * (c[..., 1] + 1) / 2).astype(int), 0, 5)
        v = values[ix + 6 * iy]
        if v.ndim == np.asarray(x).ndim - 2:
            v = v[..., None]
        return np.broadcast_to(v, np.asarray(x).shape[:-1])
    return fn


def local_thermalblock(K, P, quad, **kw):
    """local_thermalblock_problem.py:23-72: 6 x 6 checkerboard, inclusions in cells 7 and 25, mu_bar = mu_hat = 0."""
    def values(bg, fg):
        v = [bg] * 36
        for i in (7, 25):
            v[i] = fg
        return v
    mesh = OracleMesh([-1, -1], [1, 1], K, P)
    th = [lambda mu: 1.0, lambda mu: 1.1 + np.sin(float(np.ravel(mu)[0]))]
    return OracleDiscretization(mesh, [_checkerboard(values(1., 0.)), _checkerboard(values(0., 1.))], th, np.eye(2),
                                lambda x, c, k: 0.5 * np.pi ** 2 * _cos(x), _checkerboard(values(1., 1.1)),
                                _checkerboard(values(1., 1.1)), 0.0, 0.0, quad=quad, **kw)


def eta(d, mu, sqrt_local, U=None):
    U = d.solve(mu) if U is None else U
    e, (nc, r, df), _ = d.estimate(U, mu, decompose=True, sqrt_local=sqrt_local)
    return e, np.linalg.norm(nc), np.linalg.norm(r), np.linalg.norm(df)


def line(tag, e, ref):
    print('{:66s} eta={:.12f} rel={:+.3e}  |nc|={:.6f} |r|={:.6f} |df|={:.6f}'.format(tag, e[0], e[0] / ref - 1, *e[1:]))


def main():
    ref = REF['os2015_4_2_2']
    K, P = [4, 4], [2, 2]
    print('# reference: OS2015 [4,4],2,[2,2],4 = {!r} (online_adaptive_lrbms.py:49); sqrt variant of the local indicators,\n'
          '# alpha as written (estimators.py:114-121), mu = parameter_range[0] = 0.1 (the maximum over linspace(0.1, 1, 3))'.format(ref))
    print('\n## 1. quadrature presets x Oswald conventions x mu (sqrt variant unless stated)')
    for name, q in (('uniform5 (round 1)', QuadratureSpec.uniform(5)), ('dune orders', QuadratureSpec.dune()),
                    ('dune orders, flux over_integrate 2', QuadratureSpec.dune(flux_over_integrate=2))):
        for kw in ({}, {'oswald_patch': 'vertex'}, {'oswald_zero_on': 'subdomain'}, {'oswald_zero_on': 'none'},
                   {'accumulate_coupling_across_q': True}):
            d = os2015(K, P, q, **kw)
            for mu in (0.1, 0.55, 1.0):
                line('{} {} mu={}'.format(name, kw, mu), eta(d, mu, True), ref)
            line('{} {} mu=0.1 NO sqrt (HEAD)'.format(name, kw), eta(d, 0.1, False), ref)
    print('\n## 2. one integrand at a time away from the dune orders (mu = 0.1)')
    base = QuadratureSpec.dune()
    print('# base:', base.as_dict())
    for f in QuadratureSpec.FIELDS:
        for o in (1, 2, 3, 4, 5, 7, 8):
            if o != getattr(base, f):
                line('{} -> order {} (dune: {})'.format(f, o, getattr(base, f)), eta(os2015(K, P, base.with_(**{f: o})), 0.1, True), ref)
    print('\n## 3. one uniform order for everything (mu = 0.1)')
    for o in (2, 3, 4, 5, 7, 8):
        line('uniform order {}'.format(o), eta(os2015(K, P, QuadratureSpec.uniform(o)), 0.1, True), ref)
    print('\n## 4. inexact full-order solve (reference: bicgstab.ilut, precision 1e-6, online_adaptive_lrbms.py:71)')
    d = os2015(K, P, base)
    A, b = d.assemble_global(0.1), d.b
    for drop in (1e-2, 1e-4):
        ilu = spla.spilu(A.tocsc(), drop_tol=drop, fill_factor=10)
        M = spla.LinearOperator(A.shape, ilu.solve)
        for tol in (1e-4, 1e-6, 1e-8):
            u, _ = spla.bicgstab(A, b, rtol=tol, atol=0, M=M, maxiter=400)
            res = np.linalg.norm(A @ u - b) / np.linalg.norm(b)
            line('bicgstab + ilu(drop {:g}) rtol {:g}: residual {:.1e}'.format(drop, tol, res),
                 eta(d, 0.1, True, U=u.reshape(d.S, d.n)), ref)
    print('\n## 5. mu that would reproduce the value (none is natural)')
    for mu in (0.099, 0.1, 0.101, 0.102, 0.103):
        line('mu = {}'.format(mu), eta(d, mu, True), ref)
    for key, mk, mus, tag in (('os2015_6_4_6', os2015, (0.1, 0.55, 1.0), 'OS2015'),
                              ('local_thermalblock_6_4_6', local_thermalblock, (0.0, 0.5 * np.pi, np.pi), 'local thermalblock')):
        ref2 = REF[key]
        print('\n## 6. {} [6,6],4,[6,6],4 = {!r} -- NOT reproduced by any reading'.format(tag, ref2))
        q = QuadratureSpec.dune() if mk is os2015 else QuadratureSpec.dune(0, 2, 0, 0)
        for K2 in ([12, 12], [6, 6]):
            for kw in ({}, {'oswald_patch': 'vertex'}, {'accumulate_coupling_across_q': True},
                       {'accumulate_coupling_across_q': True, 'oswald_patch': 'vertex'}):
                d2 = mk(K2, [6, 6], q, **kw)
                for mu in mus:
                    for sq in (True, False):
                        line('K={} {} mu={:.3f} sqrt={:d}'.format(K2, kw, mu, sq), eta(d2, mu, sq), ref2)


if __name__ == '__main__':
    main()
