"""Quadrature of the offline assembly: which rule every integrand of the path is integrated with.

dune-gdt integrates each local integrand with ``QuadratureRules::rule(type, integrand order + over_integrate)``; the
``over_integrate`` arguments are in the reference tree (discretize_elliptic_block_swipdg.py:247,267,327,347,369,405,519,
655,660,782), the integrand orders follow the ``order()`` methods of dune-gdt's local integrands, and dune-geometry
answers a request for order p with the symmetric rules tabulated below (triangle) / the Gauss-Legendre rule of
ceil((p + 1) / 2) points (edge).  The host samples the data functions at the points of these rules; the kernels of
csrc/assemble.hip receive the rules (``lrbms_quadrature``) and the samples and do the arithmetic.

``QuadratureSpec.dune(...)`` (default of ``Engine``): the reference's orders, from the declared ``order`` of the data
functions.  ``QuadratureSpec.uniform(5)``: the round-1 convention (7-point / 3-point rule for everything).
"""
import ctypes
import itertools

import numpy as np

MAXQV, MAXQF = 16, 4


def _orbit3(a):
    b = 1.0 - 2.0 * a
    return [(b, a, a), (a, b, a), (a, a, b)]


def _orbit6(a, b):
    c = 1.0 - a - b
    return sorted(set(itertools.permutations((a, b, c))))


def _orbit_rot(a, b):
    c = 1.0 - a - b
    return [(a, b, c), (b, c, a), (c, a, b)]


def _assemble(parts):
    pts, w = [], []
    for orbit, weight in parts:
        pts += orbit
        w += [weight] * len(orbit)
    return np.array(pts, dtype=np.float64), np.array(w, dtype=np.float64)


_C = [(1.0 / 3.0, 1.0 / 3.0, 1.0 / 3.0)]
_s15 = np.sqrt(15.0)
# requested order -> (barycentric points, weights summing to 1); abscissae polished against the moment equations
_TRI = {
    1: _assemble([(_C, 1.0)]),
    2: _assemble([(_orbit3(1.0 / 6.0), 1.0 / 3.0)]),
    3: _assemble([(_C, -27.0 / 48.0), (_orbit3(0.2), 25.0 / 48.0)]),
    4: _assemble([(_orbit3(0.09157621350977078), 0.10995174365532198),
                  (_orbit3(0.44594849091596483), 0.22338158967801136)]),
    5: _assemble([(_C, 0.225),
                  (_orbit3((6.0 + _s15) / 21.0), (155.0 + _s15) / 1200.0),
                  (_orbit3((6.0 - _s15) / 21.0), (155.0 - _s15) / 1200.0)]),
    7: _assemble([(_orbit_rot(0.06238226509440212, 0.06751786707391609), 0.0530340563148725),
                  (_orbit_rot(0.05522545665692661, 0.3215024938519818), 0.08776281742889211),
                  (_orbit_rot(0.03432430294509715, 0.6609491961867356), 0.05755008556996317),
                  (_orbit_rot(0.5158423343535917, 0.2777161669763918), 0.13498637401960553)]),
    8: _assemble([(_C, 0.14431560767770213),
                  (_orbit3(0.4592925882926677), 0.09509163426733876),
                  (_orbit3(0.17056930775169654), 0.10321737053473663),
                  (_orbit3(0.050547228317034024), 0.03245849762320667),
                  (_orbit6(0.008394777409865857, 0.26311282963483285), 0.027230314174408615)]),
}
_TRI[6] = _TRI[7]
_TRI[0] = _TRI[1]


def triangle_rule(order):
    order = int(order)
    if order not in _TRI:
        raise ValueError('no triangle rule tabulated for order {}'.format(order))
    return _TRI[order]


def edge_rule(order):
    """Gauss-Legendre on [0, 1] with ceil((order + 1) / 2) points (ascending, symmetric about 1/2)."""
    npts = max(1, (int(order) + 2) // 2)
    if npts > MAXQF:
        raise ValueError('edge rule of order {} needs more than {} points'.format(order, MAXQF))
    x, w = np.polynomial.legendre.leggauss(npts)
    return 0.5 * (x + 1.0), 0.5 * w


class QuadratureSpec:
    """Requested quadrature order per integrand (same fields and formulas as the oracle's spec; tests compare them)."""

    FIELDS = ('system_volume', 'system_inner_face', 'system_coupling_face', 'system_boundary_face', 'rhs', 'f2',
              'energy_volume', 'energy_face', 'elliptic_bar', 'flux_face', 'df_aa', 'df_ab', 'df_bb', 'ceps')

    def __init__(self, **orders):
        missing, extra = set(self.FIELDS) - set(orders), set(orders) - set(self.FIELDS)
        assert not missing and not extra, (missing, extra)
        for k, v in orders.items():
            setattr(self, k, int(v))
        if self.system_coupling_face != self.system_boundary_face:
            raise ValueError('coupling and Dirichlet boundary faces must share one rule (whether a face of the subdomain '
                             'boundary is one or the other depends on the subdomain, the sample layout does not)')

    def as_dict(self):
        return {k: getattr(self, k) for k in self.FIELDS}

    def with_(self, **orders):
        d = self.as_dict()
        d.update(orders)
        return QuadratureSpec(**d)

    @classmethod
    def uniform(cls, order=5):
        return cls(**{k: order for k in cls.FIELDS})

    @classmethod
    def dune(cls, lambda_order=2, f_order=2, lambda_bar_order=2, lambda_hat_order=2, kappa_order=0, p=1):
        L, F, LB, LH, K = lambda_order, f_order, lambda_bar_order, lambda_hat_order, kappa_order
        return cls(system_volume=L + K + 2 * (p - 1) + 2,          # block_swipdg.py:405 over_integrate=2
                   system_inner_face=L + K + 2 * p + 2,            # inner faces inherit the operator's over_integrate
                   system_coupling_face=L + K + 2 * p,             # :409, :426: built without over_integrate
                   system_boundary_face=L + K + 2 * p,
                   rhs=F + p + 2,                                  # :519
                   f2=2 * F + 2,                                   # :782
                   energy_volume=L + K + 2 * (p - 1),              # :655 over_integrate=0
                   energy_face=L + K + 2 * p,                      # :660
                   elliptic_bar=LB + K + 2 * (p - 1),              # :685
                   flux_face=L + K + p,                            # :165 (no over_integrate)
                   df_aa=LH + 2 * L + 3 * K + 2 * (p - 1) + 2,     # :327
                   df_ab=LH + L + 2 * K + (p - 1) + 1 + 2,         # :369
                   df_bb=LH + K + 2 + 2,                           # :347
                   ceps=LH + K)                                    # :776

    @classmethod
    def for_problem(cls, lambda_funcs, f, lambda_bar, lambda_hat):
        """The reference's orders for a problem: the declared polynomial order of each data function (expression
        functions: their ``order=`` argument; checkerboard / constant / elementwise functions: 0)."""
        order = lambda fn: int(getattr(fn, 'order', 2))  # noqa: E731
        return cls.dune(lambda_order=max(order(fn) for fn in lambda_funcs), f_order=order(f),
                        lambda_bar_order=order(lambda_bar), lambda_hat_order=order(lambda_hat))


# ---------------------------------------------------------------------------------------------------- native mirror
class TriRule(ctypes.Structure):
    _fields_ = [('n', ctypes.c_int32), ('pad', ctypes.c_int32), ('w', ctypes.c_double * MAXQV), ('b', (ctypes.c_double * 3) * MAXQV)]


class EdgeRule(ctypes.Structure):
    _fields_ = [('n', ctypes.c_int32), ('pad', ctypes.c_int32), ('w', ctypes.c_double * MAXQF), ('t', ctypes.c_double * MAXQF)]


TRI_FIELDS = ('system_volume', 'energy_volume', 'elliptic_bar', 'rhs', 'f2', 'ceps', 'df_aa', 'df_ab', 'df_bb')
EDGE_FIELDS = ('system_inner_face', 'system_coupling_face', 'energy_face', 'flux_face')


class NativeQuadrature(ctypes.Structure):
    """``lrbms_quadrature`` of include/lrbms_hip.h: the rules plus the layout of the sample records derived from them."""
    _fields_ = [(k, TriRule) for k in TRI_FIELDS] + [(k, EdgeRule) for k in EDGE_FIELDS] + \
               [(k, ctypes.c_int32) for k in ('nfs', 'o_sysv', 'o_sysf', 'o_enf', 'o_flf', 'o_env', 'lam_stride',
                                              'o_aa', 'o_ab', 'lamdf_stride', 'o_haa', 'o_hab', 'o_hbb', 'o_hceps', 'lhat_stride',
                                              'o_frhs', 'o_ff2', 'f_stride', 'lbar_stride', 'pad_')]


def _tri(order):
    b, w = triangle_rule(order)
    if len(w) > MAXQV:
        raise ValueError('triangle rule of order {} has more than {} points'.format(order, MAXQV))
    r = TriRule()
    r.n = len(w)
    for k in range(len(w)):
        r.w[k] = float(w[k])
        for v in range(3):
            r.b[k][v] = float(b[k, v])
    return r


def _edge(order):
    t, w = edge_rule(order)
    r = EdgeRule()
    r.n = len(w)
    for k in range(len(w)):
        r.w[k], r.t[k] = float(w[k]), float(t[k])
    return r


def native_quadrature(spec):
    """``NativeQuadrature`` for a spec.  Sample records (all fp64, per element):

    * lambda_q [Q][S_ext][n_T][lam_stride]:  system volume points | 3 faces x nfs system face points (face f: the inner
      rule if the template has a neighbour element across it, else the coupling / boundary rule; parametrised from local
      vertex f + 1 to f + 2) | 3 x energy face points | 3 x flux face points | energy volume points
    * lambda_q (df) [Q][S][n_T][lamdf_stride]:  df_aa points | df_ab points
    * lambda_hat [S][n_T][lhat_stride]:  df_aa | df_ab | df_bb | ceps points
    * f [S][n_T][f_stride]:  rhs points | f2 points;       lambda_bar [S][n_T][lbar_stride]: elliptic_bar points"""
    q = NativeQuadrature()
    for k in TRI_FIELDS:
        setattr(q, k, _tri(getattr(spec, k)))
    for k in EDGE_FIELDS:
        setattr(q, k, _edge(getattr(spec, k)))
    n = lambda k: getattr(q, k).n  # noqa: E731
    q.nfs = max(n('system_inner_face'), n('system_coupling_face'))
    q.o_sysv = 0
    q.o_sysf = q.o_sysv + n('system_volume')
    q.o_enf = q.o_sysf + 3 * q.nfs
    q.o_flf = q.o_enf + 3 * n('energy_face')
    q.o_env = q.o_flf + 3 * n('flux_face')
    q.lam_stride = q.o_env + n('energy_volume')
    q.o_aa, q.o_ab = 0, n('df_aa')
    q.lamdf_stride = n('df_aa') + n('df_ab')
    q.o_haa, q.o_hab = 0, n('df_aa')
    q.o_hbb = q.o_hab + n('df_ab')
    q.o_hceps = q.o_hbb + n('df_bb')
    q.lhat_stride = q.o_hceps + n('ceps')
    q.o_frhs, q.o_ff2 = 0, n('rhs')
    q.f_stride = n('rhs') + n('f2')
    q.lbar_stride = n('elliptic_bar')
    return q
