"""Quadrature rules of the oracle -- TEST INFRASTRUCTURE ONLY.

dune-gdt integrates every local integrand with ``QuadratureRules<D, dim>::rule(type, integrand order +
over_integrate)`` (the ``over_integrate`` arguments are in the reference tree:
discretize_elliptic_block_swipdg.py:247,267,327,347,369,405,519,655,660,782; the integrand orders are
[UPSTREAM-RECALL] of dune-gdt ``local/integrands/*.hh``).  dune-geometry answers such a request

* on a triangle with its table of symmetric rules, by requested order p:
  1 -> centroid; 2 -> 3 interior points (1/6, 1/6, 2/3); 3 -> 4 points (Strang-Fix, negative centre weight);
  4 -> 6 points (Dunavant, degree 4); 5 -> 7 points (Radon, degree 5); 6 and 7 -> 12 points (Gatermann,
  degree 7, rotational symmetry only); 8 -> 16 points (Dunavant, degree 8)
  ([UPSTREAM-RECALL] dune-geometry ``quadraturerules/simplexquadrature.cc``);
* on an edge with the Gauss-Legendre rule of ceil((p + 1) / 2) points.

The abscissae below were polished against the moment equations to machine precision
(every rule integrates all monomials up to its degree with residual <= 6e-17).

``QuadratureSpec`` names one requested order per integrand of the path; ``QuadratureSpec.uniform()`` is the
round-1 convention (one degree-5 rule everywhere), ``QuadratureSpec.dune(...)`` the per-integrand orders of
dune-gdt.  PARITY: pinned by the 12-digit estimate of python/scripts/online_adaptive_lrbms.py:49 through
tests/test_reference_pin.py.
"""
import itertools

import numpy as np


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
    """(barycentric points [k, 3], weights [k] summing to 1) dune-geometry hands out for a requested order."""
    order = int(order)
    if order not in _TRI:
        raise ValueError('no triangle rule tabulated for order {}'.format(order))
    return _TRI[order]


def edge_rule(order):
    """Gauss-Legendre on [0, 1] with ceil((order + 1) / 2) points: (t [k], w [k] summing to 1)."""
    npts = max(1, (int(order) + 2) // 2)
    x, w = np.polynomial.legendre.leggauss(npts)
    return 0.5 * (x + 1.0), 0.5 * w


class QuadratureSpec:
    """Requested quadrature order per integrand of the hot path."""

    FIELDS = ('system_volume', 'system_inner_face', 'system_coupling_face', 'system_boundary_face', 'rhs', 'f2',
              'energy_volume', 'energy_face', 'elliptic_bar', 'flux_face', 'df_aa', 'df_ab', 'df_bb', 'ceps')

    def __init__(self, **orders):
        missing = set(self.FIELDS) - set(orders)
        extra = set(orders) - set(self.FIELDS)
        assert not missing and not extra, (missing, extra)
        for k, v in orders.items():
            setattr(self, k, int(v))

    def with_(self, **orders):
        d = {k: getattr(self, k) for k in self.FIELDS}
        d.update(orders)
        return QuadratureSpec(**d)

    def as_dict(self):
        return {k: getattr(self, k) for k in self.FIELDS}

    @classmethod
    def uniform(cls, order=5):
        """Round-1 convention: the degree-5 triangle rule and the 3-point Gauss rule for every integrand."""
        return cls(**{k: order for k in cls.FIELDS})

    @classmethod
    def dune(cls, lambda_order=2, f_order=2, lambda_bar_order=2, lambda_hat_order=2, kappa_order=0, p=1,
             flux_over_integrate=0):
        """Orders dune-gdt requests for P``p`` DG ([UPSTREAM-RECALL] ``order()`` of the local integrands) plus the
        ``over_integrate`` arguments the reference passes:

        * elliptic volume integrand: lambda + kappa + 2 (p - 1); system ``over_integrate=2`` (block_swipdg.py:405),
          products 0 (:655), ``make_local_elliptic_matrix_operator`` default 0 (:685);
        * IPDG face integrands: lambda + kappa + 2 p; inner faces of a subdomain inherit the operator's
          ``over_integrate=2`` (:405), the coupling and boundary operators are built without one (:409,:426) -> 0;
          penalty product 0 (:660);
        * L2 functional: f + p, ``over_integrate=2`` (:519); ``apply_l2_product(f, f, over_integrate=2)`` (:782): 2 f + 2;
        * df products ``over_integrate=2`` (:327,:347,:369): aa lambda_hat + 2 lambda + 3 kappa + 2 (p - 1);
          ab lambda_hat + lambda + 2 kappa + (p - 1) + 1 (RT0); bb lambda_hat + kappa + 2;
        * flux reconstruction: the inner IPDG integrand with the constant test function: lambda + kappa + p
          (+ ``flux_over_integrate``; the reference passes none, :165);
        * min eigenvalue of lambda_hat kappa (:776): sampled at the points of a rule of order lambda_hat + kappa.
        """
        L, F, LB, LH, K = lambda_order, f_order, lambda_bar_order, lambda_hat_order, kappa_order
        return cls(system_volume=L + K + 2 * (p - 1) + 2,
                   system_inner_face=L + K + 2 * p + 2,
                   system_coupling_face=L + K + 2 * p,
                   system_boundary_face=L + K + 2 * p,
                   rhs=F + p + 2,
                   f2=2 * F + 2,
                   energy_volume=L + K + 2 * (p - 1),
                   energy_face=L + K + 2 * p,
                   elliptic_bar=LB + K + 2 * (p - 1),
                   flux_face=L + K + p + flux_over_integrate,
                   df_aa=LH + 2 * L + 3 * K + 2 * (p - 1) + 2,
                   df_ab=LH + L + 2 * K + (p - 1) + 1 + 2,
                   df_bb=LH + K + 2 + 2,
                   ceps=LH + K)


# the round-1 names, kept for callers that want "the" uniform rule
TRI_BARY, TRI_W = triangle_rule(5)
EDGE_T, EDGE_W = edge_rule(5)
