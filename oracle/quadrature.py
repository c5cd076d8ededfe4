"""Quadrature rules fixed by the oracle -- TEST INFRASTRUCTURE ONLY.

dune-gdt picks Gauss rules of order ``integrand order + over_integrate``
(discretize_elliptic_block_swipdg.py:405,519,655).  Which exact
rule that yields is not determinable from the reference tree, so the oracle
fixes one rule per entity type and uses it for every integrand:

* triangle: 7-point Radon/Dunavant rule, exact to degree 5
  (P1 x expression-of-order-2 x over_integrate=2 = degree 5, the highest the path asks for);
* edge: 3-point Gauss-Legendre, exact to degree 5.

PARITY UNPINNED (see oracle/__init__.py).
"""
import numpy as np

_s15 = np.sqrt(15.0)
_b1 = (6.0 + _s15) / 21.0
_b2 = (6.0 - _s15) / 21.0
_w1 = (155.0 + _s15) / 1200.0
_w2 = (155.0 - _s15) / 1200.0

# barycentric coordinates (l0, l1, l2) and weights (sum to 1)
TRI_BARY = np.array([
    [1.0 / 3.0, 1.0 / 3.0, 1.0 / 3.0],
    [1.0 - 2.0 * _b1, _b1, _b1],
    [_b1, 1.0 - 2.0 * _b1, _b1],
    [_b1, _b1, 1.0 - 2.0 * _b1],
    [1.0 - 2.0 * _b2, _b2, _b2],
    [_b2, 1.0 - 2.0 * _b2, _b2],
    [_b2, _b2, 1.0 - 2.0 * _b2],
])
TRI_W = np.array([0.225, _w1, _w1, _w1, _w2, _w2, _w2])

_g = 0.5 * np.sqrt(3.0 / 5.0)
EDGE_T = np.array([0.5 - _g, 0.5, 0.5 + _g])
EDGE_W = np.array([5.0 / 18.0, 8.0 / 18.0, 5.0 / 18.0])
