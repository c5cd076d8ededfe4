"""Data functions: stand-ins for the dune.xt.functions factories the problem files call
(reference OS2015_academic_problem.py:6-11, thermalblock_problem.py:5-10).

A function is evaluated on the host at quadrature points only (coefficient sampling; the samples
are what the HIP assembly kernels consume).  Call signature used by both the product and the
oracle: ``fn(x, centers, keys)`` with ``x [..., npts, 2]`` points, ``centers [..., 2]`` the centres
of the elements the points belong to (decides the cell of piecewise-constant functions) and
``keys [..., 3]`` the canonical element key (global coarse cx, cy, triangle t).
"""
import numpy as np

_SAFE = {name: getattr(np, name) for name in ('cos', 'sin', 'exp', 'sqrt', 'tan', 'log', 'abs', 'pi')}
_SAFE['pow'] = np.power


def _per_element(v, x):
    """Broadcast per-element values to the point array: centres / keys come either per element ([..., 2|3] against
    x [..., npts, 2]) or already repeated per point (same leading shape as x)."""
    shape = np.asarray(x).shape[:-1]
    v = np.asarray(v)
    if v.ndim == len(shape) - 1:
        v = v[..., None]
    return np.broadcast_to(v, shape)


class ExpressionFunction:
    """``make_expression_function_1x1(grid, 'x', expr, order=, name=)``: numpy-evaluated expression in x[0], x[1]."""

    def __init__(self, variable, expression, order=2, name='expression'):
        assert variable == 'x'
        self.expression, self.order, self.name = expression, order, name
        self._code = compile(expression.replace('^', '**'), '<expression {}>'.format(name), 'eval')

    def __call__(self, x, centers=None, keys=None):
        x = np.asarray(x, dtype=np.float64)
        env = dict(_SAFE)
        env['x'] = (x[..., 0], x[..., 1])
        out = eval(self._code, {'__builtins__': {}}, env)
        return np.broadcast_to(np.asarray(out, dtype=np.float64), x.shape[:-1])


class ConstantFunction:
    order = 0

    def __init__(self, value, name='constant'):
        self.value, self.name = np.asarray(value, dtype=np.float64), name

    def __call__(self, x, centers=None, keys=None):
        return np.broadcast_to(self.value, np.asarray(x).shape[:-1])


class CheckerboardFunction:
    """``make_checkerboard_function_1x1``: piecewise constant on a Cartesian ``num_elements`` checkerboard,
    ``values[ix + nx * iy]``; the cell is decided by the element centre."""
    order = 0

    def __init__(self, lower_left, upper_right, num_elements, values, name='checkerboard'):
        self.ll = np.asarray(lower_left, dtype=np.float64)
        self.ur = np.asarray(upper_right, dtype=np.float64)
        self.num = (int(num_elements[0]), int(num_elements[1]))
        self.values = np.asarray(values, dtype=np.float64).reshape(-1)
        self.name = name

    def __call__(self, x, centers, keys=None):
        c = np.asarray(centers, dtype=np.float64)
        ix = np.clip(np.floor(self.num[0] * (c[..., 0] - self.ll[0]) / (self.ur[0] - self.ll[0])).astype(np.int64),
                     0, self.num[0] - 1)
        iy = np.clip(np.floor(self.num[1] * (c[..., 1] - self.ll[1]) / (self.ur[1] - self.ll[1])).astype(np.int64),
                     0, self.num[1] - 1)
        v = self.values[ix + self.num[0] * iy]
        return _per_element(v, x)


class ElementwiseFunction:
    """Piecewise constant per fine element: ``table[cy, cx, t]`` indexed by the canonical element key
    (synthetic multiscale coefficient fields)."""
    order = 0

    def __init__(self, table, name='elementwise'):
        self.table, self.name = np.asarray(table, dtype=np.float64), name

    def __call__(self, x, centers, keys):
        k = np.asarray(keys)
        v = self.table[k[..., 1], k[..., 0], k[..., 2]]
        return _per_element(v, x)


class SumFunction:
    def __init__(self, functions, coefficients, name='sum'):
        self.functions, self.coefficients, self.name = list(functions), [float(c) for c in coefficients], name
        self.order = max(getattr(f, 'order', 2) for f in self.functions)

    def __call__(self, x, centers, keys):
        return sum(c * f(x, centers, keys) for f, c in zip(self.functions, self.coefficients))


def make_expression_function_1x1(grid, variable, expression, order=2, name='expression'):
    return ExpressionFunction(variable, expression, order=order, name=name)


def make_constant_function_1x1(grid, value, name='constant'):
    return ConstantFunction(value, name=name)


def make_constant_function_2x2(grid, value, name='constant'):
    return ConstantFunction(np.asarray(value, dtype=np.float64).reshape(2, 2), name=name)


def make_checkerboard_function_1x1(grid_provider, lower_left, upper_right, num_elements, values, name='checkerboard'):
    return CheckerboardFunction(lower_left, upper_right, num_elements, values, name=name)
