"""Minimal stand-ins for the pyMOR parameter classes the path touches
(reference OS2015_academic_problem.py:14,42-44; thermalblock_problem.py:16,47-50;
discretize_elliptic_block_swipdg.py:59-60,757,809)."""
import numpy as np


class Parameter(dict):
    """pyMOR ``Parameter``: dict name -> ndarray."""

    def __hash__(self):
        return hash(tuple((k, tuple(np.ravel(v))) for k, v in sorted(self.items())))


def parse_parameter(mu, parameter_type):
    """``d.parse_parameter(mu)`` (used at online_adaptive_lrbms.py:90,134)."""
    if isinstance(mu, Parameter):
        return mu
    if isinstance(mu, dict):
        return Parameter({k: np.asarray(v, dtype=np.float64).reshape(parameter_type[k]) for k, v in mu.items()})
    if not parameter_type:
        return Parameter()
    flat = np.atleast_1d(np.asarray(list(mu) if not np.isscalar(mu) else [mu], dtype=np.float64)).ravel()
    out, pos = Parameter(), 0
    for k in sorted(parameter_type):
        shape = parameter_type[k]
        size = int(np.prod(shape)) if len(shape) else 1
        out[k] = flat[pos:pos + size].reshape(shape)
        pos += size
    assert pos == len(flat), 'parameter has wrong size'
    return out


class ParameterFunctional:
    def evaluate(self, mu=None):
        raise NotImplementedError

    def __call__(self, mu=None):
        return self.evaluate(mu)


class ExpressionParameterFunctional(ParameterFunctional):
    def __init__(self, expression, parameter_type, name=None):
        self.expression, self.parameter_type, self.name = expression, dict(parameter_type), name
        self._code = compile(expression, '<functional>', 'eval')

    def evaluate(self, mu=None):
        mu = parse_parameter(mu, self.parameter_type)
        env = {k: (v.item() if np.size(v) == 1 else v) for k, v in mu.items()}
        env.update({'sin': np.sin, 'cos': np.cos, 'exp': np.exp, 'sqrt': np.sqrt, 'pi': np.pi})
        return float(eval(self._code, {'__builtins__': {}}, env))


class ProjectionParameterFunctional(ParameterFunctional):
    def __init__(self, component_name, component_shape, coordinates=(), name=None):
        self.component_name, self.component_shape = component_name, tuple(component_shape)
        self.coordinates, self.name = tuple(coordinates), name
        self.parameter_type = {component_name: self.component_shape}

    def evaluate(self, mu=None):
        mu = parse_parameter(mu, self.parameter_type)
        return float(np.asarray(mu[self.component_name])[self.coordinates])


class ConstantParameterFunctional(ParameterFunctional):
    def __init__(self, value):
        self.value, self.parameter_type = float(value), {}

    def evaluate(self, mu=None):
        return self.value


class ProductParameterFunctional(ParameterFunctional):
    def __init__(self, factors):
        self.factors = list(factors)

    def evaluate(self, mu=None):
        out = 1.0
        for f in self.factors:
            out *= f.evaluate(mu) if hasattr(f, 'evaluate') else float(f)
        return out


class CubicParameterSpace:
    def __init__(self, parameter_type, minimum, maximum):
        self.parameter_type, self.minimum, self.maximum = dict(parameter_type), float(minimum), float(maximum)

    def _size(self):
        return int(sum(int(np.prod(s)) if len(s) else 1 for s in self.parameter_type.values()))

    def sample_uniformly(self, counts):
        axes = [np.linspace(self.minimum, self.maximum, counts) for _ in range(self._size())]
        grid = np.stack(np.meshgrid(*axes, indexing='ij'), axis=-1).reshape(-1, self._size())
        return [parse_parameter(row, self.parameter_type) for row in grid]

    def sample_randomly(self, count, seed=None):
        rng = np.random.default_rng(seed)
        vals = rng.uniform(self.minimum, self.maximum, size=(count, self._size()))
        return [parse_parameter(row, self.parameter_type) for row in vals]
