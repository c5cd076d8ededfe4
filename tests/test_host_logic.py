"""Host-side logic that runs without a GPU: coefficient sampling, parameter functionals, problem dicts."""
import numpy as np
import pytest

from common import oracle_from_problem
from pylrbms_amd import OS2015_academic_problem, multiscale_problem, thermalblock_problem
from pylrbms_amd.engine import sample_function, sample_points
from pylrbms_amd.parameters import CubicParameterSpace, ExpressionParameterFunctional, parse_parameter


@pytest.mark.parametrize('mk', [
    lambda: OS2015_academic_problem.init_grid_and_problem({'num_subdomains': [2, 2],
                                                           'half_num_fine_elements_per_subdomain_and_dim': 4}),
    lambda: thermalblock_problem.init_grid_and_problem({'num_subdomains': [2, 2],
                                                        'half_num_fine_elements_per_subdomain_and_dim': 4}),
    lambda: multiscale_problem.init_grid_and_problem({'num_subdomains': [3, 2], 'coarse_per_subdomain': 2}),
])
def test_samples_equal_the_oracles(mk):
    p = mk()
    g = p['grid']
    x, c, k = sample_points(g, list(range(g.num_subdomains)))
    d = oracle_from_problem(p)
    m = d.mesh
    Em, fm = m.face_minus[:, 0], m.face_minus[:, 1]
    has = m.face_plus[:, 0] >= 0
    Ep, fp = m.face_plus[has, 0], m.face_plus[has, 1]
    for q, fn in enumerate(p['lambda']['functions']):
        lam = sample_function(fn, x, c, k).reshape(-1, 16)
        assert np.array_equal(lam[:, :7], d._vol(fn, 5))
        lm, lp = d._face_sides(fn, 5)
        assert np.array_equal(lam[Em][np.arange(len(Em))[:, None], 7 + 3 * fm[:, None] + np.arange(3)[None, :]], lm)
        assert np.array_equal(lam[Ep][np.arange(len(Ep))[:, None], 7 + 3 * fp[:, None] + (2 - np.arange(3))[None, :]],
                              lp[has])
    f = sample_function(p['f'], x, c, k, volume_only=True).reshape(-1, 7)
    assert np.array_equal(f, d._vol(p['f'], 5))


def test_problem_dict_keys_match_the_reference():
    p = OS2015_academic_problem.init_grid_and_problem({'num_subdomains': [2, 2],
                                                       'half_num_fine_elements_per_subdomain_and_dim': 4})
    for key in ('grid', 'boundary_info', 'inner_boundary_id', 'lambda', 'lambda_bar', 'lambda_hat', 'kappa', 'f',
                'parameter_type', 'mu_bar', 'mu_hat', 'mu_min', 'mu_max', 'parameter_range'):
        assert key in p                                    # OS2015_academic_problem.py:52-67
    assert p['inner_boundary_id'] == 2 ** 64 - 43
    assert p['parameter_range'] == (0.1, 1)
    tb = thermalblock_problem.init_grid_and_problem({'num_subdomains': [2, 2],
                                                     'half_num_fine_elements_per_subdomain_and_dim': 4})
    assert [c.evaluate((1, 2, 3, 4)) for c in tb['lambda']['coefficients']] == [3.0, 1.0, 4.0, 2.0]


def test_parameters():
    pt = {'diffusion': (1,)}
    f = ExpressionParameterFunctional('diffusion', pt)
    assert f.evaluate(0.25) == 0.25 and f.evaluate((0.25,)) == 0.25 and f.evaluate({'diffusion': [0.25]}) == 0.25
    mu = parse_parameter(0.5, pt)
    assert mu['diffusion'].shape == (1,)
    space = CubicParameterSpace(pt, 0.1, 1.0)
    assert len(space.sample_uniformly(3)) == 3
    assert all(0.1 <= m['diffusion'][0] <= 1.0 for m in space.sample_randomly(5, seed=1))
    with pytest.raises(AssertionError):
        parse_parameter((1, 2), pt)


def _dof_points(g):
    t = g.template
    org = np.stack([g.subdomain_origin(i) for i in range(g.num_subdomains)])
    return (org[:, None, None, :] + t.points[None]).reshape(-1, 2)


@pytest.mark.parametrize('pc,kc,pf,kf', [([2, 3], 2, [2, 3], 4), ([2, 3], 2, [4, 6], 2), ([1, 1], 3, [3, 3], 4)])
def test_prolongation_reproduces_p1_functions_on_nested_grids(pc, kc, pf, kf):
    """EOC harness (reference EOC.py:300-314 ``prolong``): convex weights, exact for globally linear functions, and an
    arbitrary P1-DG function is reproduced at the fine vertices (checked by evaluating the parent's local P1 function)."""
    from pylrbms_amd.EOC import prolongation_map
    from pylrbms_amd.grid import make_multiscale_grid
    gc, gf = make_multiscale_grid(pc, kc), make_multiscale_grid(pf, kf)
    idx, w = prolongation_map(gc, gf)
    assert idx.shape == (3 * gf.num_elements, 3) and np.abs(w.sum(axis=1) - 1).max() < 1e-14 and w.min() > -1e-14
    assert (idx // 3 == idx[:, :1] // 3).all()                     # all three parents are DoFs of ONE coarse element
    lin = lambda x: 0.3 + 1.7 * x[:, 0] - 2.2 * x[:, 1]  # noqa: E731
    assert np.abs((lin(_dof_points(gc))[idx] * w).sum(axis=1) - lin(_dof_points(gf))).max() < 1e-13
    # a discontinuous coarse function: one random P1 function per coarse element
    rng = np.random.default_rng(3)
    coef = rng.standard_normal((gc.num_elements, 3))               # a + b x + c y per coarse element
    xc, xf = _dof_points(gc), _dof_points(gf)
    Uc = coef[np.arange(3 * gc.num_elements) // 3, 0] + (coef[np.arange(3 * gc.num_elements) // 3, 1:] * xc).sum(axis=1)
    par = idx[:, 0] // 3
    Uf = coef[par, 0] + (coef[par, 1:] * xf).sum(axis=1)
    assert np.abs((Uc[idx] * w).sum(axis=1) - Uf).max() < 1e-12


def test_eoc_table_logic(capsys):
    """``EocStudy.run``: rates from consecutive levels, efficiency = norm / estimate, 'inf' for a vanishing quantity."""
    from pylrbms_amd.EOC import EocStudy

    class Fake(EocStudy):
        level_info_title, accuracies, norms, indicators = 'lvl', ('h',), ('err',), ('zero',)
        estimates, max_levels = (('eta', 'err'),), 2

        def __init__(self):
            self.data = {}

        def solve(self, level):
            pass

        def level_info(self, level):
            return str(level)

        def accuracy(self, level, id):
            return 0.5 ** level

        def compute_norm(self, level, id):
            return 3.0 * (0.5 ** level) ** 2

        def compute_indicator(self, level, id):
            return 0.0

        def compute_estimate(self, level, id):
            return 6.0 * (0.5 ** level)

    data = Fake().run()
    out = capsys.readouterr().out.splitlines()
    assert len(out) == 5 and out[2].split('|')[3].strip() == '----'
    cells = [c.strip() for c in out[4].split('|')]
    assert cells[3] == '2.00' and cells[5] == 'inf' and cells[8] == '1.00'     # EOC(err), EOC(zero), EOC(eta)
    assert abs(float(cells[7]) - data[2]['norm']['err'] / data[2]['estimate']['eta']) < 0.01
    only = Fake()
    only.run(only_these=('h', 'err'))
    assert 'eta' not in capsys.readouterr().out


def test_vtk_writer(tmp_path):
    from pylrbms_amd.grid import make_multiscale_grid
    from pylrbms_amd.visualize import write_vtk
    g = make_multiscale_grid([2, 2], 2)
    t = g.template
    vals = np.arange(4 * t.n, dtype=np.float64).reshape(4, t.n)
    fn = write_vtk(str(tmp_path / 'u'), g, range(4), {'u': vals})
    lines = open(fn).read().splitlines()
    assert lines[0].startswith('# vtk DataFile') and lines[3] == 'DATASET UNSTRUCTURED_GRID'
    npts = 4 * t.n
    assert 'POINTS {} double'.format(npts) in lines and 'CELLS {} {}'.format(npts // 3, 4 * npts // 3) in lines
    i = lines.index('SCALARS u double 1')
    assert np.allclose(np.array(lines[i + 2:i + 2 + npts], dtype=float), vals.reshape(-1))
    assert g.visualize(str(tmp_path / 'grid')).endswith('grid.vtk')
