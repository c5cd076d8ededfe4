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
        assert np.array_equal(lam[:, :7], d.lam_vol[q])
        lm, lp = d.lam_face[q]
        assert np.array_equal(lam[Em][np.arange(len(Em))[:, None], 7 + 3 * fm[:, None] + np.arange(3)[None, :]], lm)
        assert np.array_equal(lam[Ep][np.arange(len(Ep))[:, None], 7 + 3 * fp[:, None] + (2 - np.arange(3))[None, :]],
                              lp[has])
    f = sample_function(p['f'], x, c, k, volume_only=True).reshape(-1, 7)
    assert np.array_equal(f, d.f_vol)


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
