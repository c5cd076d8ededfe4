"""Synthetic multiscale diffusion problem of SURVEY.md section 8(d) (BASELINE.json configs 2-4).

Omega = [0,1]^2, Sx x Sy subdomains of k_c x k_c coarse squares (8 triangles each), kappa = I, Q = 2:
lambda_0 = per-fine-element lognormal field exp(sigma g), g ~ N(0,1) i.i.d., rng(20240531);
lambda_1 = channel indicator (horizontal strips in the pattern of the reference's
artificial_channels_problem.py:38-42, tiled per 4x4 subdomains); theta = (1, mu), mu in [0.1, 1],
mu_bar = mu_hat = 1  =>  lambda_bar = lambda_hat = lambda_0 + lambda_1;
f = pi^2/2 cos(pi/2 x) cos(pi/2 y) (OS2015_academic_problem.py:47).
"""
import numpy as np

from pylrbms_amd.functions import (ElementwiseFunction, SumFunction, make_constant_function_2x2,
                                   make_expression_function_1x1)
from pylrbms_amd.grid import make_boundary_info, make_multiscale_grid
from pylrbms_amd.parameters import ExpressionParameterFunctional

SEED = 20240531


def channel_table(Kx, Ky, tile):
    """Indicator of horizontal channels: inside every tile of ``tile`` coarse rows, rows at 1/4 and 5/8 of the
    tile height carry a channel spanning the middle 3/4 of the tile width."""
    table = np.zeros((Ky, Kx, 8))
    cy = np.arange(Ky)[:, None]
    cx = np.arange(Kx)[None, :]
    ry, rx = cy % tile, cx % tile
    rows = (ry == tile // 4) | (ry == (5 * tile) // 8)
    cols = (rx >= tile // 8) & (rx < tile - tile // 8)
    table[(rows & cols)] = 1.0
    return table


def init_grid_and_problem(config, mu_bar=1, mu_hat=1, mpi_comm=None):
    Px, Py = config['num_subdomains']
    kc = config.get('coarse_per_subdomain', 4)
    sigma = config.get('lognormal_sigma', 1.0)
    grid = make_multiscale_grid((Px, Py), kc, mpi_comm=mpi_comm)
    Kx, Ky = grid.K
    rng = np.random.default_rng(config.get('seed', SEED))
    lognormal = np.exp(sigma * rng.standard_normal((Ky, Kx, 8)))
    channels = channel_table(Kx, Ky, 4 * kc)
    lambda_0 = ElementwiseFunction(lognormal, name='lambda_0')
    lambda_1 = ElementwiseFunction(channels, name='lambda_1')
    parameter_type = {'diffusion': (1,)}
    coefficients = [ExpressionParameterFunctional('1.', parameter_type),
                    ExpressionParameterFunctional('diffusion', parameter_type)]
    kappa = make_constant_function_2x2(grid, [[1., 0.], [0., 1.]], name='kappa')
    f = make_expression_function_1x1(grid, 'x', '0.5*pi*pi*cos(0.5*pi*x[0])*cos(0.5*pi*x[1])', order=2, name='f')
    lambda_bar = SumFunction([lambda_0, lambda_1], [1.0, mu_bar], name='lambda_bar')
    lambda_hat = SumFunction([lambda_0, lambda_1], [1.0, mu_hat], name='lambda_hat')
    return {'grid': grid,
            'mpi_comm': mpi_comm,
            'boundary_info': make_boundary_info(grid, {'type': 'xt.grid.boundaryinfo.alldirichlet'}),
            'inner_boundary_id': grid.inner_boundary_segment_index,
            'lambda': {'functions': [lambda_0, lambda_1], 'coefficients': coefficients},
            'lambda_bar': lambda_bar,
            'lambda_hat': lambda_hat,
            'kappa': kappa,
            'f': f,
            'parameter_type': parameter_type,
            'mu_bar': (mu_bar,),
            'mu_hat': (mu_hat,),
            'mu_min': (min(0.1, mu_bar, mu_hat),),
            'mu_max': (max(1, mu_bar, mu_hat),),
            'parameter_range': (min(0.1, mu_bar, mu_hat), max(1, mu_bar, mu_hat))}
