"""Synthetic 3D multiscale diffusion problem of BASELINE.json config 5 (3D diffusion, 8 x 8 x 8 subdomains, SWIPDG p = 2).

Omega = [0,1]^3, P^3 subdomains of k_c^3 cubes (six tetrahedra each), kappa = I, Q = 2:
lambda_0 = exp(sigma sin(2 pi w x) cos(2 pi w y) sin(2 pi w z + 1))  (oscillating at the scale of the subdomains, w = P / 2),
lambda_1 = smooth channels along x (bumps in y, z around the channel axes of the 2D problem, tiled per 4 subdomains);
theta = (1, mu), mu in [0.1, 1], mu_bar = mu_hat = 1  =>  lambda_bar = lambda_hat = lambda_0 + lambda_1;
f = 3/4 pi^2 cos(pi/2 x) cos(pi/2 y) cos(pi/2 z) (the 3D analogue of OS2015_academic_problem.py:47).
The data are smooth, so both sides of a face see the same coefficient; declared polynomial order 2 as for the reference's
expression functions (OS2015_academic_problem.py:39-47)."""
import numpy as np

from pylrbms_amd.grid3d import make_grid3d


def init_grid_and_problem(config, mu_bar=1.0, mu_hat=1.0, rank=0, world_size=1):
    P = tuple(config['num_subdomains'])
    kc = config.get('cubes_per_subdomain', 4)
    sigma = config.get('sigma', 0.5)
    grid = make_grid3d(num_subdomains=P, cubes_per_subdomain_and_dim=kc, rank=rank, world_size=world_size)
    w = max(P) / 2.0
    tp = 2.0 * np.pi

    def lambda_0(x):
        return np.exp(sigma * np.sin(tp * w * x[..., 0]) * np.cos(tp * w * x[..., 1]) * np.sin(tp * w * x[..., 2] + 1.0))

    def lambda_1(x):
        per = 4.0 / max(P)
        yy, zz = np.mod(x[..., 1], per) / per, np.mod(x[..., 2], per) / per
        bump = lambda t, c: np.exp(-((t - c) / 0.06) ** 2)                      # noqa: E731
        return (bump(yy, 0.25) + bump(yy, 0.625)) * (bump(zz, 0.25) + bump(zz, 0.625))

    def f(x):
        return 0.75 * np.pi ** 2 * np.cos(0.5 * np.pi * x[..., 0]) * np.cos(0.5 * np.pi * x[..., 1]) * np.cos(0.5 * np.pi * x[..., 2])

    return {'grid': grid,
            'lambda': {'functions': [lambda_0, lambda_1], 'coefficients': [lambda mu: 1.0, lambda mu: float(mu)]},
            'lambda_bar': lambda x: lambda_0(x) + mu_bar * lambda_1(x),
            'lambda_hat': lambda x: lambda_0(x) + mu_hat * lambda_1(x),
            'kappa': np.eye(3), 'f': f, 'data_degree': config.get('data_degree', 2),
            'mu_bar': mu_bar, 'mu_hat': mu_hat, 'parameter_range': (0.1, 1.0)}
