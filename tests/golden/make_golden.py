"""Generates tests/golden/*.npz with the CPU ORACLE (python tests/golden/make_golden.py).

The reference contributes no vectors for this path (its tests hold no numerical fixture and
its arithmetic cannot be imported here), so these fixtures are SELF-generated: they pin the oracle against silent
drift and give the HIP path stored expected outputs, nothing more.  Data only: inputs and expected outputs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from common import energy_orthonormalize, make_bases, oracle_from_problem  # noqa: E402
from oracle.lrbms import OracleReductor  # noqa: E402
from pylrbms_amd import OS2015_academic_problem, multiscale_problem, thermalblock_problem  # noqa: E402

CASES = {
    'os2015_2x2': (lambda: OS2015_academic_problem.init_grid_and_problem(
        {'num_subdomains': [2, 2], 'half_num_fine_elements_per_subdomain_and_dim': 4}), 3, 0.3),
    'thermalblock_2x2': (lambda: thermalblock_problem.init_grid_and_problem(
        {'num_subdomains': [2, 2], 'half_num_fine_elements_per_subdomain_and_dim': 4}), 2, (0.5, 1.0, 0.2, 0.8)),
    'multiscale_3x3': (lambda: multiscale_problem.init_grid_and_problem(
        {'num_subdomains': [3, 3], 'coarse_per_subdomain': 2}), 4, 0.4),
}


def build(name):
    mk, N, mu = CASES[name]
    p = mk()
    d = oracle_from_problem(p)
    V = energy_orthonormalize(make_bases(d.S, d.n, N, seed=7), d)
    red = OracleReductor(d, [V[ii] for ii in range(d.S)])
    rd = red.reduce()
    u = np.stack(rd.solve(mu))
    eta, (nc, r, df), ind = rd.estimate([u[ii] for ii in range(d.S)], mu, decompose=True)
    U = d.solve(mu)
    eta_f, (ncf, rf, dff), _ = d.estimate(U, mu, decompose=True)
    centre = d.S // 2
    out = {
        'N': N, 'mu': np.atleast_1d(np.asarray(mu, dtype=np.float64)), 'V': V,
        'A0_centre': d.block(d.A[0], centre, centre).toarray(),
        'b': d.b, 'f2': d.local_eta_rf_squared, 'ceps': d.min_diffusion_evs,
        'rhs_red': np.stack(rd.rhs), 'E_red': np.stack(rd.energy),
        'nc_centre': rd.nc[centre], 'r_dd_centre': rd.r_dd[centre], 'df_bb_centre': rd.df_bb[centre],
        'u': u, 'eta': np.array([eta]), 'eta_nc': nc, 'eta_r': r, 'eta_df': df, 'indicators': ind,
        'fom_u': U, 'fom_eta': np.array([eta_f]), 'fom_eta_nc': ncf, 'fom_eta_r': rf, 'fom_eta_df': dff,
        # online enrichment: neighbourhood corrector of every subdomain at mu (block_swipdg.py:227-316)
        'local_correction': np.stack([d.solve_for_local_correction(ii, mu) for ii in range(d.S)]),
    }
    return p, out


def build3d(name):
    """3D / P2 (config 5) fixture: inputs and the oracle's expected outputs for one of tests/common3d.py's problems."""
    import common3d as c3
    p = c3.make_problem(name)
    d = c3.oracle_of(p)
    V = c3.make_bases3d(d.S, d.n, p['N'], seed=3)
    rd = c3.reduce_with_oracle(p, d, V)
    u = np.stack(rd.solve(p['mu']))
    rng = np.random.default_rng(5)
    ur = rng.standard_normal((d.S, p['N']))
    nc, r, df = rd.local_terms([ur[ii] for ii in range(d.S)], p['mu'])
    blk = c3.oracle_dense_blocks(p, d, rd, 0)
    return {'N': p['N'], 'mu': np.array([p['mu']]), 'V': V, 'u_random': ur, 'b': d.b, 'f2': d.f2, 'ceps': d.ceps,
            'rhs_red': np.stack(rd.rhs), 'u': u, 'eta_nc': nc, 'eta_r': r, 'eta_df': df,
            'G_nc_0': blk['G_nc'], 'G_bb_0': blk['G_bb'], 'G_rdd_0': blk['G_rdd'], 'r_fd_0': blk['r_fd'], 'G_ab_0': blk['G_ab'],
            'G_aa_0': blk['G_aa'], 'B_sys_0': blk['B_sys'], 'eta': np.array([rd.estimate([u[ii] for ii in range(d.S)], p['mu'])])}


CASES3D = ('aniso_2x2x1', 'q3_2x1x2')


if __name__ == '__main__':
    for name in CASES3D:
        path = os.path.join(HERE, 'cfg5_' + name + '.npz')
        np.savez_compressed(path, **build3d(name))
        print('wrote', path)
    for name in CASES:
        _, out = build(name)
        path = os.path.join(HERE, name + '.npz')
        np.savez_compressed(path, **out)
        print(name, os.path.getsize(path) // 1024, 'KiB')
