"""Parity of the HIP hot path (through the C ABI) with the CPU oracle -- runs on the MI355X box (`-m gpu`).

Tolerance: fp64, relative to the largest entry of each reference array: 1e-11 for assembled / projected arrays
(observed ~1e-15), 1e-10 for the reduced solve (north_star: online reduced solve within 1e-10)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))

from common import (compare_all, energy_orthonormalize, make_bases, oracle_from_problem, theta_bar_of, theta_of)

pytestmark = pytest.mark.gpu

TOL = 1e-11


def _engine(p):
    from pylrbms_amd.engine import Engine
    lam = p['lambda']
    return Engine(p['grid'], lam['functions'], p['kappa'], p['f'], p['lambda_bar'], p['lambda_hat'],
                  theta_bar_of(p)).assemble()


def _problems():
    from pylrbms_amd import OS2015_academic_problem, multiscale_problem, thermalblock_problem
    return {
        # BASELINE.json config 1
        'thermalblock_2x2': (lambda: thermalblock_problem.init_grid_and_problem(
            {'num_subdomains': [2, 2], 'half_num_fine_elements_per_subdomain_and_dim': 4}), 2, (0.5, 1.0, 0.2, 0.8)),
        'os2015_2x2': (lambda: OS2015_academic_problem.init_grid_and_problem(
            {'num_subdomains': [2, 2], 'half_num_fine_elements_per_subdomain_and_dim': 4}), 3, 0.3),
        'os2015_4x4_h8': (lambda: OS2015_academic_problem.init_grid_and_problem(
            {'num_subdomains': [4, 4], 'half_num_fine_elements_per_subdomain_and_dim': 8}), 5, 1.0),
        # edge cases: a single subdomain (no neighbours), strips, ragged template (kx != ky), N = 1, odd N
        'single_subdomain': (lambda: multiscale_problem.init_grid_and_problem(
            {'num_subdomains': [1, 1], 'coarse_per_subdomain': 3}), 4, 0.5),
        'strip_1x4': (lambda: multiscale_problem.init_grid_and_problem(
            {'num_subdomains': [1, 4], 'coarse_per_subdomain': 2}), 1, 0.9),
        'multiscale_5x3_N7': (lambda: multiscale_problem.init_grid_and_problem(
            {'num_subdomains': [5, 3], 'coarse_per_subdomain': 2}), 7, 0.15),
        # BASELINE.json config 2 geometry (k_c = 4, N = 20) on a smaller subdomain grid
        'multiscale_4x3_kc4_N20': (lambda: multiscale_problem.init_grid_and_problem(
            {'num_subdomains': [4, 3], 'coarse_per_subdomain': 4}), 20, 0.7),
        # BASELINE.json config 3's exact kernel shapes (k_c = 4, N = 40, Q = 2: k_f1v<3,2,1,2,4>, k_prep_lds<3> with the G_nc fold,
        # k_f2<5>, k_thin3<3>) on a 3 x 3 grid (one interior subdomain, every boundary kind) directly against the oracle
        'multiscale_3x3_kc4_N40': (lambda: multiscale_problem.init_grid_and_problem(
            {'num_subdomains': [3, 3], 'coarse_per_subdomain': 4}), 40, 0.3),
        # N = 34: the second instantiation of the lean projection kernel for three row tiles (k_f1v<3,2,1,2,3>: six levels of
        # column tiles), with short packed tails of the symmetric groups (2 columns each)
        'multiscale_3x2_kc4_N34': (lambda: multiscale_problem.init_grid_and_problem(
            {'num_subdomains': [3, 2], 'coarse_per_subdomain': 4}), 34, 0.55),
        # the widest supported basis (N = 64, QN = 128): k_f1 needs two column slices -> its generic (runtime-Q) producer
        'multiscale_2x2_kc4_N64': (lambda: multiscale_problem.init_grid_and_problem(
            {'num_subdomains': [2, 2], 'coarse_per_subdomain': 4}), 64, 0.8),
        # a larger template (n = 864, 24 touching elements per side) through the same fused kernels
        'multiscale_2x2_kc6_N24': (lambda: multiscale_problem.init_grid_and_problem(
            {'num_subdomains': [2, 2], 'coarse_per_subdomain': 6}), 24, 0.45),
        # SURVEY.md 8d sweep point k_c = 8 (n = 1536, 32 touching elements per side: k_thin_nc with 78 KB of LDS)
        'multiscale_2x2_kc8_N40': (lambda: multiscale_problem.init_grid_and_problem(
            {'num_subdomains': [2, 2], 'coarse_per_subdomain': 8}), 40, 0.6),
        # SURVEY.md 8d sweep point k_c = 16 (n = 6144, 2 048 elements per subdomain): fused in the factored layout only --
        # k_f1u splits the element range four ways so that its stiffness table fits the LDS
        'multiscale_2x2_kc16_N40': (lambda: multiscale_problem.init_grid_and_problem(
            {'num_subdomains': [2, 2], 'coarse_per_subdomain': 16}), 40, 0.35),
    }


@pytest.mark.parametrize('name', list(_problems()))
def test_every_array_matches_the_oracle(name):
    mk, N, mu = _problems()[name]
    p = mk()
    eng = _engine(p)
    d = oracle_from_problem(p)
    if 'kc16' in name:
        assert eng.ctx.fused_supported(eng.Q, N, factored=True) and not eng.ctx.fused_supported(eng.Q, N)
    elif 'kc8' in name or 'kc6' in name or 'kc4' in name:
        assert eng.ctx.fused_supported(eng.Q, N), 'these templates must run through the fused pass'
    V = energy_orthonormalize(make_bases(d.S, d.n, N, seed=3), d)
    res = compare_all(p, eng, V, mu)
    its = res.pop('cg_iterations')
    assert its > 0
    bad = {k: v for k, v in res.items() if not (v < (1e-10 if k == 'u_solve' else TOL))}
    assert not bad, bad


def test_rectangular_template():
    """kx != ky: 6x2 coarse squares in 3x1 subdomains."""
    from pylrbms_amd.functions import make_constant_function_2x2, make_expression_function_1x1
    from pylrbms_amd.grid import DDSubdomainsGrid, make_boundary_info
    from pylrbms_amd.parameters import ExpressionParameterFunctional
    grid = DDSubdomainsGrid([0, 0], [3, 1], (6, 4), (3, 1))
    pt = {'diffusion': (1,)}
    p = {'grid': grid, 'boundary_info': make_boundary_info(grid, {'type': 'xt.grid.boundaryinfo.alldirichlet'}),
         'lambda': {'functions': [make_expression_function_1x1(grid, 'x', '1+x[0]*x[1]'),
                                  make_expression_function_1x1(grid, 'x', '0.5+sin(x[0])*sin(x[0])')],
                    'coefficients': [ExpressionParameterFunctional('1.', pt), ExpressionParameterFunctional('diffusion', pt)]},
         'lambda_bar': make_expression_function_1x1(grid, 'x', '1.5+x[0]*x[1]+sin(x[0])*sin(x[0])'),
         'lambda_hat': make_expression_function_1x1(grid, 'x', '1.5+x[0]*x[1]+sin(x[0])*sin(x[0])'),
         'kappa': make_constant_function_2x2(grid, [[2., 0.5], [0.5, 1.]]),      # anisotropic constant tensor
         'f': make_expression_function_1x1(grid, 'x', 'exp(x[0])*cos(3*x[1])'),
         'mu_bar': (1.,), 'mu_hat': (1.,)}
    eng = _engine(p)
    d = oracle_from_problem(p)
    V = energy_orthonormalize(make_bases(d.S, d.n, 6, seed=5), d)
    res = compare_all(p, eng, V, 0.45)
    res.pop('cg_iterations')
    bad = {k: v for k, v in res.items() if not (v < (1e-10 if k == 'u_solve' else TOL))}
    assert not bad, bad


@pytest.mark.parametrize('name', ['os2015_2x2', 'thermalblock_2x2', 'multiscale_3x3'])
def test_golden_fixtures(name):
    """The committed (oracle-generated) vectors: stored inputs V, mu -> stored u, eta triple, Gram blocks."""
    from make_golden import CASES
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', name + '.npz'), allow_pickle=False)
    mk, N, mu = CASES[name]
    p = mk()
    eng = _engine(p)
    grid = p['grid']
    S = grid.num_subdomains
    V = eng.ctx.from_numpy(ref['V'])
    buf = eng.project_and_estimate(V)
    host = lambda x: x.detach().cpu().numpy()  # noqa: E731
    assert np.abs(host(eng.b).reshape(-1) - ref['b']).max() < TOL * np.abs(ref['b']).max()
    assert np.allclose(host(eng.f2), ref['f2'], rtol=TOL) and np.allclose(host(eng.ceps), ref['ceps'], rtol=TOL)
    assert np.abs(host(buf['sys'][1]) - ref['rhs_red']).max() < TOL * np.abs(ref['rhs_red']).max()
    assert np.abs(host(buf['sys'][2]) - ref['E_red']).max() < TOL * np.abs(ref['E_red']).max()
    theta = theta_of(p, mu)
    u, info = eng.reduced_solve(theta, buf['sys'][0], buf['sys'][1])
    assert np.linalg.norm(host(u) - ref['u']) < 1e-10 * np.linalg.norm(ref['u'])
    eta = host(eng.reduced_estimate(theta, eng.ctx.from_numpy(ref['u']), buf['grams']))
    for row, key in enumerate(('eta_nc', 'eta_r', 'eta_df')):
        assert np.abs(eta[row] - ref[key]).max() < 1e-10 * max(np.abs(ref[key]).max(), 1e-300), key
    # FOM estimate = the same pipeline with the full-order vector as a one-column basis
    U = eng.ctx.from_numpy(ref['fom_u'][:, :, None])
    bufU = eng.project_and_estimate(U, project_system=False)
    etaU = host(eng.reduced_estimate(theta, eng.ctx.from_numpy(np.ones((S, 1))), bufU['grams']))
    for row, key in enumerate(('fom_eta_nc', 'fom_eta_r', 'fom_eta_df')):
        assert np.abs(etaU[row] - ref[key]).max() < 1e-9 * max(np.abs(ref[key]).max(), 1e-300), key
    # online enrichment correctors (iterative on the GPU, direct in the oracle: 1e-8)
    corr, _ = eng.local_corrections(theta, list(range(S)))
    assert np.abs(host(corr) - ref['local_correction']).max() < 1e-8 * np.abs(ref['local_correction']).max()


def test_full_size_properties_config2():
    """BASELINE.json config 2 at full size (8x8 subdomains, N = 20) through size-independent properties:
    symmetry of every Gram, linearity (reduced estimate of u == estimate of the reconstruction V u as a one-column
    basis), and agreement of the fp64-MFMA GEMM with torch.matmul on the same device operands."""
    import torch
    from pylrbms_amd import multiscale_problem
    p = multiscale_problem.init_grid_and_problem({'num_subdomains': [8, 8], 'coarse_per_subdomain': 4})
    eng = _engine(p)
    S, n, N = eng.S, eng.t.n, 20
    V = eng.ctx.from_numpy(make_bases(S, n, N, seed=2))
    buf = eng.project_and_estimate(V)
    from pylrbms_amd.engine import expand_factored_grams
    G_nc, r_fd, G_rdd, G_bb, G_ab, G_aa = expand_factored_grams(buf['grams'])
    assert float((G_nc - G_nc.transpose(1, 2)).abs().max()) <= 1e-12 * float(G_nc.abs().max())
    for G in (G_rdd, G_bb):                                   # block-compact: blocks 0 and 5..8 are symmetric
        for b in (0, 5, 6, 7, 8):
            assert float((G[:, b] - G[:, b].transpose(1, 2)).abs().max()) <= 1e-12 * float(G.abs().max())
    B_sys = buf['sys'][0]
    assert float((B_sys[:, :, 2] - B_sys[:, :, 2].transpose(2, 3)).abs().max()) <= 1e-12 * float(B_sys.abs().max())
    X = eng.ctx.from_numpy(np.random.default_rng(3).standard_normal((S, n, 5 * N)))
    ref = torch.matmul(X.transpose(1, 2), X)
    got = eng.ctx.gemm_tn(X, X)
    assert float((ref - got).abs().max()) <= 1e-12 * float(ref.abs().max())
    theta = theta_of(p, 0.35)
    rng = np.random.default_rng(5)
    u = eng.ctx.from_numpy(rng.standard_normal((S, N)))
    eta = eng.reduced_estimate(theta, u, buf['grams'])
    U = torch.einsum('snk,sk->sn', V, u)[:, :, None].contiguous()
    bufU = eng.project_and_estimate(U, project_system=False)
    etaU = eng.reduced_estimate(theta, eng.ctx.from_numpy(np.ones((S, 1))), bufU['grams'])
    scale = etaU.abs().max(dim=1, keepdim=True).values
    assert float(((eta - etaU).abs() / scale).max()) < 1e-9


@pytest.mark.parametrize('shape, N', [((8, 8), 20), ((5, 3), 33)])
def test_k_split_of_the_projection_kernel_is_order_independent(monkeypatch, shape, N):
    """k_f1u spreads the elements of a subdomain over 2 / 4 workgroups when a rank has few subdomains (partial tiles handed
    over write-through, the workgroup that arrives last sums them in a fixed order).  The split results agree with the
    unsplit ones to rounding, repeated split passes are bit-identical although the arrival order varies, and a pass with
    different data in between leaves nothing behind in the hand-over buffers.  (5 x 3 subdomains: the parts of one
    subdomain land on different XCDs under round-robin placement.)"""
    import torch
    from pylrbms_amd import multiscale_problem
    p = multiscale_problem.init_grid_and_problem({'num_subdomains': list(shape), 'coarse_per_subdomain': 4})
    eng = _engine(p)
    S, n = eng.S, eng.t.n
    V = eng.ctx.from_numpy(make_bases(S, n, N, seed=2))
    V2 = eng.ctx.from_numpy(make_bases(S, n, N, seed=9))
    outs = {}
    for ks in ('1', '2', '4'):
        eng.ctx.set_option('f1_ksplit', int(ks))
        buf = eng.project_and_estimate(V)
        first = [x.clone() for x in buf['sys']] + [x.clone() for x in buf['grams']]
        for rep in range(4):
            eng.project_and_estimate(V2, buf)                 # other data through the same hand-over buffers
            buf = eng.project_and_estimate(V, buf)
            for a, b in zip(first, list(buf['sys']) + list(buf['grams'])):
                assert torch.equal(a, b), 'K-split {}: repeat {} differs'.format(ks, rep)
        outs[ks] = first
    for ks in ('2', '4'):
        for a, b in zip(outs['1'], outs[ks]):
            assert float((a - b).abs().max()) <= 1e-13 * float(a.abs().max())


@pytest.mark.parametrize('shape, kc, N', [((6, 5), 4, 40), ((3, 3), 4, 36), ((4, 4), 2, 20), ((3, 2), 4, 38)])
def test_forms_of_the_projection_kernel_agree(shape, kc, N):
    """LRBMS_OPT_F1_FORM: the lean kernel k_f1v (symmetric groups cut into column blocks, mirror entries stored from the upper
    triangle, mass / stiffness / rhs rows on the apply MFMA), the unified kernel k_f1u and the producer / consumer kernel k_f1
    compute the same projected operators to summation-order rounding; the symmetric outputs of k_f1v are EXACTLY symmetric (both
    halves come from one accumulator entry)."""
    import torch
    from pylrbms_amd import multiscale_problem
    p = multiscale_problem.init_grid_and_problem({'num_subdomains': list(shape), 'coarse_per_subdomain': kc})
    eng = _engine(p)
    V = eng.ctx.from_numpy(make_bases(eng.S, eng.t.n, N, seed=6))
    outs = {}
    try:
        for form in (0, 2, 1, 3):      # 0: the leanest form (k_f1w at N = 34 .. 40, else k_f1v), 3: k_f1v where 0 took k_f1w
            eng.ctx.set_option('f1_form', form)
            eng.ctx.kernel_timing(True)
            buf = eng.project_and_estimate(V)
            ran = {k for k, _ in eng.ctx.kernel_timing_read()}
            eng.ctx.kernel_timing(False)
            lean3 = 34 <= N <= 40 and eng.Q == 2      # k_f1w's shape (the k_f1v instantiations for it are retired: form 3 runs k_f1u there)
            want = {0: 'k_f1w' if lean3 else 'k_f1v', 2: 'k_f1u', 1: 'k_f1', 3: 'k_f1u' if lean3 else 'k_f1v'}[form]
            assert want in ran, (form, sorted(ran))
            outs[form] = [x.clone() for x in buf['sys']] + [x.clone() for x in buf['grams']]
        if 34 <= N <= 40 and eng.Q == 2:
            # k_f1w multiplies the flux rows with W' = hab A_ab, which lrbms_assemble_products leaves with the context for ITS Aab; a
            # pass that is handed any other array forms the factors itself (k_wab) -- the same bits
            eng.ctx.set_option('f1_form', 0)
            eng.Aab = eng.Aab.clone()
            eng.__dict__.pop('_bound_pass', None)
            eng.ctx.kernel_timing(True)
            buf = eng.project_and_estimate(V)
            ran = {k for k, _ in eng.ctx.kernel_timing_read()}
            eng.ctx.kernel_timing(False)
            assert 'k_wab' in ran and 'k_f1w' in ran, sorted(ran)
            for a, b in zip(outs[0], list(buf['sys']) + list(buf['grams'])):
                assert torch.equal(a, b)
    finally:
        eng.ctx.set_option('f1_form', 0)
    for form in (2, 1, 3):
        for a, b in zip(outs[0], outs[form]):
            assert float((a - b).abs().max()) <= 1e-12 * float(a.abs().max()), form
    B_sys, rhs_red, E_red, M_red = outs[0][:4]
    assert torch.equal(B_sys[:, :, 2], B_sys[:, :, 2].transpose(2, 3))
    assert torch.equal(E_red, E_red.transpose(1, 2)) and torch.equal(M_red, M_red.transpose(1, 2))
    G_aa = outs[0][4 + 5]
    for q in range(eng.Q):
        for q2 in range(eng.Q):
            assert torch.equal(G_aa[q, q2], G_aa[q2, q].transpose(1, 2))


@pytest.mark.parametrize('shape, kc, N', [((6, 5), 4, 40), ((3, 3), 4, 36), ((4, 4), 2, 20), ((3, 2), 1, 6), ((2, 3), 4, 48)])
def test_forms_of_the_preparation_agree(shape, kc, N):
    """LRBMS_OPT_PREP_LDS: the preparation from one copy of the basis slab in LDS (k_prep_lds: flux image, vertex averages and --
    folded in -- G_nc[self, self] in its rank-2 form), the same without the fold (k_f3 runs) and the two streaming sweeps
    (k_flux_compact, k_vertex_avg) give the same outputs: flux image and vertex averages feed every estimator operator, so all of
    them are compared -- bit for bit where the arithmetic is the same expression (everything but G_nc), to rounding for G_nc (the
    fold sums 2 rows per element, Z = L^T G W, where k_f3 sums 3); G_nc is symmetric (off-diagonal tiles are mirrored: exactly; inside
    a diagonal tile the two halves are separate sums: to rounding)."""
    import torch
    from pylrbms_amd import multiscale_problem
    p = multiscale_problem.init_grid_and_problem({'num_subdomains': list(shape), 'coarse_per_subdomain': kc})
    eng = _engine(p)
    V = eng.ctx.from_numpy(make_bases(eng.S, eng.t.n, N, seed=8))
    outs = {}
    try:
        for form in (1, 2, 0):
            eng.ctx.set_option('prep_lds', form)
            buf = eng.project_and_estimate(V)
            outs[form] = [x.clone() for x in buf['sys']] + [x.clone() for x in buf['grams']]
    finally:
        eng.ctx.set_option('prep_lds', 1)
    G_nc = outs[1][4]
    assert float((G_nc - G_nc.transpose(1, 2)).abs().max()) <= 1e-14 * float(G_nc.abs().max())
    for form in (2, 0):
        for i, (a, b) in enumerate(zip(outs[1], outs[form])):
            if i == 4:          # G_nc [S][N][N]: another summation
                assert float((a - b).abs().max()) <= 1e-12 * float(a.abs().max()), form
            else:
                assert torch.equal(a, b), (form, i)


@pytest.mark.parametrize('shape, kc, N, vertex_patch', [((20, 16), 2, 10, False), ((20, 15), 4, 40, False), ((18, 16), 2, 6, True),
                                                        ((17, 17), -34, 8, False)])
def test_persistent_preparation_equals_one_workgroup_per_subdomain(shape, kc, N, vertex_patch):
    """More subdomains than CUs: k_prep_lds runs one workgroup per CU that takes its subdomains one after the other and holds the next
    one's slab in registers while it works on the current one (asm prefetch, template tables kept in LDS).  Bit for bit the result of
    one workgroup per subdomain (LRBMS_OPT_PREP_LDS 3) -- whole pass, the two phases, an incremental subset -- with outputs and work
    buffer poisoned in front of every run; on the small template (k_c = 2: fewer flux items than threads, a single tile) also against
    the streaming sweeps and, through compare_all, against the oracle."""
    import torch
    from pylrbms_amd import multiscale_problem, thermalblock_problem
    from pylrbms_amd.engine import Engine
    if kc < 0:      # the thermal block problem: Q = 4 affine components (the coefficient prefetch over four component blocks); -kc: the GLOBAL
                    # number of coarse squares per direction (grid.py:24-25), two per subdomain here
        p = thermalblock_problem.init_grid_and_problem({'num_subdomains': list(shape), 'half_num_fine_elements_per_subdomain_and_dim': -kc})
    else:
        p = multiscale_problem.init_grid_and_problem({'num_subdomains': list(shape), 'coarse_per_subdomain': kc})
    lam = p['lambda']
    eng = Engine(p['grid'], lam['functions'], p['kappa'], p['f'], p['lambda_bar'], p['lambda_hat'], theta_bar_of(p),
                 conventions={'oswald_vertex_patch': True} if vertex_patch else None).assemble()
    assert eng.S > 256
    V = eng.ctx.from_numpy(make_bases(eng.S, eng.t.n, N, seed=31))
    buf = eng.alloc_reduce_buffers(N)
    args = (V, eng.F, eng.A_diag, eng.A_cpl, eng.P_diag, eng.b, eng.ebar, eng.caa, eng.Aab, eng.Bbb, buf['work'], buf['sys'], buf['grams'])
    everything = lambda: list(buf['sys']) + list(buf['grams'])

    def run(prep, how):
        eng.ctx.set_option('prep_lds', prep)
        for x in everything() + [buf['work']]:
            x.fill_(float('nan'))
        if how == 'whole':
            eng.ctx.project_estimate_fused(*args, phase=0)
        elif how == 'phases':
            eng.ctx.project_estimate_fused(*args, phase=1)
            eng.ctx.project_estimate_fused(*args, phase=2)
        else:
            eng.ctx.project_estimate_fused(*args, phase=5)
        return [x.clone() for x in everything()]

    try:
        ref = run(3, 'whole')
        assert all(bool(torch.isfinite(x).all()) for x in ref)
        for how in ('whole', 'phases', 'one call'):
            got = run(1, how)
            for i, (a, b) in enumerate(zip(ref, got)):
                assert torch.equal(a, b), (how, i)
        # an incremental subset of more subdomains than CUs: persistent as well, the other rows untouched
        subset = [i for i in range(eng.S) if i % 16 != 3]
        assert len(subset) > 256
        for x in everything():
            x.fill_(7.0)
        buf['work'].fill_(float('nan'))
        eng.project_and_estimate(V, buf, subset=subset)
        rest = torch.tensor([i for i in range(eng.S) if i % 16 == 3], device=V.device)
        sub = torch.tensor(subset, device=V.device)
        for i, (a, b) in enumerate(zip(ref, everything())):
            sdim = 1 if (a.dim() >= 2 and a.shape[0] == eng.Q and a.shape[1] == eng.S) else 0
            if a.dim() >= 3 and a.shape[0] == eng.Q and a.shape[1] == eng.Q and a.shape[2] == eng.S:
                sdim = 2
            assert torch.equal(a.index_select(sdim, sub), b.index_select(sdim, sub)), i
            assert bool((b.index_select(sdim, rest) == 7.0).all()), i
        if kc == 2 and not vertex_patch:
            stream = run(0, 'whole')
            for i, (a, b) in enumerate(zip(ref, stream)):
                if i == 4:          # G_nc: another summation (test_forms_of_the_preparation_agree)
                    assert float((a - b).abs().max()) <= 1e-12 * float(a.abs().max())
                else:
                    assert torch.equal(a, b), i
    finally:
        eng.ctx.set_option('prep_lds', 1)
    if kc == 2 and not vertex_patch:
        d = oracle_from_problem(p)
        Vh = energy_orthonormalize(make_bases(d.S, d.n, N, seed=32), d)
        res = compare_all(p, eng, Vh, 0.6, oracle=d, do_solve=False)
        bad = {k: v for k, v in res.items() if not v < TOL}
        assert not bad, bad


@pytest.mark.parametrize('shape, kc, N', [((3, 3), 2, 5), ((4, 3), 1, 2), ((2, 2), 2, 40)])
def test_vertex_patch_of_the_oswald_interpolation(shape, kc, N):
    """LRBMS_OPT_OSWALD_VERTEX_PATCH (conventions={'oswald_vertex_patch': True}): the Oswald average at a cross point runs over
    the elements of ALL four subdomains that meet there -- the reading that reproduces the reference's printed nonconformity value
    (tests/test_reference_pin.py) -- instead of HEAD's subdomain + face neighbours.  The diagonal subdomains enter through the
    factored layout (one more column block A_diag in F_nc, added to z at the corner vertices by the estimate kernels): local
    nonconformity terms of a reduced model against the oracle's reductor with ``oswald_patch='vertex'`` (1e-10), single and
    batched; residual and diffusive-flux terms are untouched; the dense five-slot layout refuses the option."""
    from pylrbms_amd import multiscale_problem
    from pylrbms_amd._native import NativeError
    from pylrbms_amd.engine import Engine
    from oracle.lrbms import OracleReductor
    p = multiscale_problem.init_grid_and_problem({'num_subdomains': list(shape), 'coarse_per_subdomain': kc})
    lam = p['lambda']
    eng = Engine(p['grid'], lam['functions'], p['kappa'], p['f'], p['lambda_bar'], p['lambda_hat'], theta_bar_of(p),
                 conventions={'oswald_vertex_patch': True}).assemble()
    head = _engine(p)
    d = oracle_from_problem(p, oswald_patch='vertex')
    V = energy_orthonormalize(make_bases(d.S, d.n, N, seed=12), d)
    Vd = eng.ctx.from_numpy(V)
    buf = eng.project_and_estimate(Vd)
    assert buf['grams'][7].shape[3] == 3 * N + 4 * eng.ctx.nvs          # A_a | C_a | M | A_diag
    rd = OracleReductor(d, [V[ii] for ii in range(d.S)]).reduce()
    rng = np.random.default_rng(3)
    mus = [0.15, 0.6, 1.0]
    U = rng.standard_normal((d.S, N, len(mus)))
    thetas = np.stack([theta_of(p, mu) for mu in mus])
    eta_b = eng.ctx.reduced_estimate_batch(thetas, eng.ctx.from_numpy(U), buf['grams'], eng.f2, eng.ceps, eng.hdiam).cpu().numpy()
    buf_h = head.project_and_estimate(head.ctx.from_numpy(V))
    differs = 0.0
    for m, mu in enumerate(mus):
        um = np.ascontiguousarray(U[:, :, m])
        eta_s = eng.reduced_estimate(thetas[m], eng.ctx.from_numpy(um), buf['grams']).cpu().numpy()
        _, (nc, r, df), _ = rd.estimate([um[ii] for ii in range(d.S)], mu, decompose=True)
        for row, ref_row in enumerate((nc, r, df)):
            scale = max(np.abs(ref_row).max(), 1e-300)
            assert np.abs(eta_s[row] - ref_row).max() < 1e-10 * scale, (m, row)
            assert np.abs(eta_b[row, :, m] - ref_row).max() < 1e-10 * scale, (m, row)
        eta_h = head.reduced_estimate(thetas[m], head.ctx.from_numpy(um), buf_h['grams']).cpu().numpy()
        assert np.abs(eta_h[1] - eta_s[1]).max() < 1e-12 * np.abs(eta_s[1]).max()          # residual, diffusive flux: the same
        assert np.abs(eta_h[2] - eta_s[2]).max() < 1e-12 * np.abs(eta_s[2]).max()
        differs = max(differs, float(np.abs(eta_h[0] - eta_s[0]).max() / np.abs(eta_s[0]).max()))
    assert differs > 1e-4                                                # the two readings are different operators (cross points exist)
    with pytest.raises(NativeError, match='VERTEX_PATCH'):
        eng.project_and_estimate(Vd, eng.alloc_reduce_buffers(N, factored=False))            # dense five-slot layout
    with pytest.raises(NativeError, match='VERTEX_PATCH'):
        eng.ctx.oswald_apply(Vd)                                                              # five-slot image basis
    from pylrbms_amd.engine import expand_factored_grams
    with pytest.raises(NotImplementedError):
        expand_factored_grams(buf['grams'])


def test_launch_matrix_against_one_oracle_checked_result():
    """Every launch combination of the fused pass against ONE result that is itself compared with the oracle: layout {factored,
    dense} x launch policy {one stream, forked over the library streams} x preparation {streaming sweeps + k_f3, LDS slab with the
    G_nc fold, LDS slab without it} x projection kernel {k_f1w, k_f1 (producer / consumer), k_f1u, k_f1u again (form 3: k_f1v where it is instantiated)} x {whole pass, phase 1 then
    phase 2, phase 5 = both halves in one call with the halo-dependent one on library stream 0}: 144 cells on a 5 x 4 grid of the config-3 template (k_c = 4, N = 40).  Outputs AND the work buffer are poisoned with NaN
    in front of every cell, so a cell that forgets a launch (round 3: k_vertex_side in factored x unforked x phase 2) cannot pass on
    what the cell before left behind.  Reference: the unfused kernels' result, compared with the oracle's reductor at 1e-11; every
    cell against it at 1e-12 (dense-layout arrays; factored cells expanded)."""
    import itertools
    import torch
    from pylrbms_amd import multiscale_problem
    from pylrbms_amd.engine import expand_factored_grams
    p = multiscale_problem.init_grid_and_problem({'num_subdomains': [5, 4], 'coarse_per_subdomain': 4})
    eng = _engine(p)
    d = oracle_from_problem(p)
    N = 40
    V = energy_orthonormalize(make_bases(d.S, d.n, N, seed=21), d)
    res = compare_all(p, eng, V, 0.4, oracle=d, do_solve=False)
    bad = {k: v for k, v in res.items() if not v < TOL}
    assert not bad, bad
    Vd = eng.ctx.from_numpy(V)
    ref = eng.project_and_estimate(Vd, fused=False)
    names = ('B_sys', 'rhs_red', 'E_red', 'M_red', 'G_nc', 'r_fd', 'G_rdd', 'G_bb', 'G_ab', 'G_aa')
    ref = dict(zip(names, [x.clone() for x in list(ref['sys']) + list(ref['grams'])]))
    scale = {k: float(v.abs().max()) for k, v in ref.items()}
    bufs = {True: eng.alloc_reduce_buffers(N, factored=True), False: eng.alloc_reduce_buffers(N, factored=False)}
    failures, seen = [], set()
    try:
        for factored, streams, prep, form, phased in itertools.product((True, False), (0, 1), (0, 1, 2), (0, 1, 2, 3), (0, 1, 5)):
            eng.ctx.set_option('streams', streams)
            eng.ctx.set_option('prep_lds', prep)
            eng.ctx.set_option('f1_form', form)
            buf = bufs[factored]
            for x in list(buf['sys']) + list(buf['grams']) + [buf['work']]:
                x.fill_(float('nan'))
            args = (Vd, eng.F, eng.A_diag, eng.A_cpl, eng.P_diag, eng.b, eng.ebar, eng.caa, eng.Aab, eng.Bbb, buf['work'], buf['sys'],
                    buf['grams'])
            eng.ctx.kernel_timing(True)
            if phased == 1:
                eng.ctx.project_estimate_fused(*args, phase=1)
                eng.ctx.project_estimate_fused(*args, phase=2)
            else:
                eng.ctx.project_estimate_fused(*args, phase=phased)
            seen.update(k for k, _ in eng.ctx.kernel_timing_read())
            eng.ctx.kernel_timing(False)
            for name, got in zip(names, list(buf['sys']) + list(expand_factored_grams(buf['grams']))):
                err = float((got - ref[name]).abs().max()) / scale[name]
                if not err <= 1e-12:          # (NaN fails too)
                    failures.append(((factored, streams, prep, form, phased), name, err))
    finally:
        for k, v in (('streams', -1), ('prep_lds', 1), ('f1_form', 0)):
            eng.ctx.set_option(k, v)
    assert not failures, failures[:12]
    # the matrix did reach every kernel variant it is meant to enumerate
    for k in ('k_f1w', 'k_f1u', 'k_f1', 'k_prep_lds', 'k_prep_lds<side>', 'k_flux_compact', 'k_vertex_avg',
              'k_flux_side', 'k_vertex_side', 'k_f2', 'k_f3', 'k_thin3', 'k_thin', 'k_thin_nc', 'k_thin_rt', 'k_coupling', 'k_thin_expand'):
        assert k in seen, (k, sorted(seen))


def test_full_size_properties_config3(monkeypatch):
    """BASELINE.json config 3 at full size (32x32 subdomains, N = 40: the benchmark workload).  Size-independent
    properties: the fused pass gives bit-identical results with its kernels serial or forked over the library's
    streams (every reduction has a fixed order), every symmetric operator is symmetric, the energy Gram of an
    energy-orthonormal basis is the identity, and the reduced estimate is linear in the sense of the config-2 test."""
    import torch
    from pylrbms_amd import multiscale_problem
    p = multiscale_problem.init_grid_and_problem({'num_subdomains': [32, 32], 'coarse_per_subdomain': 4})
    eng = _engine(p)
    S, n, N = eng.S, eng.t.n, 40
    assert eng.ctx.fused_supported(eng.Q, N)
    V = eng.ctx.from_numpy(make_bases(S, n, N, seed=4))
    eng.ctx.set_option('streams', 0)
    buf = eng.project_and_estimate(V)
    serial = [x.clone() for x in buf['sys']] + [x.clone() for x in buf['grams']]
    eng.ctx.set_option('streams', 1)
    buf = eng.project_and_estimate(V, buf)
    forked = list(buf['sys']) + list(buf['grams'])
    for a, b in zip(serial, forked):
        assert torch.equal(a, b)
    eng.ctx.set_option('streams', -1)
    # the two halves of the pass (halo-independent / halo-dependent) give the same bits as the whole
    # (outputs AND the work buffer poisoned: the halves must produce every intermediate themselves -- flux image and vertex
    # averages of the own basis in phase 1, the neighbours' shares in phase 2; a work buffer left over from the whole pass hid a
    # missing k_vertex_side launch in round 3)
    args = (V, eng.F, eng.A_diag, eng.A_cpl, eng.P_diag, eng.b, eng.ebar, eng.caa, eng.Aab, eng.Bbb, buf['work'], buf['sys'],
            buf['grams'])
    for prep in (0, 1, 3):      # streaming sweeps; LDS copy (1 024 subdomains: one persistent workgroup per CU); one workgroup per subdomain
        eng.ctx.set_option('prep_lds', prep)
        ref = serial
        if prep == 0:          # the streaming sweeps + k_f3: G_nc is another summation (test_forms_of_the_preparation_agree)
            for x in list(buf['sys']) + list(buf['grams']) + [buf['work']]:
                x.fill_(float('nan'))
            eng.project_and_estimate(V, buf)
            ref = [x.clone() for x in buf['sys']] + [x.clone() for x in buf['grams']]
        for x in list(buf['sys']) + list(buf['grams']) + [buf['work']]:
            x.fill_(float('nan'))
        eng.ctx.project_estimate_fused(*args, phase=1)
        eng.ctx.project_estimate_fused(*args, phase=2)
        for a, b in zip(ref, list(buf['sys']) + list(buf['grams'])):
            assert torch.equal(a, b)
        # ... and so do both halves in ONE call (phase 5: the halo-dependent one on library stream 0)
        for x in list(buf['sys']) + list(buf['grams']) + [buf['work']]:
            x.fill_(float('nan'))
        eng.ctx.project_estimate_fused(*args, phase=5)
        if prep == 0:
            for a, b in zip(ref, list(buf['sys']) + list(buf['grams'])):
                assert torch.equal(a, b)
    for a, b in zip(serial, list(buf['sys']) + list(buf['grams'])):
        assert torch.equal(a, b)
    del serial
    from pylrbms_amd.engine import expand_factored_grams
    G_nc, r_fd, G_rdd, G_bb, G_ab, G_aa = expand_factored_grams(buf['grams'])
    assert float((G_nc - G_nc.transpose(1, 2)).abs().max()) <= 1e-12 * float(G_nc.abs().max())
    for G in (G_rdd, G_bb):
        for b in (0, 5, 6, 7, 8):
            assert float((G[:, b] - G[:, b].transpose(1, 2)).abs().max()) <= 1e-12 * float(G.abs().max())
    B_sys, rhs_red, E_red, M_red = buf['sys']
    assert float((B_sys[:, :, 2] - B_sys[:, :, 2].transpose(2, 3)).abs().max()) <= 1e-12 * float(B_sys.abs().max())
    # coupling blocks: block [s, slot] is the transpose of block [nbr(s, slot), 4 - slot]
    nbr = torch.as_tensor(eng.nbr.astype(np.int64), device=B_sys.device)
    for slot in (0, 1):
        s_idx = torch.nonzero(nbr[:, slot] >= 0)[:, 0]
        other = nbr[s_idx, slot]
        for q in range(eng.Q):
            a = B_sys[q, s_idx, slot]
            b = B_sys[q, other, 4 - slot].transpose(1, 2)
            assert float((a - b).abs().max()) <= 1e-12 * float(B_sys.abs().max())
    # energy-orthonormalise with the projected energy product, project again: E_red == I
    Lh = np.linalg.cholesky(E_red.cpu().numpy())                  # [S, N, N]: small, on the host
    Vo = torch.bmm(V, eng.ctx.from_numpy(np.linalg.inv(Lh).transpose(0, 2, 1))).contiguous()
    bufo = eng.project_and_estimate(Vo, buf)
    eye = torch.eye(N, dtype=torch.float64, device=Vo.device)[None]
    assert float((bufo['sys'][2] - eye).abs().max()) < 1e-10
    theta = theta_of(p, 0.35)
    u = eng.ctx.from_numpy(np.random.default_rng(5).standard_normal((S, N)))
    eta = eng.reduced_estimate(theta, u, bufo['grams'])
    U = torch.einsum('snk,sk->sn', Vo, u)[:, :, None].contiguous()
    bufU = eng.project_and_estimate(U, project_system=False)
    etaU = eng.reduced_estimate(theta, eng.ctx.from_numpy(np.ones((S, 1))), bufU['grams'])
    scale = etaU.abs().max(dim=1, keepdim=True).values
    assert float(((eta - etaU).abs() / scale).max()) < 1e-9


def test_native_argument_checks_raise():
    from pylrbms_amd._native import NativeError
    from pylrbms_amd import multiscale_problem
    p = multiscale_problem.init_grid_and_problem({'num_subdomains': [2, 2], 'coarse_per_subdomain': 2})
    eng = _engine(p)
    with pytest.raises(NativeError):
        eng.ctx.oswald_apply(eng.ctx.zeros(eng.S, eng.t.n + 1, 3))        # wrong shape
    with pytest.raises(NativeError):
        eng.ctx.oswald_apply(eng.ctx.zeros(eng.S, eng.t.n, 3).float())    # wrong dtype
    buf = eng.project_and_estimate(eng.ctx.from_numpy(make_bases(eng.S, eng.t.n, 2, seed=1)))
    with pytest.raises(NativeError):                                      # LRBMS_E_NOT_CONVERGED surfaces as an error
        eng.ctx.reduced_solve(np.array([1.0, 0.5]), buf['sys'][0], buf['sys'][1], rtol=1e-30, max_iter=3)


def test_batched_reduced_solve_matches_single_and_oracle():
    """O1 throughput form: 6 parameters in one batched PCG == 6 single solves == oracle dense solves."""
    from pylrbms_amd import multiscale_problem
    from oracle.lrbms import OracleReductor
    p = multiscale_problem.init_grid_and_problem({'num_subdomains': [4, 3], 'coarse_per_subdomain': 2})
    eng = _engine(p)
    d = oracle_from_problem(p)
    N = 6
    V = energy_orthonormalize(make_bases(d.S, d.n, N, seed=9), d)
    buf = eng.project_and_estimate(eng.ctx.from_numpy(V))
    mus = [0.1, 0.25, 0.4, 0.55, 0.8, 1.0]
    thetas = np.stack([theta_of(p, mu) for mu in mus])
    ub, info = eng.ctx.reduced_solve_batch(thetas, buf['sys'][0], buf['sys'][1])
    assert info['iterations'] > 0 and info['relative_residual'] <= 1e-13
    ub = ub.cpu().numpy()
    rd = OracleReductor(d, [V[ii] for ii in range(d.S)]).reduce()
    for m, mu in enumerate(mus):
        us, _ = eng.reduced_solve(thetas[m], buf['sys'][0], buf['sys'][1])
        ref = np.stack(rd.solve(mu))
        assert np.linalg.norm(ub[:, :, m] - ref) < 1e-10 * np.linalg.norm(ref)
        assert np.linalg.norm(us.cpu().numpy() - ref) < 1e-10 * np.linalg.norm(ref)
    # E1 throughput form on the batched solutions == single-parameter kernel == oracle
    ub_dev = eng.ctx.from_numpy(ub)
    eta_b = eng.ctx.reduced_estimate_batch(thetas, ub_dev, buf['grams'], eng.f2, eng.ceps, eng.hdiam).cpu().numpy()
    for m, mu in enumerate(mus):
        um = np.ascontiguousarray(ub[:, :, m])
        eta_s = eng.reduced_estimate(thetas[m], eng.ctx.from_numpy(um), buf['grams']).cpu().numpy()
        _, (nc, r, df), _ = rd.estimate([um[ii] for ii in range(d.S)], mu, decompose=True)
        for row, ref_row in enumerate((nc, r, df)):
            scale = max(np.abs(ref_row).max(), 1e-300)
            assert np.abs(eta_b[row, :, m] - ref_row).max() < 1e-9 * scale
            assert np.abs(eta_s[row] - ref_row).max() < 1e-9 * scale


@pytest.mark.parametrize('shape, N', [((4, 3), 6), ((6, 6), 40), ((3, 2), 64)])
def test_batched_solve_of_several_groups_on_the_library_streams(shape, N):
    """lrbms_reduced_solve_batch with nmu > 16: groups of 16 parameters as independent CG runs on the caller's stream and the
    library's side streams, launches interleaved, every group scattering its columns into u [S, N, nmu] -- every column equals
    the oracle's dense solve (1e-10), with the per-call preconditioner, with a prebuilt one and with the VALU panel matvec
    (LRBMS_OPT_SOLVE_VALU); 17 and 64 parameters cover a one-column last group and four full groups; N = 40 is the basis size of
    config 3, N = 64 the limit.  No host threads: one call, one caller."""
    from pylrbms_amd import multiscale_problem
    from oracle.lrbms import OracleReductor
    p = multiscale_problem.init_grid_and_problem({'num_subdomains': list(shape), 'coarse_per_subdomain': 2})
    eng = _engine(p)
    d = oracle_from_problem(p)
    V = energy_orthonormalize(make_bases(d.S, d.n, N, seed=9), d)
    buf = eng.project_and_estimate(eng.ctx.from_numpy(V))
    B, rhs = buf['sys'][0], buf['sys'][1]
    rd = OracleReductor(d, [V[ii] for ii in range(d.S)]).reduce()
    for nmu in (17, 64):
        mus = list(np.linspace(0.1, 1.0, nmu))
        thetas = np.stack([theta_of(p, mu) for mu in mus])
        refs = [np.stack(rd.solve(mu)) for mu in mus]

        def check(tag):
            ub, info = eng.ctx.reduced_solve_batch(thetas, B, rhs)
            assert tuple(ub.shape) == (eng.S, N, nmu) and info['iterations'] > 0 and info['relative_residual'] <= 1e-13, tag
            ub = ub.cpu().numpy()
            for m in range(nmu):
                assert np.linalg.norm(ub[:, :, m] - refs[m]) < 1e-10 * np.linalg.norm(refs[m]), (tag, nmu, m)
            return ub
        u0 = check('per-call preconditioner')
        eng.ctx.reduced_precond_use(eng.ctx.reduced_precond_build(theta_of(p, 0.55), B))
        try:
            check('prebuilt preconditioner')
            if nmu == 17:
                eng.ctx.set_option('solve_valu', 1)
                check('VALU panel matvec')
        finally:
            eng.ctx.set_option('solve_valu', 0)
            eng.ctx.reduced_precond_use(None)
        u1 = check('per-call preconditioner again')
        assert np.array_equal(u0, u1)                  # nothing left behind in the work buffers or on the side streams
    # the sweep wrapper: any number of parameters, 64 per call
    mus = list(np.linspace(0.1, 1.0, 70))
    thetas = np.stack([theta_of(p, mu) for mu in mus])
    ub, info = eng.ctx.reduced_solve_batches(thetas, B, rhs)
    assert tuple(ub.shape) == (eng.S, N, 70)
    for m in (0, 63, 64, 69):
        ref = np.stack(rd.solve(mus[m]))
        assert np.linalg.norm(ub[:, :, m].cpu().numpy() - ref) < 1e-10 * np.linalg.norm(ref)
    # the batched estimate takes the arrays of a 64-parameter call as they are (passes of 16 over the same arrays)
    ub64 = ub[:, :, :64].contiguous()
    eta64 = eng.ctx.reduced_estimate_batch(thetas[:64], ub64, buf['grams'], eng.f2, eng.ceps, eng.hdiam).cpu().numpy()
    for m in (0, 15, 16, 40, 63):
        um = ub64[:, :, m].contiguous()
        one = eng.reduced_estimate(thetas[m], um, buf['grams']).cpu().numpy()
        assert np.abs(eta64[:, :, m] - one).max() <= 1e-11 * np.abs(one).max(), m
    from pylrbms_amd._native import NativeError
    with pytest.raises(NativeError):
        eng.ctx.reduced_solve_batch(np.tile(thetas[:1], (65, 1)), B, rhs)          # more than 64 per call


def test_two_level_preconditioner_and_prebuilt_form(monkeypatch):
    """The coarse level of the reduced solvers: same solutions as the oracle's dense solves with it, without it
    (LRBMS_OPT_COARSE 0) and with a preconditioner prebuilt at ANOTHER parameter (lrbms_reduced_precond_build / _use);
    fewer iterations with it; basis size mismatch of a prebuilt preconditioner falls back to per-call ones."""
    from pylrbms_amd import multiscale_problem
    from oracle.lrbms import OracleReductor
    p = multiscale_problem.init_grid_and_problem({'num_subdomains': [8, 8], 'coarse_per_subdomain': 2})
    eng = _engine(p)
    d = oracle_from_problem(p)
    N = 5
    V = energy_orthonormalize(make_bases(d.S, d.n, N, seed=4), d)
    buf = eng.project_and_estimate(eng.ctx.from_numpy(V))
    B, rhs = buf['sys'][0], buf['sys'][1]
    rd = OracleReductor(d, [V[ii] for ii in range(d.S)]).reduce()
    mus = [0.1, 0.37, 1.0]
    thetas = np.stack([theta_of(p, mu) for mu in mus])
    refs = [np.stack(rd.solve(mu)) for mu in mus]

    def check(tag):
        ub, info_b = eng.ctx.reduced_solve_batch(thetas, B, rhs)
        its = [info_b['iterations']]
        for m in range(len(mus)):
            us, info = eng.reduced_solve(thetas[m], B, rhs)
            its.append(info['iterations'])
            assert np.linalg.norm(us.cpu().numpy() - refs[m]) < 1e-10 * np.linalg.norm(refs[m]), tag
            assert np.linalg.norm(ub[:, :, m].cpu().numpy() - refs[m]) < 1e-10 * np.linalg.norm(refs[m]), tag
        return its

    with_coarse = check('per-call two-level')
    eng.ctx.set_option('coarse', 0)
    jacobi_only = check('block-Jacobi only')
    eng.ctx.set_option('coarse', 1)
    assert max(with_coarse) < min(jacobi_only)
    pc = eng.ctx.reduced_precond_build(theta_of(p, 0.55), B)
    # the hand-written block-tridiagonal coarse inverse against rocSOLVER's dense one, and against numpy on the 5-point
    # coarse matrix itself (entries (0, 0) of the combined blocks)
    eng.ctx.set_option('coarse', 2)
    pc_lib = eng.ctx.reduced_precond_build(theta_of(p, 0.55), B)
    eng.ctx.set_option('coarse', 1)
    S = d.S
    A0inv, A0inv_lib = (x[2 + S * N * N:].reshape(S, S).cpu().numpy() for x in (pc, pc_lib))
    assert float(pc[0]) == 1.0 and float(pc_lib[0]) == 1.0
    th = theta_of(p, 0.55)
    Bh = B.cpu().numpy()
    A0 = np.zeros((S, S))
    for s_ in range(S):
        for slot, t_ in enumerate(p['grid'].neighbor_slots[s_]):
            if t_ >= 0:
                A0[s_, int(t_)] = sum(th[q] * Bh[q, s_, slot, 0, 0] for q in range(len(th)))
    ref_inv = np.linalg.inv(A0)
    assert np.abs(A0inv - ref_inv).max() < 1e-11 * np.abs(ref_inv).max()
    assert np.abs(A0inv_lib - ref_inv).max() < 1e-11 * np.abs(ref_inv).max()
    eng.ctx.reduced_precond_use(pc)
    prebuilt = check('prebuilt at mu = 0.55')
    assert max(prebuilt) < min(jacobi_only)
    # a preconditioner for another basis size is ignored (the context keys it on N)
    buf3 = eng.project_and_estimate(eng.ctx.from_numpy(np.ascontiguousarray(V[:, :, :3])))
    u3, _ = eng.reduced_solve(thetas[0], buf3['sys'][0], buf3['sys'][1])
    rd3 = OracleReductor(d, [V[ii][:, :3] for ii in range(d.S)]).reduce()
    ref3 = np.stack(rd3.solve(mus[0]))
    assert np.linalg.norm(u3.cpu().numpy() - ref3) < 1e-10 * np.linalg.norm(ref3)
    eng.ctx.reduced_precond_use(None)


@pytest.mark.parametrize('conv, okw', [
    ({'accumulate_coupling_across_q': True}, {'accumulate_coupling_across_q': True}),
    ({'oswald_zero_on_subdomain_boundary': True}, {'oswald_zero_on': 'subdomain'}),
    ({'accumulate_coupling_across_q': True, 'oswald_zero_on_subdomain_boundary': True},
     {'accumulate_coupling_across_q': True, 'oswald_zero_on': 'subdomain'}),
])
def test_open_conventions_are_switchable_in_the_kernels(conv, okw):
    """The conventions the reference tree leaves open (coupling matrices accumulated across the affine components,
    block_swipdg.py:551-565 vs :581-583; Oswald interpolant zero on the whole subdomain boundary, :108-113) as switches of the
    HIP path (lrbms_ctx_set_option), against the oracle's flags on an unsymmetric problem: assembled blocks, every projected
    array (fused in both layouts and unfused) and the estimator triple."""
    from pylrbms_amd import multiscale_problem
    from pylrbms_amd.engine import Engine
    p = multiscale_problem.init_grid_and_problem({'num_subdomains': [3, 2], 'coarse_per_subdomain': 2})
    lam = p['lambda']
    eng = Engine(p['grid'], lam['functions'], p['kappa'], p['f'], p['lambda_bar'], p['lambda_hat'], theta_bar_of(p),
                 conventions=conv).assemble()
    d = oracle_from_problem(p, **okw)
    V = energy_orthonormalize(make_bases(d.S, d.n, 5, seed=7), d)
    res = compare_all(p, eng, V, 0.3, oracle=d)
    res.pop('cg_iterations')
    bad = {k: v for k, v in res.items() if not (v < (1e-10 if k == 'u_solve' else TOL))}
    assert not bad, bad
    # the switches do change the result (the default conventions differ by far more than the tolerance)
    d0 = oracle_from_problem(p)
    if 'accumulate_coupling_across_q' in conv:
        assert abs(d.A[1] - d0.A[1]).max() > 1e-3 * abs(d0.A[1]).max()
    if 'oswald_zero_on_subdomain_boundary' in conv:
        assert abs(d.Avg[0][0] - d0.Avg[0][0]).max() > 1e-3
