"""Shared helpers of the parity tests: build the CPU oracle for a problem dict and map between the oracle's
(generic, neighbourhood-compact) layouts and the product's (template, fixed 5-slot) layouts."""
import numpy as np

from oracle.lrbms import OracleDiscretization, OracleReductor
from oracle.mesh import OracleMesh


def oracle_quadrature_of(p):
    """The oracle's QuadratureSpec with the orders the product uses by default for this problem (the reference's orders
    from the declared order of the data functions): both sides integrate every integrand with the same rule."""
    from oracle.quadrature import QuadratureSpec as OracleSpec
    from pylrbms_amd.quadrature import QuadratureSpec
    lam = p['lambda']
    return OracleSpec(**QuadratureSpec.for_problem(lam['functions'], p['f'], p['lambda_bar'], p['lambda_hat']).as_dict())


def oracle_from_problem(p, **kw):
    grid = p['grid']
    mesh = OracleMesh(grid.lower_left, grid.upper_right, grid.K, grid.P)
    lam = p['lambda']
    kw.setdefault('quad', oracle_quadrature_of(p))
    thetas = [(lambda mu, c=c: c.evaluate(mu)) for c in lam['coefficients']]
    kappa = np.asarray(getattr(p['kappa'], 'value', p['kappa']), dtype=np.float64).reshape(2, 2)
    return OracleDiscretization(mesh, lam['functions'], thetas, kappa, p['f'], p['lambda_bar'], p['lambda_hat'],
                                p['mu_bar'], p['mu_hat'], **kw)


def theta_bar_of(p):
    return np.array([c.evaluate(p['mu_bar']) for c in p['lambda']['coefficients']])


def theta_of(p, mu):
    return np.array([c.evaluate(mu) for c in p['lambda']['coefficients']])


def make_bases(S, n, N, seed=0):
    """Constant + (N-1) seeded random columns per subdomain, NOT orthonormalised."""
    V = np.empty((S, n, N))
    for ii in range(S):
        rng = np.random.default_rng(seed + ii)
        V[ii, :, 0] = 1.0
        V[ii, :, 1:] = rng.standard_normal((n, N - 1))
    return V


def energy_orthonormalize(V, d_oracle):
    """Gram-Schmidt w.r.t. the oracle's local energy product (B1) -- used to make well-conditioned test bases."""
    out = np.empty_like(V)
    for ii in range(V.shape[0]):
        P = d_oracle.block(d_oracle.energy_product, ii, ii).toarray()
        G = V[ii].T @ P @ V[ii]
        L = np.linalg.cholesky(G)
        out[ii] = np.linalg.solve(L, V[ii].T).T
    return out


def slots_of(grid, ii):
    return [int(j) for j in grid.neighbor_slots[ii]]


def expand_square(M, grid, ii, block, per_slot):
    """Oracle compact [m*b, m*b] (neighbourhood order) -> fixed-slot [5*b, 5*b] with zeros for missing slots."""
    sl = [k for k, j in enumerate(slots_of(grid, ii)) if j >= 0]
    out = np.zeros((5 * per_slot, 5 * per_slot))
    for a, ka in enumerate(sl):
        for b, kb in enumerate(sl):
            out[ka * per_slot:(ka + 1) * per_slot, kb * per_slot:(kb + 1) * per_slot] = \
                M[a * per_slot:(a + 1) * per_slot, b * per_slot:(b + 1) * per_slot]
    return out


def expand_cols(M, grid, ii, per_slot):
    """Oracle compact [..., m*b] -> fixed-slot [..., 5*b]."""
    sl = [k for k, j in enumerate(slots_of(grid, ii)) if j >= 0]
    out = np.zeros(M.shape[:-1] + (5 * per_slot,))
    for a, ka in enumerate(sl):
        out[..., ka * per_slot:(ka + 1) * per_slot] = M[..., a * per_slot:(a + 1) * per_slot]
    return out


def compact_blocks(D, QN):
    """Dense fixed-slot [5 QN, 5 QN] -> block-compact [9, QN, QN] (block 0 = [self,self], 1 + side = [a,self],
    5 + side = [a,a]) plus the largest entry of the blocks the compact layout drops (must be structurally zero)."""
    out = np.zeros((9, QN, QN))
    rest = D.copy()
    def blk(r, c):
        return slice(r * QN, (r + 1) * QN), slice(c * QN, (c + 1) * QN)
    out[0] = D[blk(2, 2)]
    rest[blk(2, 2)] = 0.0
    for side, slot in enumerate((0, 1, 3, 4)):
        out[1 + side] = D[blk(slot, 2)]
        out[5 + side] = D[blk(slot, slot)]
        rest[blk(slot, 2)] = 0.0
        rest[blk(2, slot)] = 0.0
        rest[blk(slot, slot)] = 0.0
    return out, float(np.abs(rest).max())


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(np.abs(b).max(), 1e-300)
    return float(np.abs(a - b).max() / scale)


def compare_all(p, engine, V, mu, do_solve=True, oracle=None):
    """Run the HIP path and the oracle on the same inputs; return {name: relative max error}."""
    from pylrbms_amd.engine import blockell_to_dense, coupling_to_dense
    grid = p['grid']
    t = grid.template
    d = oracle if oracle is not None else oracle_from_problem(p)
    S, n, Q = d.S, d.n, d.Q
    N = V.shape[2]
    res = {}
    eng = engine
    assert eng.S == S, 'compare_all is for single-rank engines'

    def host(x):
        return x.detach().cpu().numpy()

    # ---- assembly
    A_diag, A_cpl = host(eng.A_diag), host(eng.A_cpl)
    e_d, e_c = 0.0, 0.0
    for q in range(Q):
        for ii in range(S):
            ref = d.block(d.A[q], ii, ii).toarray()
            e_d = max(e_d, rel_err(blockell_to_dense(t, A_diag[q, ii]), ref))
            for side in range(4):
                jj = grid.neighbor_slots[ii, (0, 1, 3, 4)[side]]
                if jj >= 0:
                    refc = d.block(d.A[q], ii, int(jj)).toarray()
                    e_c = max(e_c, np.abs(coupling_to_dense(t, A_cpl[q, ii, side], side) - refc).max() /
                              max(np.abs(ref).max(), 1e-300))
    res['A_diag'], res['A_cpl'] = e_d, e_c
    res['b'] = rel_err(host(eng.b).reshape(-1), d.b)
    res['f2'] = rel_err(host(eng.f2), d.local_eta_rf_squared)
    res['ceps'] = rel_err(host(eng.ceps), d.min_diffusion_evs)
    res['hdiam'] = abs(eng.hdiam - d.subdomain_diameters[0]) / d.subdomain_diameters[0]
    P_diag = host(eng.P_diag)
    res['P_diag'] = max(rel_err(blockell_to_dense(t, P_diag[ii]), d.block(d.energy_product, ii, ii).toarray())
                        for ii in range(S))
    stiff = d.stiff.reshape(S, t.n_T, 3, 3)
    ebar = host(eng.ebar)
    res['ebar'] = max(rel_err(np.concatenate([(ebar[ii, e] * stiff[ii, e]).ravel() for e in range(t.n_T)]),
                              np.concatenate([d.block(d.elliptic_bar, ii, ii)[3 * e:3 * e + 3, 3 * e:3 * e + 3].toarray().ravel()
                                              for e in range(t.n_T)])) for ii in range(S))
    caa = host(eng.caa)
    e = 0.0
    for q in range(Q):
        for q2 in range(Q):
            for ii in range(S):
                ref = d.block(d.caa[q][q2], ii, ii)
                got = np.concatenate([(caa[q, q2, ii, el] * stiff[ii, el]).ravel() for el in range(t.n_T)])
                want = np.concatenate([ref[3 * el:3 * el + 3, 3 * el:3 * el + 3].toarray().ravel() for el in range(t.n_T)])
                e = max(e, np.abs(got - want).max() / max(np.abs(want).max(), 1e-300) if np.abs(want).max() > 0
                        else np.abs(got).max())
    res['caa'] = e
    Aab, Bbb = host(eng.Aab), host(eng.Bbb)
    e_ab, e_bb = 0.0, 0.0
    ab_asm_scale = max(max(np.abs(d.Aab[q][ii]).max() for q in range(Q) for ii in range(S)), 1e-300)
    for ii in range(S):
        Bd = np.zeros((t.n_rt, t.n_rt))
        for el in range(t.n_T):
            rt = t.elem_rt[el]
            Bd[np.ix_(rt, rt)] += Bbb[ii, el]
        e_bb = max(e_bb, rel_err(Bd, d.Bbb[ii].toarray()))
        for q in range(Q):
            Ad = np.zeros((n, t.n_rt))
            for el in range(t.n_T):
                Ad[3 * el:3 * el + 3, t.elem_rt[el]] += Aab[q, ii, el]
            want = d.Aab[q][ii].toarray()
            e_ab = max(e_ab, np.abs(Ad - want).max() / ab_asm_scale)
    res['Aab'], res['Bbb'] = e_ab, e_bb

    # ---- apply + projections
    Vd = eng.ctx.from_numpy(V)
    buf = eng.project_and_estimate(Vd, fused=False)
    fbuf = dbuf = None
    if eng.ctx.fused_supported(Q, N, factored=True):
        fbuf = eng.project_and_estimate(Vd, eng.alloc_reduce_buffers(N), fused=True)                  # factored layout (default)
        assert len(fbuf['grams']) == 8
    if eng.ctx.fused_supported(Q, N):
        dbuf = eng.project_and_estimate(Vd, eng.alloc_reduce_buffers(N, factored=False), fused=True)  # dense layout
    red = OracleReductor(d, [V[ii] for ii in range(S)])
    OI, RT = red.image_bases()
    mesh = d.mesh
    Wt, Rt = host(buf['Wt']), host(buf['Rt'])
    e_w, e_r = 0.0, 0.0
    wscale = max(np.abs(V).max(), 1e-300)
    rscale = max(max(np.abs(b).max() for blocks in RT for b in blocks), 1e-300)
    for ii in range(S):
        hood = mesh.neighborhood_of(ii)
        Wref = np.hstack([OI[kk][mesh.neighborhood_of(kk).index(ii)] for kk in hood])
        Rref = np.hstack([RT[kk][mesh.neighborhood_of(kk).index(ii)] for kk in hood])
        e_w = max(e_w, np.abs(Wt[ii] - expand_cols(Wref, grid, ii, N)).max() / wscale)
        e_r = max(e_r, np.abs(Rt[ii] - expand_cols(Rref, grid, ii, Q * N)).max() / rscale)
    res['Wt'], res['Rt'] = e_w, e_r

    rd = red.reduce()
    B_sys, rhs_red, E_red, M_red = [host(x) for x in buf['sys']]
    G_nc, r_fd, G_rdd, G_bb, G_ab, G_aa = [host(x) for x in buf['grams']]
    errs = {k: 0.0 for k in ('B_sys', 'rhs_red', 'E_red', 'M_red', 'G_nc', 'r_fd', 'G_rdd', 'G_bb', 'G_ab', 'G_aa')}
    sys_scale = max(np.abs(rd.op[ii][ii][q]).max() for ii in range(S) for q in range(Q))
    # floors: a constant basis (N = 1) has zero gradient, so df_aa / df_ab are exactly 0 up to rounding noise
    vscale = float(np.abs(V).max()) ** 2
    ab_scale = max(max(np.abs(rd.df_ab[ii][q]).max() for ii in range(S) for q in range(Q)), vscale)
    aa_scale = max(max(np.abs(rd.df_aa[ii][q][q2]).max() for ii in range(S) for q in range(Q) for q2 in range(Q)),
                   vscale)
    for ii in range(S):
        sl = slots_of(grid, ii)
        for q in range(Q):
            for k, jj in enumerate(sl):
                ref = rd.op[ii][jj][q] if jj >= 0 else np.zeros((N, N))
                errs['B_sys'] = max(errs['B_sys'], np.abs(B_sys[q, ii, k] - ref).max() / sys_scale)
            errs['G_ab'] = max(errs['G_ab'], np.abs(G_ab[q, ii] - expand_cols(rd.df_ab[ii][q], grid, ii, Q * N)).max() / ab_scale)
            for q2 in range(Q):
                errs['G_aa'] = max(errs['G_aa'], np.abs(G_aa[q, q2, ii] - rd.df_aa[ii][q][q2]).max() / aa_scale)
        errs['rhs_red'] = max(errs['rhs_red'], rel_err(rhs_red[ii], rd.rhs[ii]))
        errs['E_red'] = max(errs['E_red'], rel_err(E_red[ii], rd.energy[ii]))
        errs['M_red'] = max(errs['M_red'], rel_err(M_red[ii], rd.l2[ii]))
        errs['G_nc'] = max(errs['G_nc'], rel_err(G_nc[ii], expand_square(rd.nc[ii], grid, ii, None, N)))
        errs['r_fd'] = max(errs['r_fd'], rel_err(r_fd[ii], expand_cols(rd.r_fd[ii], grid, ii, Q * N)))
        for name, got, ref in (('G_rdd', G_rdd[ii], rd.r_dd[ii]), ('G_bb', G_bb[ii], rd.df_bb[ii])):
            dense = expand_square(ref, grid, ii, None, Q * N)
            blocks, dropped = compact_blocks(dense, Q * N)
            errs[name] = max(errs[name], rel_err(got, blocks))
            # the blocks the compact layout does not store are structurally zero in the ORACLE's dense operator
            errs[name] = max(errs[name], dropped / max(np.abs(dense).max(), 1e-300))
    res.update(errs)
    if fbuf is not None:      # the fused pass must produce the same arrays as the unfused kernels (and the oracle), in
        from pylrbms_amd.engine import expand_factored_grams      # both output layouts (factored blocks expanded for this)
        names = ('B_sys', 'rhs_red', 'E_red', 'M_red', 'G_nc', 'r_fd', 'G_rdd', 'G_bb', 'G_ab', 'G_aa')
        for tag, xb in (('fused_', fbuf), ('fused_dense_', dbuf)):
            if xb is None:      # large templates (k_c = 16) run fused in the factored layout only
                continue
            for name, a, b in zip(names, list(xb['sys']) + list(expand_factored_grams(xb['grams'])),
                                  list(buf['sys']) + list(buf['grams'])):
                a, b = host(a), host(b)
                res[tag + name] = float(np.abs(a - b).max() / max(np.abs(b).max(), vscale * 1e-3))

    # ---- online: estimate for a random coefficient vector, then solve + estimate
    theta = theta_of(p, mu)
    rng = np.random.default_rng(99)
    u = rng.standard_normal((S, N))
    eta = host(eng.reduced_estimate(theta, eng.ctx.from_numpy(u), buf['grams']))
    _, (nc, r, df), _ = rd.estimate([u[ii] for ii in range(S)], mu, decompose=True)
    res['eta_nc'], res['eta_r'], res['eta_df'] = rel_err(eta[0], nc), rel_err(eta[1], r), rel_err(eta[2], df)
    if fbuf is not None:      # the estimate on the factored layout: single-parameter kernel and the batched (MFMA) one
        eta_f = host(eng.reduced_estimate(theta, eng.ctx.from_numpy(u), fbuf['grams']))
        res['eta_factored'] = max(rel_err(eta_f[0], nc), rel_err(eta_f[1], r), rel_err(eta_f[2], df))
        ub = eng.ctx.from_numpy(np.repeat(u[:, :, None], 3, axis=2) * np.array([1.0, -0.5, 2.0])[None, None, :])
        thb = np.stack([theta, theta, theta])
        for tag, grams in (('eta_batch_factored', fbuf['grams']), ('eta_batch_dense', dbuf['grams'] if dbuf else None)):
            if grams is None:
                continue
            eb = host(eng.ctx.reduced_estimate_batch(thb, ub, grams, eng.f2, eng.ceps, eng.hdiam))
            _, (nc2, r2, df2), _ = rd.estimate([-0.5 * u[ii] for ii in range(S)], mu, decompose=True)
            res[tag] = max(rel_err(eb[0, :, 0], nc), rel_err(eb[1, :, 0], r), rel_err(eb[2, :, 0], df),
                           rel_err(eb[0, :, 1], nc2), rel_err(eb[1, :, 1], r2), rel_err(eb[2, :, 1], df2))
    if do_solve:
        u_dev, info = eng.reduced_solve(theta, buf['sys'][0], buf['sys'][1])
        u_ref = np.stack(rd.solve(mu))
        res['u_solve'] = float(np.linalg.norm(host(u_dev) - u_ref) / np.linalg.norm(u_ref))
        res['cg_iterations'] = info['iterations']
    return res
