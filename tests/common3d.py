"""Shared helpers of the 3D / P2 parity tests (BASELINE.json config 5): problems, the CPU oracle for them, and the maps between
the oracle's generic layouts (global sparse matrices, neighbourhood-compact dense operators) and the product's template layouts."""
import numpy as np

from oracle.lrbms3d import Discretization3D, Reductor3D
from oracle.mesh3d import KuhnMesh3D

KAPPA_ANISO = np.array([[1.0, 0.1, 0.0], [0.1, 1.5, 0.2], [0.0, 0.2, 0.8]])


def _one(x):
    return 1.0 + 0.0 * x[..., 0]


def _lam1(x):
    return x[..., 0] * x[..., 1] + 0.5 + 0.3 * np.sin(3.0 * x[..., 2])


def _lam2(x):
    return 1.0 + 0.5 * np.cos(2.0 * x[..., 0] + x[..., 1]) * x[..., 2]


def _f(x):
    return 1.0 + x[..., 2] + np.cos(2.0 * x[..., 0])


def _lbar(x):
    return 1.0 + 0.5 * _lam1(x)


PROBLEMS = {
    # name: (P, kc, lambdas, thetas, kappa, N, mu)
    'aniso_2x2x1': ([2, 2, 1], 2, [_one, _lam1], [lambda mu: 1.0, lambda mu: mu], KAPPA_ANISO, 4, 0.3),
    'interior_3x3x3': ([3, 3, 3], 1, [_one, _lam1], [lambda mu: 1.0, lambda mu: mu], np.eye(3), 5, 0.7),
    'q1_strip': ([3, 1, 1], 2, [_lam2], [lambda mu: mu], np.eye(3), 3, 1.3),
    'q3_2x1x2': ([2, 1, 2], 1, [_one, _lam1, _lam2], [lambda mu: 1.0, lambda mu: mu, lambda mu: mu * mu], KAPPA_ANISO, 6, 0.6),
    'wide_basis': ([2, 1, 1], 2, [_one, _lam1], [lambda mu: 1.0, lambda mu: mu], np.eye(3), 30, 0.4),
    # unequal cubes per direction: sides with fewer faces / nodes than the padded tables hold, odd cube counts in the traversal
    'kc_3x1x2': ([2, 2, 2], (3, 1, 2), [_one, _lam1], [lambda mu: 1.0, lambda mu: mu], KAPPA_ANISO, 4, 0.8),
    # the template and basis size of BASELINE.json config 5 itself (k_c = 4: n = 3 840, n_bf = 192, n_b = 386; Q = 2, N = 30), on
    # four subdomains: the kernel instantiations the benchmark times
    'cfg5_template': ([2, 1, 2], 4, [_one, _lam1], [lambda mu: 1.0, lambda mu: mu], np.eye(3), 30, 0.45),
}


def make_problem(name):
    P, kc, lams, thetas, kappa, N, mu = PROBLEMS[name]
    from pylrbms_amd.grid3d import make_grid3d
    grid = make_grid3d(num_subdomains=P, cubes_per_subdomain_and_dim=kc, kappa=kappa)
    return dict(name=name, grid=grid, lambdas=lams, thetas=thetas, kappa=kappa, f=_f, lambda_bar=_lbar, lambda_hat=_lbar,
                mu_bar=0.5, mu_hat=0.5, N=N, mu=mu, P=P, kc=kc)


def oracle_of(p):
    mesh = KuhnMesh3D(np.asarray(p['P']) * np.asarray(p['kc']), p['P'])
    return Discretization3D(mesh, p['lambdas'], p['thetas'], p['kappa'], p['f'], p['lambda_bar'], p['lambda_hat'], p['mu_bar'],
                            p['mu_hat'])


def make_bases3d(S, n, N, seed=0):
    V = np.empty((S, n, N))
    for ii in range(S):
        rng = np.random.default_rng(seed + ii)
        V[ii, :, 0] = 1.0
        V[ii, :, 1:] = rng.standard_normal((n, N - 1))
        V[ii] = np.linalg.qr(V[ii])[0]          # orthonormal columns (first one the constant): well-conditioned reduced systems
    return V


def theta_of(p, mu):
    return np.array([t(mu) for t in p['thetas']], dtype=np.float64)


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def oracle_assembled(p, d):
    """The oracle's assembled quantities in the product's template layouts (dict of numpy arrays)."""
    grid, t, m = p['grid'], p['grid'].template, d.mesh
    S, nT, Q = grid.num_subdomains, t.n_T, d.Q
    from pylrbms_amd.grid3d import SIDE_TO_SLOT
    out = dict(A_diag=np.zeros((Q, S, nT, 5, 10, 10)), A_cpl=np.zeros((Q, S, 6, t.ncf, 10, 10)), Cf=np.zeros((Q, S, nT, 4, 10)),
               ebar=np.zeros((S, nT, 10, 10)), Aaa=np.zeros((Q, Q, S, nT, 10, 10)), P_diag=np.zeros((S, nT, 5, 10, 10)))
    Pm = d.P.tocsr()                  # local energy product at mu_bar: block diagonal over the subdomains
    for s in range(S):
        dofs = d.dofs_of(s)
        own = Pm[dofs][:, dofs].toarray().reshape(nT, 10, nT, 10)
        off = Pm[dofs].copy()
        off = off.tolil()
        off[:, dofs] = 0.0
        assert abs(off.tocsr()).max() == 0.0, 'the local energy product couples subdomains'
        for e in range(nT):
            out['P_diag'][s, e, 0] = own[e, :, e, :]
            for f in range(4):
                nb = t.nb_elem[e, f]
                if nb >= 0:
                    out['P_diag'][s, e, 1 + f] = own[e, :, nb, :]
    for q in range(Q):
        A = d.A_q[q].tocsr()
        F = d.F_q[q].tocsr()
        for s in range(S):
            dofs, el = d.dofs_of(s), m.elements_of(s)
            As = A[dofs].tocsc()
            own = As[:, dofs].toarray().reshape(nT, 10, nT, 10)
            for e in range(nT):
                out['A_diag'][q, s, e, 0] = own[e, :, e, :]
                for f in range(4):
                    nb = t.nb_elem[e, f]
                    if nb >= 0:
                        out['A_diag'][q, s, e, 1 + f] = own[e, :, nb, :]
                    else:
                        side = -(nb + 1)
                        s2 = grid.neighbor_slots[s, SIDE_TO_SLOT[side]]
                        if s2 >= 0:
                            eo = t.nb_out[e, f]
                            cols = d.dofs_of(s2)[10 * eo:10 * eo + 10]
                            out['A_cpl'][q, s, side, t.face_pos[e, f]] = As[10 * e:10 * e + 10][:, cols].toarray()
                    fid = m.elem_face[el[e], f]
                    out['Cf'][q, s, e, f] = m.elem_face_sign[el[e], f] * F[fid][:, dofs[10 * e:10 * e + 10]].toarray().ravel()
    E = d.E.tocsr()
    for s in range(S):
        dofs, el = d.dofs_of(s), m.elements_of(s)
        blk = E[dofs][:, dofs].toarray().reshape(nT, 10, nT, 10)
        out['ebar'][s] = blk[np.arange(nT), :, np.arange(nT), :]
        for q in range(Q):
            for q2 in range(Q):
                out['Aaa'][q, q2, s] = d.Aaa[q][q2][el]
    out['Aab'] = np.stack([d.Aab[q].reshape(S, nT, 10, 4) for q in range(Q)])
    out['Bbb'] = d.Bbb.reshape(S, nT, 4, 4)
    out['b'] = d.b.reshape(S, t.n)
    out['f2'], out['ceps'], out['bdiv'] = d.f2, d.ceps, d.bdiv.reshape(S, nT)
    return out


def oracle_dense_blocks(p, d, rd, ii):
    """The oracle's projected operators of subdomain ii, columns re-indexed to the product's 7 slots (zero where there is no
    neighbour): dict with G_nc [7N, 7N], G_bb / G_rdd [7QN, 7QN], r_fd [7QN], G_ab [Q, N, 7QN], G_aa [Q, Q, N, N], B_sys [Q, 7, N, N]."""
    grid = p['grid']
    N, Q = p['N'], d.Q
    QN = Q * N
    slots = [int(v) for v in grid.neighbor_slots[ii]]
    hood = rd.hood[ii]
    idx_n = np.concatenate([np.arange(slots.index(kk) * N, (slots.index(kk) + 1) * N) for kk in hood])
    idx_c = np.concatenate([np.arange(slots.index(kk) * QN, (slots.index(kk) + 1) * QN) for kk in hood])
    out = dict(G_nc=np.zeros((7 * N, 7 * N)), G_bb=np.zeros((7 * QN, 7 * QN)), G_rdd=np.zeros((7 * QN, 7 * QN)), r_fd=np.zeros(7 * QN),
               G_ab=np.zeros((Q, N, 7 * QN)), G_aa=np.zeros((Q, Q, N, N)), B_sys=np.zeros((Q, 7, N, N)))
    out['G_nc'][np.ix_(idx_n, idx_n)] = rd.nc[ii]
    out['G_bb'][np.ix_(idx_c, idx_c)] = rd.df_bb[ii]
    out['G_rdd'][np.ix_(idx_c, idx_c)] = rd.r_dd[ii]
    out['r_fd'][idx_c] = rd.r_fd[ii]
    for q in range(Q):
        out['G_ab'][q][:, idx_c] = rd.df_ab[ii][q]
        for q2 in range(Q):
            out['G_aa'][q, q2] = rd.df_aa[ii][q][q2]
        for jj, blocks in rd.op[ii].items():
            out['B_sys'][q, slots.index(jj)] = blocks[q]
    return out


def reduce_with_oracle(p, d, V):
    return Reductor3D(d, [V[ii] for ii in range(d.S)]).reduce()
