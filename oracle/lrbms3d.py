"""3D / P2 CPU oracle of the LRBMS hot path (BASELINE.json config 5) -- TEST INFRASTRUCTURE ONLY.

The dimension- and order-independent formulas of SURVEY.md App. A (SWIPDG bilinear form A.2 with
``sigma_p = 20 / 38`` for ``p = 2`` and ``beta = 1 / (d - 1) = 1 / 2``, products A.3, Oswald interpolation error A.4,
RT0 diffusive-flux reconstruction A.5, estimator operators A.6, estimate A.7, projection A.8) restated for P2
discontinuous Lagrange elements on the tetrahedral mesh of oracle/mesh3d.py.

PARITY: UNPINNED.  The reference binds the 2D P1 operators only (discretize_elliptic_block_swipdg.py:22-23; :195 uses
``x[0], x[1]``), so there is no reference output, fixture or golden value for this configuration.  The oracle is
validated by properties instead (tests/test_oracle3d.py): symmetry and positive definiteness, exact reproduction of
quadratic solutions, local conservation of the reconstructed flux, vanishing nonconformity for conforming functions,
experimental orders of convergence (energy 2, L2 3), and the identity "reduced estimate of u == full-order estimate
of its reconstruction".

Numbering: block DG mapper ``dof = 10 * element + local`` with the local order 4 vertices, then the 6 edge midpoints
(0-1, 0-2, 0-3, 1-2, 1-3, 2-3); RT0 DoFs are global face indices, restricted per subdomain in increasing order.
Quadrature: Stroud conical-product Gauss-Jacobi rules (exact to degree ``2 n - 1`` with ``n^3`` / ``n^2`` points); the
degree is chosen from the polynomial degree declared for the data functions, so polynomial data are integrated exactly.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from scipy.special import roots_jacobi

from .mesh3d import EDGE_VERTS

SIGMA_INNER_P2 = 20.0      # dune-gdt swipdg inner_sigma(polorder <= 2)     [UPSTREAM-RECALL, SURVEY App. A.2]
SIGMA_BOUNDARY_P2 = 38.0   # dune-gdt swipdg boundary_sigma(polorder <= 2)
BETA_3D = 0.5              # 1 / (d - 1)
NLOC = 10


# --------------------------------------------------------------------------------------------- quadrature, basis
def _gauss_jacobi01(n, alpha):
    """Nodes / weights on [0, 1] for the weight (1 - x)^alpha."""
    x, w = roots_jacobi(n, alpha, 0.0)
    return 0.5 * (x + 1.0), w / 2.0 ** (alpha + 1.0)


def tet_rule(degree):
    """(barycentric points [k, 4], weights [k] summing to 1) exact for polynomials of the given degree."""
    n = max(1, (int(degree) + 2) // 2)
    x0, w0 = _gauss_jacobi01(n, 2.0)
    x1, w1 = _gauss_jacobi01(n, 1.0)
    x2, w2 = _gauss_jacobi01(n, 0.0)
    a, b, c = np.meshgrid(x0, x1, x2, indexing='ij')
    w = (w0[:, None, None] * w1[None, :, None] * w2[None, None, :]).ravel()
    x = a.ravel()
    y = (b * (1.0 - a)).ravel()
    z = (c * (1.0 - a) * (1.0 - b)).ravel()
    bary = np.stack([1.0 - x - y - z, x, y, z], axis=1)
    return bary, w * 6.0            # the conical weights sum to the reference volume 1/6


def tri_rule(degree):
    """(barycentric points [k, 3], weights [k] summing to 1) on a triangle."""
    n = max(1, (int(degree) + 2) // 2)
    x0, w0 = _gauss_jacobi01(n, 1.0)
    x1, w1 = _gauss_jacobi01(n, 0.0)
    a, b = np.meshgrid(x0, x1, indexing='ij')
    w = (w0[:, None] * w1[None, :]).ravel()
    x = a.ravel()
    y = (b * (1.0 - a)).ravel()
    return np.stack([1.0 - x - y, x, y], axis=1), w * 2.0


def p2_basis(lam):
    """P2 Lagrange basis at barycentric points ``lam [..., 4]``: values ``[..., 10]`` and the derivatives with respect
    to the barycentric coordinates ``[..., 10, 4]`` (chain rule with grad lambda gives physical gradients)."""
    lam = np.asarray(lam, dtype=np.float64)
    phi = np.empty(lam.shape[:-1] + (NLOC,))
    dphi = np.zeros(lam.shape[:-1] + (NLOC, 4))
    for i in range(4):
        phi[..., i] = lam[..., i] * (2.0 * lam[..., i] - 1.0)
        dphi[..., i, i] = 4.0 * lam[..., i] - 1.0
    for k, (i, j) in enumerate(EDGE_VERTS):
        phi[..., 4 + k] = 4.0 * lam[..., i] * lam[..., j]
        dphi[..., 4 + k, i] = 4.0 * lam[..., j]
        dphi[..., 4 + k, j] = 4.0 * lam[..., i]
    return phi, dphi


def _feval(fn, x):
    out = np.asarray(fn(x), dtype=np.float64)
    return np.broadcast_to(out, x.shape[:-1]).copy()


class Discretization3D:
    """Block SWIPDG P2 discretization on a ``KuhnMesh3D`` plus everything the localized estimator needs.

    ``lambda_funcs``: callables ``x [..., 3] -> [...]`` (affine components), ``thetas``: callables ``mu -> float``;
    ``data_degree``: polynomial degree the quadrature assumes for every data function (lambda_q, f, lambda_bar,
    lambda_hat)."""

    def __init__(self, mesh, lambda_funcs, thetas, kappa, f, lambda_bar, lambda_hat, mu_bar, mu_hat, data_degree=2,
                 oswald_patch='neighborhood'):
        assert oswald_patch in ('neighborhood', 'vertex')
        self.oswald_patch = oswald_patch
        self.mesh = mesh
        self.lambda_funcs, self.thetas = list(lambda_funcs), list(thetas)
        self.Q = len(self.lambda_funcs)
        self.kappa = np.asarray(kappa, dtype=np.float64).reshape(3, 3)
        self.f, self.lambda_bar, self.lambda_hat = f, lambda_bar, lambda_hat
        self.mu_bar, self.mu_hat = mu_bar, mu_hat
        self.deg = int(data_degree)
        self.S = mesh.num_subdomains
        self.nT = mesh.elements_per_subdomain
        self.n = NLOC * self.nT
        self.ndof = NLOC * mesh.num_elements
        self._assemble_system()
        self._assemble_rhs_and_products()
        self._assemble_flux_reconstruction()
        self._assemble_oswald()
        self._assemble_estimator_operators()

    # ------------------------------------------------------------------ helpers
    def theta(self, mu):
        return np.array([t(mu) for t in self.thetas], dtype=np.float64)

    def dofs_of(self, ii):
        return np.arange(ii * self.n, (ii + 1) * self.n)

    def _vol_points(self, degree):
        m = self.mesh
        bary, w = tet_rule(degree)
        x = np.einsum('kv,evd->ekd', bary, m.vertices[m.elements])          # [nT, k, 3]
        phi, dphi = p2_basis(bary)                                          # [k, 10], [k, 10, 4]
        grad = np.einsum('kiv,eva->ekia', dphi, m.grad_lambda)              # [nT, k, 10, 3]
        return x, w, phi, grad

    def _face_side(self, elems, lfaces, xq):
        """Basis values and physical gradients of the elements ``elems`` at the physical points ``xq [F, k, 3]``."""
        m = self.mesh
        lam = m.barycentric(elems[:, None], xq)
        phi, dphi = p2_basis(lam)                                           # [F, k, 10], [F, k, 10, 4]
        grad = np.einsum('fkiv,fva->fkia', dphi, m.grad_lambda[elems])
        return phi, grad

    def _face_points(self, faces, degree):
        m = self.mesh
        bary, w = tri_rule(degree)
        xq = np.einsum('kv,fvd->fkd', bary, m.vertices[m.face_vertices[faces]])
        return xq, w

    @staticmethod
    def _scatter(rows_e, cols_e, blocks, ndof):
        """blocks [F, 10, 10] coupling test element rows_e with trial element cols_e -> COO triplets."""
        r = (NLOC * rows_e[:, None, None] + np.arange(NLOC)[None, :, None]) + 0 * np.arange(NLOC)[None, None, :]
        c = (NLOC * cols_e[:, None, None] + np.arange(NLOC)[None, None, :]) + 0 * np.arange(NLOC)[None, :, None]
        return sp.coo_matrix((blocks.ravel(), (r.ravel(), c.ravel())), shape=(ndof, ndof))

    # ------------------------------------------------------------------ SWIPDG system
    def _swipdg(self, lam_fn, penalty_only=False, subdomain_dirichlet=False):
        """Global SWIPDG matrix for the scalar factor ``lam_fn``.  ``penalty_only`` / ``subdomain_dirichlet``: the
        energy product of App. A.3 -- volume term plus the penalty part of every face term, with every face on a
        subdomain boundary treated as a Dirichlet face of that subdomain (no coupling between subdomains)."""
        m, ndof, K = self.mesh, self.ndof, self.kappa
        x, w, phi, grad = self._vol_points(self.deg + 2)
        lam = _feval(lam_fn, x)                                             # [nT, k]
        kgrad = np.einsum('ab,ekib->ekia', K, grad)
        vol = np.einsum('k,e,ek,ekia,ekja->eij', w, m.volume, lam, grad, kgrad)
        e_all = np.arange(m.num_elements)
        A = self._scatter(e_all, e_all, vol, ndof).tocsr()
        deg_f = self.deg + 4
        inner = np.nonzero(~m.face_is_boundary)[0]
        if subdomain_dirichlet:
            inner = inner[~m.face_is_coupling[inner]]
        # ---- inner faces
        if len(inner):
            xq, wq = self._face_points(inner, deg_f)
            Em, Ep = m.face_minus[inner, 0], m.face_plus[inner, 0]
            nrm, area = m.face_normal[inner], m.face_area[inner]
            pm, gm = self._face_side(Em, None, xq)
            pp, gp = self._face_side(Ep, None, xq)
            lm = _feval(lam_fn, xq)                                         # continuous data: lambda^- = lambda^+
            delta = np.einsum('fa,ab,fb->f', nrm, K, nrm)                   # kappa constant: delta^+ = delta^-
            gamma, wgt = 0.5 * delta, 0.5
            sigma = lm * SIGMA_INNER_P2 * gamma[:, None] / area[:, None] ** BETA_3D            # 1/2 (l^- + l^+) = l
            Dm = np.einsum('fk,ab,fkib,fa->fki', lm, K, gm, nrm)            # lambda kappa grad phi^- . n
            Dp = np.einsum('fk,ab,fkib,fa->fki', lm, K, gp, nrm)
            ww = wq[None, :] * area[:, None]
            c = 0.0 if penalty_only else 1.0
            # test i, trial j (App. A.2: en/en, en/ne, ne/en, ne/ne)
            mm = np.einsum('fk,fkij->fij', ww, c * (-wgt * pm[..., :, None] * Dm[..., None, :] - wgt * Dm[..., :, None] * pm[..., None, :])
                           + sigma[..., None, None] * pm[..., :, None] * pm[..., None, :])
            mp = np.einsum('fk,fkij->fij', ww, c * (-wgt * pm[..., :, None] * Dp[..., None, :] + wgt * Dm[..., :, None] * pp[..., None, :])
                           - sigma[..., None, None] * pm[..., :, None] * pp[..., None, :])
            pmn = np.einsum('fk,fkij->fij', ww, c * (wgt * pp[..., :, None] * Dm[..., None, :] - wgt * Dp[..., :, None] * pm[..., None, :])
                            - sigma[..., None, None] * pp[..., :, None] * pm[..., None, :])
            ppn = np.einsum('fk,fkij->fij', ww, c * (wgt * pp[..., :, None] * Dp[..., None, :] + wgt * Dp[..., :, None] * pp[..., None, :])
                            + sigma[..., None, None] * pp[..., :, None] * pp[..., None, :])
            A = A + (self._scatter(Em, Em, mm, ndof) + self._scatter(Em, Ep, mp, ndof) + self._scatter(Ep, Em, pmn, ndof)
                     + self._scatter(Ep, Ep, ppn, ndof)).tocsr()
        # ---- Dirichlet faces: the physical boundary, and (energy product) both sides of every coupling face
        sides = [(np.nonzero(m.face_is_boundary)[0], 0, 1.0)]
        if subdomain_dirichlet:
            cpl = np.nonzero(m.face_is_coupling)[0]
            sides += [(cpl, 0, 1.0), (cpl, 1, -1.0)]
        for faces, side, sgn in sides:
            if not len(faces):
                continue
            xq, wq = self._face_points(faces, deg_f)
            E = (m.face_minus if side == 0 else m.face_plus)[faces, 0]
            nrm, area = sgn * m.face_normal[faces], m.face_area[faces]
            ph, gr = self._face_side(E, None, xq)
            lm = _feval(lam_fn, xq)
            delta = np.einsum('fa,ab,fb->f', nrm, K, nrm)
            sigma = lm * SIGMA_BOUNDARY_P2 * delta[:, None] / area[:, None] ** BETA_3D
            D = np.einsum('fk,ab,fkib,fa->fki', lm, K, gr, nrm)
            ww = wq[None, :] * area[:, None]
            c = 0.0 if penalty_only else 1.0
            blk = np.einsum('fk,fkij->fij', ww, c * (-ph[..., :, None] * D[..., None, :] - D[..., :, None] * ph[..., None, :])
                            + sigma[..., None, None] * ph[..., :, None] * ph[..., None, :])
            A = A + self._scatter(E, E, blk, ndof).tocsr()
        return A

    def _assemble_system(self):
        self.A_q = [self._swipdg(fn) for fn in self.lambda_funcs]

    def system_matrix(self, mu):
        th = self.theta(mu)
        return sum(t * A for t, A in zip(th, self.A_q)).tocsr()

    def solve(self, mu):
        return spla.spsolve(self.system_matrix(mu).tocsc(), self.b)

    # ------------------------------------------------------------------ RHS and products (App. A.3)
    def _assemble_rhs_and_products(self):
        m = self.mesh
        x, w, phi, grad = self._vol_points(self.deg + 4)
        fv = _feval(self.f, x)
        self.b = np.einsum('k,e,ek,ki->ei', w, m.volume, fv, phi).ravel()
        mass = np.einsum('k,e,ki,kj->eij', w, m.volume, phi, phi)
        e_all = np.arange(m.num_elements)
        self.M = self._scatter(e_all, e_all, mass, self.ndof).tocsr()
        lb = _feval(self.lambda_bar, x)
        kgrad = np.einsum('ab,ekib->ekia', self.kappa, grad)
        ebar = np.einsum('k,e,ek,ekia,ekja->eij', w, m.volume, lb, grad, kgrad)
        self.E = self._scatter(e_all, e_all, ebar, self.ndof).tocsr()      # E_ii = its diagonal block of subdomain ii
        th_bar = self.theta(self.mu_bar)
        # local energy product: theta_q(mu_bar) [volume + penalties of the faces in / on the boundary of the subdomain]
        self.P = sum(t * self._swipdg(fn, penalty_only=True, subdomain_dirichlet=True)
                     for t, fn in zip(th_bar, self.lambda_funcs)).tocsr()
        self.f2 = np.array([np.einsum('k,e,ek->', w, m.volume[m.elements_of(ii)], fv[m.elements_of(ii)] ** 2)
                            for ii in range(self.S)])
        lh = _feval(self.lambda_hat, x)
        kmin = np.linalg.eigvalsh(0.5 * (self.kappa + self.kappa.T)).min()
        self.ceps = np.array([lh[m.elements_of(ii)].min() * kmin for ii in range(self.S)])
        self.hdiam = m.subdomain_diameter

    def dirichlet_rhs(self, g, mu):
        """Boundary part of the right-hand side for inhomogeneous Dirichlet data ``u = g`` on the physical boundary:
        ``sum_q theta_q int_e (-lambda_q kappa grad phi_i . n + sigma_e phi_i) g`` (the reference problems use g = 0; this
        exists for the polynomial-reproduction test of the oracle)."""
        m, K, th = self.mesh, self.kappa, self.theta(mu)
        bnd = np.nonzero(m.face_is_boundary)[0]
        xq, wq = self._face_points(bnd, self.deg + 4)
        E = m.face_minus[bnd, 0]
        nrm, area = m.face_normal[bnd], m.face_area[bnd]
        ph, gr = self._face_side(E, None, xq)
        gv = _feval(g, xq)
        delta = np.einsum('fa,ab,fb->f', nrm, K, nrm)
        out = np.zeros(self.ndof)
        for t, fn in zip(th, self.lambda_funcs):
            lm = _feval(fn, xq)
            sigma = lm * SIGMA_BOUNDARY_P2 * delta[:, None] / area[:, None] ** BETA_3D
            D = np.einsum('fk,ab,fkib,fa->fki', lm, K, gr, nrm)
            loc = np.einsum('fk,fk,fki->fi', wq[None, :] * area[:, None], gv, -D + sigma[..., None] * ph)
            np.add.at(out, (NLOC * E[:, None] + np.arange(NLOC)[None, :]).ravel(), t * loc.ravel())
        return out

    # ------------------------------------------------------------------ RT0 flux reconstruction (App. A.5)
    def _assemble_flux_reconstruction(self):
        """F_q [num_faces, ndof]: RT0 DoF of face e (mean normal flux, orientation = outward normal of the minus
        element) of the SWIPDG numerical flux of a DG function."""
        m, K = self.mesh, self.kappa
        self.F_q = []
        deg_f = self.deg + 3
        for fn in self.lambda_funcs:
            rows, cols, vals = [], [], []
            inner = np.nonzero(~m.face_is_boundary)[0]
            xq, wq = self._face_points(inner, deg_f)
            Em, Ep = m.face_minus[inner, 0], m.face_plus[inner, 0]
            nrm, area = m.face_normal[inner], m.face_area[inner]
            pm, gm = self._face_side(Em, None, xq)
            pp, gp = self._face_side(Ep, None, xq)
            lm = _feval(fn, xq)
            delta = np.einsum('fa,ab,fb->f', nrm, K, nrm)
            sigma = lm * SIGMA_INNER_P2 * 0.5 * delta[:, None] / area[:, None] ** BETA_3D
            Dm = np.einsum('fk,ab,fkib,fa->fki', lm, K, gm, nrm)
            Dp = np.einsum('fk,ab,fkib,fa->fki', lm, K, gp, nrm)
            cm = np.einsum('k,fki->fi', wq, -0.5 * Dm + sigma[..., None] * pm)       # face MEAN: weights sum to 1
            cp = np.einsum('k,fki->fi', wq, -0.5 * Dp - sigma[..., None] * pp)
            for E, cfs in ((Em, cm), (Ep, cp)):
                rows.append(np.repeat(inner, NLOC))
                cols.append((NLOC * E[:, None] + np.arange(NLOC)[None, :]).ravel())
                vals.append(cfs.ravel())
            bnd = np.nonzero(m.face_is_boundary)[0]
            xq, wq = self._face_points(bnd, deg_f)
            E = m.face_minus[bnd, 0]
            nrm, area = m.face_normal[bnd], m.face_area[bnd]
            ph, gr = self._face_side(E, None, xq)
            lm = _feval(fn, xq)
            delta = np.einsum('fa,ab,fb->f', nrm, K, nrm)
            sigma = lm * SIGMA_BOUNDARY_P2 * delta[:, None] / area[:, None] ** BETA_3D
            D = np.einsum('fk,ab,fkib,fa->fki', lm, K, gr, nrm)
            cb = np.einsum('k,fki->fi', wq, -D + sigma[..., None] * ph)
            rows.append(np.repeat(bnd, NLOC))
            cols.append((NLOC * E[:, None] + np.arange(NLOC)[None, :]).ravel())
            vals.append(cb.ravel())
            self.F_q.append(sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                                          shape=(m.num_faces, self.ndof)).tocsr())

    def flux_reconstruction(self, u, mu):
        th = self.theta(mu)
        return sum(t * (F @ u) for t, F in zip(th, self.F_q))

    def rt0_values(self, e, lam, r):
        """Value of the RT0 function with face DoFs ``r`` (global) in element ``e`` at barycentric ``lam [k, 4]``:
        ``sum_f s_f r_f |f| / (3 |T|) (x - p_f)``, ``p_f`` the vertex opposite face ``f``."""
        m = self.mesh
        X = m.vertices[m.elements[e]]
        x = lam @ X
        out = np.zeros((len(lam), 3))
        for f in range(4):
            fid = m.elem_face[e, f]
            out += m.elem_face_sign[e, f] * r[fid] * m.face_area[fid] / (3.0 * m.volume[e]) * (x - X[f])
        return out

    # ------------------------------------------------------------------ Oswald interpolation (App. A.4)
    def _assemble_oswald(self):
        """I_os [ndof, ndof]: DG coefficients of the Oswald interpolant -- at every Lagrange node the arithmetic mean of
        the values of the elements sharing it, zero on the physical (Dirichlet) boundary.  ``oswald_patch``:
        'neighborhood' (default, as oracle/lrbms.py: block_swipdg.py:91-102 computes the interpolant on the neighbourhood
        grid view): for a node of subdomain ``ii`` only elements of ``neighborhood_of(ii)`` = ii + its FACE neighbours
        count, so subdomains that touch ``ii`` in an edge or a vertex only do not enter; 'vertex': every element at the
        node."""
        m = self.mesh
        nodes = m.elem_nodes.ravel()                                 # node of every DG dof
        dof_sub = np.repeat(m.elem_subdomain, NLOC)
        if self.oswald_patch == 'vertex':
            cnt = np.bincount(nodes, minlength=m.num_nodes).astype(np.float64)
            inv = np.where(m.node_on_boundary, 0.0, 1.0 / cnt)
            avg = sp.coo_matrix((inv[nodes], (nodes, np.arange(self.ndof))), shape=(m.num_nodes, self.ndof)).tocsr()
            spread = sp.coo_matrix((np.ones(self.ndof), (np.arange(self.ndof), nodes)), shape=(self.ndof, m.num_nodes)).tocsr()
            self.I_os = (spread @ avg).tocsr()
            return
        rows, cols, vals = [], [], []
        order = np.argsort(nodes, kind='stable')
        start = np.searchsorted(nodes[order], np.arange(m.num_nodes + 1))
        hood = [set(m.neighborhood_of(ii)) for ii in range(self.S)]
        for ii in range(self.S):
            allowed = np.isin(dof_sub, list(hood[ii]))
            for dof in self.dofs_of(ii):
                g = nodes[dof]
                if m.node_on_boundary[g]:
                    continue
                src = order[start[g]:start[g + 1]]
                src = src[allowed[src]]
                rows.append(np.full(len(src), dof))
                cols.append(src)
                vals.append(np.full(len(src), 1.0 / len(src)))
        self.I_os = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                                  shape=(self.ndof, self.ndof)).tocsr()

    def oswald_error(self, u):
        return u - self.I_os @ u

    # ------------------------------------------------------------------ estimator operators (App. A.6)
    def _assemble_estimator_operators(self):
        """Per element (block-diagonal over elements, hence over subdomains):
        Bbb [4 faces x 4 faces]  int (lambda_hat kappa)^-1 psi_f . psi_g
        Aab_q [10 x 4]           int (lambda_hat kappa)^-1 (lambda_q kappa grad phi_i) . psi_f
        Aaa_qq' [10 x 10]        int (lambda_hat kappa)^-1 (lambda_q kappa grad phi_i) . (lambda_q' kappa grad phi_j)
        div [4]                  s_f |f| / |T|       (div psi_f, constant)
        bdiv                     int f             (so that r_fd = sum_T bdiv_T div r|_T)"""
        m, K = self.mesh, self.kappa
        x, w, phi, grad = self._vol_points(3 * self.deg + 4)
        Kinv = np.linalg.inv(K)
        lh = _feval(self.lambda_hat, x)
        X = m.vertices[m.elements]                                                     # [nT, 4, 3]
        coef = m.elem_face_sign * m.face_area[m.elem_face] / (3.0 * m.volume[:, None])  # [nT, 4]
        psi = coef[:, None, :, None] * (x[:, :, None, :] - X[:, None, :, :])          # [nT, k, 4, 3]
        wv = w[None, :] * m.volume[:, None] / lh                                       # weights / lambda_hat
        kpsi = np.einsum('ab,ekfb->ekfa', Kinv, psi)
        self.Bbb = np.einsum('ek,ekfa,ekga->efg', wv, psi, kpsi)
        self.Aab, self.lam_q = [], []
        for fn in self.lambda_funcs:
            lq = _feval(fn, x)
            self.lam_q.append(lq)
            # (lambda_hat kappa)^-1 (lambda_q kappa grad phi) . psi = lambda_q / lambda_hat grad phi . psi
            self.Aab.append(np.einsum('ek,ek,ekia,ekfa->eif', wv, lq, grad, psi))
        kgrad = np.einsum('ab,ekib->ekia', K, grad)
        self.Aaa = [[np.einsum('ek,ek,ek,ekia,ekja->eij', wv, self.lam_q[q], self.lam_q[q2], grad, kgrad)
                     for q2 in range(self.Q)] for q in range(self.Q)]
        self.div = m.elem_face_sign * m.face_area[m.elem_face] / m.volume[:, None]
        fv = _feval(self.f, x)
        self.bdiv = np.einsum('k,e,ek->e', w, m.volume, fv)

    # ------------------------------------------------------------------ estimate (App. A.7, estimators.py:45-112)
    def alpha(self, mu, mu2):
        return self.thetas[0](mu) / self.thetas[0](mu2)                    # as written: first component only (App. B-2)

    def gamma(self, mu, mu2):
        return max(t(mu) / t(mu2) for t in self.thetas)

    def local_terms(self, u, mu):
        """Squared local quantities (nc, r, df) per subdomain for the global DG vector ``u``."""
        m = self.mesh
        th = self.theta(mu)
        w = self.oswald_error(u)
        Ew = self.E @ w
        r = self.flux_reconstruction(u, mu)
        U = u.reshape(m.num_elements, NLOC)
        R = r[m.elem_face]                                                 # [nT, 4]
        divr = np.einsum('ef,ef->e', self.div, R)
        df_e = np.einsum('ef,efg,eg->e', R, self.Bbb, R)
        for q in range(self.Q):
            df_e += 2.0 * th[q] * np.einsum('ei,eif,ef->e', U, self.Aab[q], R)
            for q2 in range(self.Q):
                df_e += th[q] * th[q2] * np.einsum('ei,eij,ej->e', U, self.Aaa[q][q2], U)
        nc, rr, df = np.zeros(self.S), np.zeros(self.S), np.zeros(self.S)
        for ii in range(self.S):
            d, el = self.dofs_of(ii), m.elements_of(ii)
            nc[ii] = w[d] @ Ew[d]
            r_fd = self.bdiv[el] @ divr[el]
            r_dd = m.volume[el] @ divr[el] ** 2
            rr[ii] = (self.f2[ii] - 2.0 * r_fd + r_dd) * (1.0 / np.pi ** 2) / self.ceps[ii] * self.hdiam ** 2
            df[ii] = df_e[el].sum()
        return nc, rr, df

    def estimate(self, u, mu, decompose=False):
        nc, r, df = self.local_terms(u, mu)
        a_bar, a_hat, g_bar = self.alpha(mu, self.mu_bar), self.alpha(mu, self.mu_hat), self.gamma(mu, self.mu_bar)
        eta = (1.0 / np.sqrt(a_bar)) * (np.sqrt(g_bar) * np.linalg.norm(nc) + (1.0 / np.sqrt(a_hat)) * np.linalg.norm(r + df))
        ind = (2.0 / a_bar) * (g_bar * nc ** 2 + (1.0 / a_hat) * (r + df) ** 2)
        return (eta, (nc, r, df), ind) if decompose else eta

    # ------------------------------------------------------------------ norms
    def energy_norm2(self, v, mu):
        return float(v @ (self.system_matrix(mu) @ v))

    def l2_norm2(self, v):
        return float(v @ (self.M @ v))

    def interpolate(self, fn):
        """P2 nodal interpolant as a DG vector."""
        m = self.mesh
        return _feval(fn, m.node_coords[m.elem_nodes]).ravel()


class Reductor3D:
    """Galerkin projection onto local bases ``V_ii [n, N_ii]`` and the projected estimator operators (App. A.8, A.6;
    reference reductor.py:33-73): the image bases under the Oswald interpolation error and the flux reconstruction live on
    the neighbourhood of the source subdomain, and every operator of subdomain ``ii`` is projected through the images of
    the bases of ``neighborhood_of(ii)`` on ``ii``."""

    def __init__(self, d, bases):
        self.d, self.bases = d, [np.asarray(b, dtype=np.float64) for b in bases]
        assert len(self.bases) == d.S and all(b.shape[0] == d.n for b in self.bases)

    def _embed(self, ii):
        out = np.zeros((self.d.ndof, self.bases[ii].shape[1]))
        out[self.d.dofs_of(ii)] = self.bases[ii]
        return out

    def reduce(self, subdomains=None):
        """``subdomains``: only these TARGET subdomains (bench.py farms them over a process pool to time the oracle on all host
        cores; the result then holds their entries only, in that order, and cannot be solved) -- default: all."""
        d, m = self.d, self.d.mesh
        S, Q = d.S, d.Q
        targets = list(range(S)) if subdomains is None else [int(ii) for ii in subdomains]
        need = sorted({kk for ii in targets for kk in m.neighborhood_of(ii)})
        emb = {ii: self._embed(ii) for ii in need}
        W = {ii: d.oswald_error(emb[ii]) for ii in need}                                   # [ndof, N_ii]
        R = {ii: [d.F_q[q] @ emb[ii] for q in range(Q)] for ii in need}                    # [num_faces, N_ii] per q
        rd = ReducedModel3D(self)
        rd.op = [{jj: [self.bases[ii].T @ (d.A_q[q][d.dofs_of(ii)][:, d.dofs_of(jj)] @ self.bases[jj]) for q in range(Q)]
                  for jj in m.neighborhood_of(ii)} for ii in targets]
        rd.rhs = [self.bases[ii].T @ d.b[d.dofs_of(ii)] for ii in targets]
        rd.hood = [m.neighborhood_of(ii) for ii in targets]
        rd.nc, rd.r_fd, rd.r_dd, rd.df_bb, rd.df_ab, rd.df_aa = [], [], [], [], [], []
        for k, ii in enumerate(targets):
            hood, dof, el = rd.hood[k], d.dofs_of(ii), m.elements_of(ii)
            Wi = np.hstack([W[kk][dof] for kk in hood])                                    # images on ii, slot-major columns
            rd.nc.append(Wi.T @ (d.E[dof][:, dof] @ Wi))
            # flux images on ii: columns (slot, q, j)
            Rloc = np.hstack([np.hstack([R[kk][q] for q in range(Q)]) for kk in hood])     # [num_faces, sum_k Q N_k]
            Re = Rloc[m.elem_face[el]]                                                     # [nT, 4, C]
            divr = np.einsum('ef,efc->ec', d.div[el], Re)
            rd.r_fd.append(d.bdiv[el] @ divr)
            rd.r_dd.append(np.einsum('e,ec,ed->cd', m.volume[el], divr, divr))
            rd.df_bb.append(np.einsum('efc,efg,egd->cd', Re, d.Bbb[el], Re))
            Ue = self.bases[ii].reshape(d.nT, NLOC, -1)
            rd.df_ab.append([np.einsum('eia,eif,efc->ac', Ue, d.Aab[q][el], Re) for q in range(Q)])
            rd.df_aa.append([[np.einsum('eia,eij,ejb->ab', Ue, d.Aaa[q][q2][el], Ue) for q2 in range(Q)] for q in range(Q)])
        return rd

    def reconstruct(self, u):
        return np.concatenate([self.bases[ii] @ u[ii] for ii in range(self.d.S)])


class ReducedModel3D:
    def __init__(self, reductor):
        self.reductor, self.d = reductor, reductor.d

    def solve(self, mu):
        d, th = self.d, self.d.theta(mu)
        sizes = [b.shape[1] for b in self.reductor.bases]
        off = np.concatenate([[0], np.cumsum(sizes)])
        A = np.zeros((off[-1], off[-1]))
        for ii in range(d.S):
            for jj, blocks in self.op[ii].items():
                A[off[ii]:off[ii + 1], off[jj]:off[jj + 1]] = sum(t * B for t, B in zip(th, blocks))
        u = np.linalg.solve(A, np.concatenate(self.rhs))
        return [u[off[ii]:off[ii + 1]] for ii in range(d.S)]

    def local_terms(self, u, mu):
        d, th = self.d, self.d.theta(mu)
        nc, rr, df = np.zeros(d.S), np.zeros(d.S), np.zeros(d.S)
        for ii in range(d.S):
            uo = np.concatenate([u[kk] for kk in self.hood[ii]])
            ur = np.concatenate([np.concatenate([th[q] * u[kk] for q in range(d.Q)]) for kk in self.hood[ii]])
            nc[ii] = uo @ self.nc[ii] @ uo
            rr[ii] = (d.f2[ii] - 2.0 * self.r_fd[ii] @ ur + ur @ self.r_dd[ii] @ ur) * (1.0 / np.pi ** 2) / d.ceps[ii] * d.hdiam ** 2
            val = ur @ self.df_bb[ii] @ ur
            for q in range(d.Q):
                val += 2.0 * th[q] * (u[ii] @ self.df_ab[ii][q] @ ur)
                for q2 in range(d.Q):
                    val += th[q] * th[q2] * (u[ii] @ self.df_aa[ii][q][q2] @ u[ii])
            df[ii] = val
        return nc, rr, df

    def estimate(self, u, mu, decompose=False):
        d = self.d
        nc, r, df = self.local_terms(u, mu)
        a_bar, a_hat, g_bar = d.alpha(mu, d.mu_bar), d.alpha(mu, d.mu_hat), d.gamma(mu, d.mu_bar)
        eta = (1.0 / np.sqrt(a_bar)) * (np.sqrt(g_bar) * np.linalg.norm(nc) + (1.0 / np.sqrt(a_hat)) * np.linalg.norm(r + df))
        return (eta, (nc, r, df), None) if decompose else eta
