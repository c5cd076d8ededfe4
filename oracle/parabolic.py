"""CPU oracle for the parabolic LRBMS path -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED.

Restates, on top of ``oracle.lrbms``,

* ``discretize``                     python/dune/pylrbms/discretize_parabolic_block_swipdg.py:43-95
  (mass = block L2 product :49-59, zero initial data :82, ``ImplicitEulerTimeStepper(nt)`` :87)
* ``InstationaryDuneDiscretization._solve``   ...parabolic_block_swipdg.py:28-40
* ``ParabolicEstimator.estimate``    python/dune/pylrbms/estimators.py:139-168
* the elliptic-reconstruction branch of ``_estimate_elliptic``  estimators.py:63-68, :80-83
* operators ``r_ud_i`` / ``r_l2_i``  ...parabolic_block_swipdg.py:65-74

The reference's parabolic path does not run at HEAD: ``discretize_ell`` is called with one
argument (:44) but takes three, ``ParabolicEstimator(...)`` is built with 8 arguments (:76-77) for a 12-argument
constructor (estimators.py:28-30), and the estimator passes ``elliptic_reconstruction=True`` (estimators.py:143) into a
branch that starts with ``assert False`` (estimators.py:64).  What is restated here is the code as written with those
three call sites repaired; ``elliptic_reconstruction`` selects whether the terms behind the ``assert False`` are
evaluated (True) or skipped (False).

pyMOR's implicit Euler (pymor/algorithms/timestepping.py, not in the tree; published scheme):
``(M + dt A(mu)) U_{k+1} = M U_k + dt F``, ``U_0`` = initial data, ``nt`` steps, ``nt + 1`` vectors returned.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from .lrbms import OracleReductor


def _eta_per_vector(nc, r, df, a_bar, g_bar, a_hat):
    """estimators.py:99-102 for ``len(U)`` vectors.  ``mpi_norm`` lives in the absent pyMOR fork; it is read here as the
    norm over the subdomains for every vector (one eta per time step): ``np.linalg.norm(eta)`` at estimators.py:166 and
    the in-place scaling at :153 are written for an array, like ``time_residual`` beside it."""
    eta = np.sqrt(g_bar) * np.linalg.norm(nc, axis=0) + (1.0 / np.sqrt(a_hat)) * np.linalg.norm(r + df, axis=0)
    return eta * (1.0 / np.sqrt(a_bar))


class OracleParabolic:
    """``InstationaryDuneDiscretization`` on top of an ``OracleDiscretization`` ``d``."""

    def __init__(self, d, T, nt):
        self.d, self.T, self.nt = d, float(T), int(nt)
        self.dt = self.T / self.nt

    def solve(self, mu):
        """[nt + 1, S, n]; zero initial data (parabolic_block_swipdg.py:82)."""
        d = self.d
        A = d.assemble_global(mu)
        M = d.l2_product.tocsc()
        lu = spla.splu((M + self.dt * A).tocsc())
        U = np.zeros((self.nt + 1, d.ndof))
        for k in range(self.nt):
            U[k + 1] = lu.solve(M @ U[k] + self.dt * d.b)
        return U.reshape(self.nt + 1, d.S, d.n)

    def _elliptic_local(self, U, mu, elliptic_reconstruction):
        """local_eta_nc / r / df [S, len(U)] of _estimate_elliptic (estimators.py:45-91), squared as written."""
        d, m = self.d, self.d.mesh
        L = U.shape[0]
        nc, r, df = np.zeros((d.S, L)), np.zeros((d.S, L)), np.zeros((d.S, L))
        n = d.n
        A = d.assemble_global(mu)
        Minv = spla.splu(d.l2_product.tocsc())
        th = d.theta(mu)
        for k in range(L):
            bases = [U[k, ii][:, None] for ii in range(d.S)]
            red = OracleReductor(d, bases)
            rd = red.reduce(project_system=False)
            _, (e_nc, e_r, e_df), _ = rd.estimate(np.ones((d.S, 1)), mu, decompose=True)
            nc[:, k], r[:, k], df[:, k] = e_nc, e_r, e_df
            if elliptic_reconstruction:
                OI, RT = red.image_bases()
                BU = A @ U[k].reshape(-1)                                   # estimators.py:65
                BU_R = Minv.solve(BU)                                       # :66
                F_R = Minv.solve(d.b)                                       # :67
                BUF_R = BU_R - F_R                                          # :68
                for ii in range(d.S):
                    hood = m.neighborhood_of(ii)
                    M_ii = d.block(d.l2_product, ii, ii)
                    sl = slice(ii * n, (ii + 1) * n)
                    # U_r restricted to the RT space of ii: sum over the sources kk in N(ii) and the affine
                    # components (FluxReconstructionOperator is the LincombOperator sum_q theta_q FR_q)
                    ur = np.zeros(d.n_rt[ii])
                    for kk in hood:
                        blk = RT[kk][m.neighborhood_of(kk).index(ii)]       # [n_rt, Q] (one column per q)
                        ur += blk @ th
                    add = BU_R[sl] @ (M_ii @ BU_R[sl])                      # + r_l2(BU_R, BU_R)   :81
                    add -= F_R[sl] @ (M_ii @ F_R[sl])                       # - r_l2(F_R, F_R)     :82
                    add -= 2.0 * (BUF_R[sl] @ (M_ii @ (d.Div[ii] @ ur)))    # - 2 r_ud(BUF_R, U_r) :83
                    r[ii, k] += add * (1.0 / np.pi ** 2) / d.min_diffusion_evs[ii] * d.subdomain_diameters[ii] ** 2
        return nc, r, df

    def estimate(self, U, mu, elliptic_reconstruction=False):
        """ParabolicEstimator.estimate (estimators.py:141-168): returns
        ``est, (local_eta_nc, local_eta_r, local_eta_df, time_residual, time_deriv_nc)``."""
        d = self.d
        dt = self.dt
        nc, r, df = self._elliptic_local(U, mu, elliptic_reconstruction)
        a_bar, g_bar, a_hat = d.alpha(mu, d.mu_bar), d.gamma(mu, d.mu_bar), d.alpha(mu, d.mu_hat)
        eta = _eta_per_vector(nc, r, df, a_bar, g_bar, a_hat)                                         # :99-102
        A = d.assemble_global(mu)
        Minv = spla.splu(d.l2_product.tocsc())
        dU = (U[1:] - U[:-1]).reshape(U.shape[0] - 1, -1)
        time_residual = np.zeros(dU.shape[0])
        for k in range(dU.shape[0]):
            y = A @ dU[k]                                                   # :147
            time_residual[k] = Minv.solve(y) @ y                            # :148
        time_residual = np.sqrt(time_residual * dt / 3)                     # :149-150
        s = 2 * np.sqrt(dt / 3)                                             # :153-156
        eta, nc, r, df = eta * s, nc * s, r * s, df * s
        time_deriv_nc = np.zeros((d.S, dU.shape[0]))                        # :158-164
        for k in range(dU.shape[0]):
            bases = [dU[k].reshape(d.S, d.n)[ii][:, None] for ii in range(d.S)]
            rd = OracleReductor(d, bases).reduce(project_system=False)
            one = np.ones(1)
            for ii in range(d.S):
                uo = np.concatenate([one for _ in d.mesh.neighborhood_of(ii)])
                time_deriv_nc[ii, k] = uo @ rd.nc[ii] @ uo
        time_deriv_nc = np.sqrt(time_deriv_nc / dt)
        est = np.linalg.norm(eta) + np.linalg.norm(time_residual) + np.linalg.norm(time_deriv_nc)   # :166
        return est, (nc, r, df, time_residual, time_deriv_nc)


class OracleParabolicReduced:
    """The reduced instationary model: the same estimator code with the projected operators (``rd``: an
    ``OracleReducedModel`` built with ``project_system=True`` on equal basis sizes ``N``)."""

    def __init__(self, reductor, rd, T, nt):
        self.reductor, self.rd, self.d = reductor, rd, rd.d
        self.T, self.nt = float(T), int(nt)
        self.dt = self.T / self.nt
        self._ud = None

    def _mass(self):
        off = np.concatenate(([0], np.cumsum(self.rd.sizes)))
        M = np.zeros((off[-1], off[-1]))
        for ii in range(self.d.S):
            M[off[ii]:off[ii + 1], off[ii]:off[ii + 1]] = self.rd.l2[ii]
        return M, off

    def solve(self, mu):
        """Reduced implicit Euler; [nt + 1, sum N]."""
        A, b, off = self.rd.assemble(mu)
        M, _ = self._mass()
        lhs = M + self.dt * A
        u = np.zeros((self.nt + 1, off[-1]))
        for k in range(self.nt):
            u[k + 1] = np.linalg.solve(lhs, M @ u[k] + self.dt * b)
        return u

    def _split(self, u, off):
        return [u[off[ii]:off[ii + 1]] for ii in range(self.d.S)]

    def _projected_r_ud(self):
        """V_ii^T M_ii Div_ii Rt_ii (projection of r_ud_ii, parabolic_block_swipdg.py:69-70), compact column order."""
        if self._ud is None:
            d, m = self.d, self.d.mesh
            OI, RT = self.reductor.image_bases()
            self._ud = []
            for ii in range(d.S):
                hood = m.neighborhood_of(ii)
                Rt = np.hstack([RT[kk][m.neighborhood_of(kk).index(ii)] for kk in hood])
                M_ii = d.block(d.l2_product, ii, ii)
                self._ud.append(self.reductor.bases[ii].T @ (M_ii @ (d.Div[ii] @ Rt)))
        return self._ud

    def estimate(self, u, mu, elliptic_reconstruction=False):
        d, m, rd = self.d, self.d.mesh, self.rd
        dt = self.dt
        L = u.shape[0]
        A, b, off = rd.assemble(mu)
        M, _ = self._mass()
        th = d.theta(mu)
        nc, r, df = np.zeros((d.S, L)), np.zeros((d.S, L)), np.zeros((d.S, L))
        for k in range(L):
            uk = self._split(u[k], off)
            _, (e_nc, e_r, e_df), _ = rd.estimate(uk, mu, decompose=True)
            nc[:, k], r[:, k], df[:, k] = e_nc, e_r, e_df
            if elliptic_reconstruction:
                ud = self._projected_r_ud()
                BU_R = np.linalg.solve(M, A @ u[k])
                F_R = np.linalg.solve(M, b)
                BUF_R = BU_R - F_R
                for ii in range(d.S):
                    hood = m.neighborhood_of(ii)
                    sl = slice(off[ii], off[ii + 1])
                    ur = np.concatenate([np.concatenate([th[q] * uk[kk] for q in range(d.Q)]) for kk in hood])
                    add = BU_R[sl] @ rd.l2[ii] @ BU_R[sl] - F_R[sl] @ rd.l2[ii] @ F_R[sl] - 2.0 * (BUF_R[sl] @ ud[ii] @ ur)
                    r[ii, k] += add * (1.0 / np.pi ** 2) / d.min_diffusion_evs[ii] * d.subdomain_diameters[ii] ** 2
        a_bar, g_bar, a_hat = d.alpha(mu, d.mu_bar), d.gamma(mu, d.mu_bar), d.alpha(mu, d.mu_hat)
        eta = _eta_per_vector(nc, r, df, a_bar, g_bar, a_hat)
        du = u[1:] - u[:-1]
        time_residual = np.zeros(L - 1)
        for k in range(L - 1):
            y = A @ du[k]
            time_residual[k] = np.linalg.solve(M, y) @ y
        time_residual = np.sqrt(time_residual * dt / 3)
        s = 2 * np.sqrt(dt / 3)
        eta, nc, r, df = eta * s, nc * s, r * s, df * s
        time_deriv_nc = np.zeros((d.S, L - 1))
        for k in range(L - 1):
            duk = self._split(du[k], off)
            for ii in range(d.S):
                uo = np.concatenate([duk[kk] for kk in m.neighborhood_of(ii)])
                time_deriv_nc[ii, k] = uo @ rd.nc[ii] @ uo
        time_deriv_nc = np.sqrt(time_deriv_nc / dt)
        est = np.linalg.norm(eta) + np.linalg.norm(time_residual) + np.linalg.norm(time_deriv_nc)
        return est, (nc, r, df, time_residual, time_deriv_nc)
