"""``discretize`` of the parabolic block-SWIPDG LRBMS discretization
(reference python/dune/pylrbms/discretize_parabolic_block_swipdg.py:17-95; SURVEY.md section 8f "next" #3).

``M u' + A(mu) u = f`` with the block L2 product as mass (:49-59), zero initial data (:82) and pyMOR's implicit Euler
with ``nt`` steps (:87).  The reference's version does not run at HEAD; repaired call sites:

* ``discretize_ell(grid_and_problem_data)`` (:44) -> the three-argument elliptic ``discretize``;
* ``ParabolicEstimator(...)`` with 8 arguments (:76-77) -> the 12-argument constructor of estimators.py:28-30 (grid and
  ``mpi_comm`` taken from the elliptic estimator);
* the estimator's ``elliptic_reconstruction=True`` hits ``assert False`` (estimators.py:64) -> flag, default off.

``solve(mu)`` is ONE native call for the whole trajectory (``lrbms_fom_implicit_euler``): the theta-weighted block
operator plus the mass is combined once, every step is a warm-started CG on it.  On a sharded discretization the solves
run on the gathered operator, and so does the time residual of the parabolic estimate (its elliptic-reconstruction
variant needs all subdomains on one rank).
"""
import numpy as np

from pylrbms_amd.discretize_elliptic_block_swipdg import DuneDiscretization, OperatorHandle
from pylrbms_amd.discretize_elliptic_block_swipdg import discretize as discretize_ell
from pylrbms_amd.estimators import ParabolicEstimator
from pylrbms_amd.parameters import CubicParameterSpace
from pylrbms_amd.vectorarrays import BlockVectorArray


class ImplicitEulerTimeStepper:
    """Stand-in for ``pymor.algorithms.timestepping.ImplicitEulerTimeStepper``: only ``nt`` is read on this path."""

    def __init__(self, nt, solver_options='operator'):
        self.nt = int(nt)
        self.solver_options = solver_options


class InstationaryDuneDiscretization(DuneDiscretization):
    """Reference :17-40.  Built from the elliptic discretization by ``discretize`` below."""

    def _solve_options(self, inverse_options):
        opts = inverse_options if isinstance(inverse_options, dict) else {}
        return min(float(opts.get('precision', 1e-12)), 1e-10), max(int(opts.get('max_iter', 20000)), 20000)

    def solve(self, mu, inverse_options=None):
        """``_solve`` (:28-40): ``nt + 1`` vectors, the first one the (zero) initial data."""
        import torch
        eng = self.engine
        rtol, max_iter = self._solve_options(inverse_options)
        dt = self.T / self.time_stepper.nt
        U0 = self.initial_data.tensor[:, :, 0] if self.initial_data is not None else None
        if eng.S_ext != eng.S:
            # sharded: like the stationary solve, on the gathered block operator (DuneDiscretization._global_fom); the
            # initial data of the reference is zero (:82), a non-zero one would have to be gathered as well
            if U0 is not None and bool((U0 != 0).any()):
                raise NotImplementedError('non-zero initial data on a sharded discretization')
            ctx, A_diag_all, A_cpl_all, b_all = self._global_fom()
            U, info = ctx.fom_implicit_euler(self.theta(mu), dt, self.time_stepper.nt, A_diag_all, A_cpl_all, b_all, rtol=rtol,
                                             max_iter=max_iter)
            U = U[:, torch.as_tensor(eng.local, device=U.device)]
        else:
            U, info = eng.ctx.fom_implicit_euler(self.theta(mu), dt, self.time_stepper.nt, eng.A_diag, eng.A_cpl, eng.b, U0=U0,
                                                 rtol=rtol, max_iter=max_iter)
        self.last_solve_info = info
        return BlockVectorArray(U.permute(1, 2, 0), self.solution_space)

    def solve_stationary(self, mu, inverse_options=None):
        """The elliptic solve of the underlying discretization (the limit ``t -> oo``)."""
        return DuneDiscretization.solve(self, mu, inverse_options=inverse_options)

    def _time_residual_norm2(self, dU, mu):
        """``R = operator.apply(dU, mu); l2_product.apply_inverse(R).pairwise_dot(R)`` (estimators.py:146-148): [len(dU)]."""
        eng = self.engine
        if eng.S_ext != eng.S:
            # sharded: the residual norm is a sum over ALL subdomains; like the solves it is taken on the gathered block
            # operator (every rank gathers the difference vectors and evaluates the same global sum)
            from pylrbms_amd.parallel import gather_subdomain_rows
            ctx, A_d, A_c, _ = self._global_fom()
            dU_all = gather_subdomain_rows(dU.tensor.contiguous(), self._owned_subdomains(), eng.grid.num_subdomains,
                                           getattr(self.mpi_comm, 'group', None)).contiguous()
            R = ctx.fom_apply(self.theta(mu), A_d, A_c, dU_all)
            return ctx.mass_inverse_norm2(R).sum(dim=0).cpu().numpy()
        R = eng.ctx.fom_apply(self.theta(mu), eng.A_diag, eng.A_cpl, dU.tensor.contiguous())
        return eng.ctx.mass_inverse_norm2(R).sum(dim=0).cpu().numpy()

    def _reconstruction_terms(self, U, mu):
        """``r_l2(BU_R, BU_R) - r_l2(F_R, F_R) - 2 r_ud(BUF_R, U_r)`` per subdomain and vector (estimators.py:65-68,
        :80-83) -> [S, len(U)]:  BU = A(mu) U (``lrbms_fom_apply``), the ``r_l2`` terms are ``M^-1`` norms
        (``lrbms_mass_inverse_norm2``), and in ``r_ud(M^-1 (BU - f), U_r)`` the mass matrices cancel:
        ``(BU - f)^T Div U_r`` (``lrbms_flux_reconstruct`` -> ``lrbms_div_apply`` -> ``lrbms_div_pairing``)."""
        import torch
        eng = self.engine
        if eng.S_ext != eng.S:
            raise NotImplementedError('the elliptic-reconstruction terms need all subdomains on one rank')
        theta = self.theta(mu)
        c = eng.ctx
        t2 = c.mass_inverse_norm2(eng.b.reshape(eng.S, eng.t.n, 1).contiguous())            # [S, 1]
        out = []
        for c0 in range(0, len(U), 16):
            V = U.tensor[:, :, c0:c0 + 16].contiguous()
            BU = c.fom_apply(theta, eng.A_diag, eng.A_cpl, V)
            t1 = c.mass_inverse_norm2(BU)
            D = c.div_apply(c.flux_reconstruct(eng.F, V), mode=0)
            t3 = c.div_pairing(theta, D, (BU - eng.b.reshape(eng.S, eng.t.n, 1)).contiguous())
            out.append(t1 - t2 - 2.0 * t3)
        return torch.cat(out, dim=1)


def discretize(grid_and_problem_data, T, nt, solver_options=None, mpi_comm=None, device_index=None,
               elliptic_reconstruction=False):
    """Reference :43-95.  Returns ``(d, d_data)``."""
    d, d_data = discretize_ell(grid_and_problem_data, solver_options, mpi_comm, device_index=device_index)
    assert isinstance(d.parameter_space, CubicParameterSpace)              # :45
    d.__class__ = InstationaryDuneDiscretization
    d.T = float(T)
    d.time_stepper = ImplicitEulerTimeStepper(nt=nt, solver_options='operator')   # :87
    d.initial_data = d.solution_space.zeros(1, d.engine.ctx)               # :82
    d.mass = d.products['l2']                                              # :60
    for ii in d.engine.local:                                              # :65-74
        for kind in ('r_ud', 'r_l2'):
            name = '{}_{}'.format(kind, ii)
            d.operators[name] = OperatorHandle(name, 'l2' if kind == 'r_l2' else kind, ii, d)
    e = d.estimator                                                        # :76-77
    d.estimator = ParabolicEstimator(e.grid, e.min_diffusion_evs, e.subdomain_diameters, e.local_eta_rf_squared,
                                     e.lambda_coeffs, e.mu_bar, e.mu_hat, e.flux_reconstruction,
                                     e.oswald_interpolation_error, e.mpi_comm,
                                     elliptic_reconstruction=elliptic_reconstruction)
    parameter_range = grid_and_problem_data['parameter_range'] if 'parameter_range' in grid_and_problem_data else (0.1, 1.0)
    d.parameter_space = CubicParameterSpace(d.parameter_type, parameter_range[0], parameter_range[1])   # :93
    d.name = 'parabolic_block_swipdg'
    return d, d_data
