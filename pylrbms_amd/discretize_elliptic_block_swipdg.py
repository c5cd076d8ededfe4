"""``discretize`` of the block-SWIPDG LRBMS discretization
(reference python/dune/pylrbms/discretize_elliptic_block_swipdg.py:530-811).

Same entry point and return value ``(d, data)``; ``d`` answers what the reference's ``DuneDiscretization`` is asked on
the hot path: ``solve``, ``estimate``, ``operators[...]``, ``products['l2']``, ``solution_space.subspaces[i].id``,
``parse_parameter``, ``parameter_space``, ``unblock``, ``shape_functions``, ``neighborhoods``, ``estimator`` with
``flux_reconstruction`` / ``oswald_interpolation_error``.  All arithmetic is done by the HIP kernels through
``pylrbms_amd.engine.Engine``; nothing here computes on the CPU besides coefficient sampling and index bookkeeping.

"""
import numpy as np

from pylrbms_amd.engine import Engine, blockell_to_dense
from pylrbms_amd.estimators import EllipticEstimator
from pylrbms_amd.parallel import Communicator, HaloExchange, HaloPlan
from pylrbms_amd.parameters import CubicParameterSpace, parse_parameter
from pylrbms_amd.vectorarrays import BlockVectorArray, BlockVectorSpace, SubSpace


class OperatorHandle:
    """Named view of one operator of ``d.operators`` (the reference stores pyMOR operators there,
    block_swipdg.py:676,733-770).  ``matrix()`` copies the local sparse matrix to the host for inspection."""

    def __init__(self, name, kind, subdomain, discretization):
        self.name, self.kind, self.subdomain, self._d = name, kind, subdomain, discretization

    def matrix(self):
        eng, ii = self._d.engine, self._d.engine.local.index(self.subdomain)
        t = eng.t
        if self.kind == 'local_energy_dg_product':
            return blockell_to_dense(t, eng.P_diag[ii].cpu().numpy())
        if self.kind == 'l2':
            out = np.zeros((t.n, t.n))
            for e in range(t.n_T):
                out[3 * e:3 * e + 3, 3 * e:3 * e + 3] = t.area[e] / 12.0 * (1.0 + np.eye(3))
            return out
        raise NotImplementedError('matrix() of {} (a Concatenation in the reference)'.format(self.name))


class LinearImageOperator:
    """``d.estimator.oswald_interpolation_error`` / ``.flux_reconstruction``: ``apply(U)`` runs the K7 / K8 kernel and
    returns the target-major image array ``[S, rows, 5 * cols]`` (see include/lrbms_hip.h)."""

    def __init__(self, discretization, kind):
        self._d, self.kind = discretization, kind
        self.linear = True

    def apply(self, U, mu=None):
        d = self._d
        V = d._with_halo(U.tensor)
        if self.kind == 'oswald':
            return d.engine.ctx.oswald_apply(V)
        return d.engine.ctx.flux_reconstruct(d.engine.F, V)


class BlockProjectionOperator:
    """``data['local_projections'][ii]``: (solution space) -> (space of subdomain ii), picks block ``ii``
    (reference block_swipdg.py:696-697)."""

    def __init__(self, discretization, subdomain):
        self._d, self.subdomain, self.linear = discretization, subdomain, True
        self.name = 'local_projection_{}'.format(subdomain)

    def apply(self, U, mu=None):
        d, i = self._d, self._d.engine.local.index(self.subdomain)
        return BlockVectorArray(U.tensor[i:i + 1], BlockVectorSpace([d.solution_space.subspaces[i]]))


class LocalImageProjection:
    """``data['local_rt_projections'][ii]`` / ``data['local_oi_projections'][ii]`` (reference block_swipdg.py:699-718): the
    ``BlockRowOperator`` of the projections that pick, from the image of EVERY neighbour ``kk`` of ``ii`` under the flux
    reconstruction / the Oswald interpolation error, the component living on ``ii`` -- applied to a block array it is their
    sum.  The kernels K7 / K8 produce the images target-major (``d.estimator.flux_reconstruction.apply(U)`` ->
    ``[S, rows, 5 * cols]``, one column block per neighbour slot, include/lrbms_hip.h), so here ``apply`` adds the five
    slot blocks of target ``ii``: ``[rows, cols]`` on the device."""

    def __init__(self, discretization, subdomain, kind):
        self._d, self.subdomain, self.kind, self.linear = discretization, subdomain, kind, True
        self.name = 'local_{}_projection_{}'.format(kind, subdomain)

    def apply(self, images, mu=None):
        i = self._d.engine.local.index(self.subdomain)
        rows, wide = images.shape[1], images.shape[2]
        assert wide % 5 == 0, 'expected the target-major image array [S, rows, 5 * cols]'
        return images[i].view(rows, 5, wide // 5).sum(dim=1)


class LocalDivergenceOperator:
    """``data['local_div_ops'][ii]`` (reference block_swipdg.py:722-729): RT0 coefficients on subdomain ``ii`` -> DG
    coefficients of the (piecewise constant) divergence.  ``apply`` runs the native kernel (``lrbms_div_apply``, all
    subdomains of the rank in one launch, this one returned); ``matrix()`` is the dense ``[n, n_rt]`` matrix for inspection."""

    def __init__(self, discretization, subdomain):
        self._d, self.subdomain, self.linear = discretization, subdomain, True
        self.name = 'local_divergence_{}'.format(subdomain)

    def apply(self, R, mu=None):
        """R: ``[n_rt, L]`` device tensor of RT0 coefficients on this subdomain -> ``[n, L]``."""
        eng, i = self._d.engine, self._d.engine.local.index(self.subdomain)
        Rt = eng.ctx.zeros(eng.S, eng.t.n_rt, R.shape[1])
        Rt[i] = R
        per_element = eng.ctx.div_apply(Rt, mode=0)[i]               # [n_T, L]
        return per_element.repeat_interleave(3, dim=0)

    def matrix(self):
        eng = self._d.engine
        t, nbr = eng.t, eng.nbr[eng.local.index(self.subdomain)]
        out = np.zeros((t.n, t.n_rt))
        for e in range(t.n_T):
            for f in range(3):
                sign, nb = t.face_sign[e, f], t.nb_elem[e, f]
                if nb < 0:                                   # face on side -1 - nb: outward where that side is domain boundary
                    side = -1 - nb
                    if nbr[side if side < 2 else side + 1] < 0:
                        sign = 1
                out[3 * e:3 * e + 3, t.elem_rt[e, f]] = sign * t.face_len[e, f] / t.area[e]
        return out


class DuneDiscretization:
    """Block-SWIPDG discretization living on one GPU (one rank's tile of subdomains)."""

    def __init__(self, engine, grid_and_problem_data, solver_options, mpi_comm):
        self.engine = engine
        self.grid = grid_and_problem_data['grid']
        self.data = None
        self.solver_options = solver_options
        self.mpi_comm = mpi_comm if mpi_comm is not None else Communicator()
        p = grid_and_problem_data
        lam = p['lambda']
        self.lambda_coeffs = lam['coefficients'] if isinstance(lam, dict) else None
        self.parameter_type = p.get('parameter_type', {})
        self.neighborhoods = [self.grid.neighborhood_of(ii) for ii in range(self.grid.num_subdomains)]
        self.enrichment_data = (self.grid, None, p['lambda'], p['kappa'], p['f'], None)
        n = engine.t.n
        self.solution_space = BlockVectorSpace([SubSpace(n, 'domain_{}'.format(ii)) for ii in engine.local])
        self.name = 'block_swipdg'
        self._halo = {}
        self.parameter_space = None

    # ------------------------------------------------------------------ plumbing
    def _with_halo(self, t_local):
        """[S, n, L] -> [S_ext, n, L] with the halo slabs filled (one all-gather when sharded)."""
        eng = self.engine
        if eng.S_ext == eng.S:
            return t_local.contiguous()
        import torch
        L = t_local.shape[2]
        V = torch.zeros(eng.S_ext, eng.t.n, L, dtype=t_local.dtype, device=t_local.device)
        V[:eng.S] = t_local
        if L not in self._halo:
            g = self.grid
            from pylrbms_amd.grid import DDSubdomainsGrid
            plan = HaloPlan(lambda r: DDSubdomainsGrid(g.lower_left, g.upper_right, g.K, g.P, rank=r,
                                                       world_size=g.world_size), g.world_size, g.rank,
                            diagonal=bool(eng.conventions.get('oswald_vertex_patch')))
            self._halo[L] = HaloExchange(plan, L, V.device, group=getattr(self.mpi_comm, 'group', None))
        return self._halo[L](V)

    def parse_parameter(self, mu):
        return parse_parameter(mu, self.parameter_type)

    def theta(self, mu):
        mu = self.parse_parameter(mu)
        return np.array([c.evaluate(mu) for c in self.lambda_coeffs])

    def with_(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)
        return self

    # ------------------------------------------------------------------ reference API
    def unblock(self, U):
        return U.data                                                     # block-mapper ordering == global ordering

    def visualize(self, U, filename='solution', name='u', **kwargs):
        """``DuneGDTVisualizer`` (block_swipdg.py:802): writes this rank's subdomains as legacy VTK files."""
        from pylrbms_amd.visualize import visualize_block_array
        return visualize_block_array(U, self.grid, self.engine.local, filename, name=name)

    def shape_functions(self, subdomain, order=0):
        """block_swipdg.py:187-200: only ``order=0`` (the constant) works in the reference (App. B-4)."""
        assert 0 <= order <= 1
        if order == 1:
            raise NotImplementedError('order=1 calls the undefined dune_project in the reference (block_swipdg.py:197)')
        space = BlockVectorSpace([self.solution_space.subspaces[self.engine.local.index(subdomain)]])
        return BlockVectorArray(self.engine.ctx.zeros(1, self.engine.t.n, 1) + 1.0, space)

    def solve(self, mu, inverse_options=None):
        """``DuneDiscretization._solve`` (block_swipdg.py:219-225).  The reference hands the global matrix to ISTL
        (bicgstab.ilut); here: preconditioned CG on the SPD block operator, never assembled, by the native
        ``lrbms_fom_solve`` (element-block Jacobi + coarse level on the subdomain indicator functions, two to four launches
        per iteration, no host round trips).  Sharded: every rank gathers the block operator once and solves redundantly.
        (Snapshot generation is the step before the hot path: SURVEY.md section 8f #2.)"""
        import torch
        eng = self.engine
        theta = self.theta(mu)
        opts = inverse_options or {}
        rtol = float(opts.get('precision', 1e-12)) if isinstance(opts, dict) else 1e-12
        rtol = min(rtol, 1e-10)
        max_iter = int(opts.get('max_iter', 20000)) if isinstance(opts, dict) else 20000
        max_iter = max(max_iter, 20000)
        sharded = eng.S_ext != eng.S
        if not sharded:
            x, info = eng.ctx.fom_solve(theta, eng.A_diag, eng.A_cpl, eng.b, rtol=rtol, max_iter=max_iter)
            self.last_solve_info = info
            return BlockVectorArray(x.reshape(eng.S, eng.t.n, 1), self.solution_space)

        # sharded: the block operator of the whole domain is small next to 288 GB of HBM (75 MB of diagonal blocks at config
        # 3), so every rank gathers it once (one all-gather each for A_diag, A_cpl, b) and solves redundantly with the native
        # solver through a second library context that holds the GLOBAL neighbour table; it keeps its own rows.
        ctx, A_diag_all, A_cpl_all, b_all = self._global_fom()
        x, info = ctx.fom_solve(theta, A_diag_all, A_cpl_all, b_all, rtol=rtol, max_iter=max_iter)
        self.last_solve_info = info
        x = x[torch.as_tensor(eng.local, device=x.device)]
        return BlockVectorArray(x.reshape(eng.S, eng.t.n, 1), self.solution_space)

    def _owned_subdomains(self):
        """One list of global subdomain indices per rank (every rank computes the same lists)."""
        if getattr(self, '_owned', None) is None:
            from pylrbms_amd.grid import DDSubdomainsGrid
            g = self.engine.grid
            self._owned = [list(DDSubdomainsGrid(g.lower_left, g.upper_right, g.K, g.P, rank=r, world_size=g.world_size).subdomains_on_rank)
                           for r in range(g.world_size)]
        return self._owned

    def _global_fom(self):
        """(context, A_diag [Q, S_total, n_T, 4, 9], A_cpl [Q, S_total, 4, ncf, 9], b [S_total, n]) in global order."""
        if getattr(self, '_fom_global', None) is None:
            from pylrbms_amd._native import NativeContext
            from pylrbms_amd.grid import DDSubdomainsGrid
            from pylrbms_amd.parallel import gather_subdomain_rows
            eng = self.engine
            g = eng.grid
            group = getattr(self.mpi_comm, 'group', None)
            owned = [list(DDSubdomainsGrid(g.lower_left, g.upper_right, g.K, g.P, rank=r, world_size=g.world_size).subdomains_on_rank)
                     for r in range(g.world_size)]
            total = g.num_subdomains
            A_d = gather_subdomain_rows(eng.A_diag.permute(1, 0, 2, 3, 4).contiguous(), owned, total, group)
            A_c = gather_subdomain_rows(eng.A_cpl.permute(1, 0, 2, 3, 4).contiguous(), owned, total, group)
            b = gather_subdomain_rows(eng.b, owned, total, group)
            ctx = NativeContext(eng.ctx.device.index)
            nbr = np.asarray(g.neighbor_slots, dtype=np.int32).reshape(total, 5)
            ctx.mesh_upload(eng.t, eng.kappa, nbr, total, total)
            self._fom_global = (ctx, A_d.permute(1, 0, 2, 3, 4).contiguous(), A_c.permute(1, 0, 2, 3, 4).contiguous(), b.contiguous())
        return self._fom_global

    def _local_estimates(self, U, mu):
        """Per-subdomain nc / r / df for every vector of a full-order array: the vectors become a basis of ``len(U)``
        columns (in chunks of 16) pushed through K7 / K8 / P2, and the k-th unit coefficient vector selects the k-th
        pairwise form -- one batched estimate launch per chunk."""
        import torch
        eng = self.engine
        theta = self.theta(mu)
        Vall = self._with_halo(U.tensor)
        out = []
        for c0 in range(0, Vall.shape[2], 16):
            V = Vall[:, :, c0:c0 + 16].contiguous()
            L = V.shape[2]
            buf = eng.project_and_estimate(V, project_system=False)
            if L == 1:
                u = eng.ctx.zeros(eng.S_ext, 1) + 1.0
                out.append(eng.reduced_estimate(theta, u, buf['grams'])[:, :, None])
                continue
            u = torch.eye(L, dtype=V.dtype, device=V.device).expand(eng.S_ext, L, L).contiguous()
            out.append(eng.ctx.reduced_estimate_batch(np.tile(theta, (L, 1)), u, buf['grams'], eng.f2, eng.ceps, eng.hdiam))
        eta = torch.cat(out, dim=2)                                        # [3, S, len(U)]
        return eta[0], eta[1], eta[2]

    def estimate(self, U, mu=None, decompose=False):
        return self.estimator.estimate(U, self.parse_parameter(mu), self, decompose=decompose)

    def solve_for_local_correction(self, subdomain, Us, mu=None, inverse_options=None):
        """block_swipdg.py:227-316.  ``Us`` (the current solution on the neighbourhood) is accepted and unused, as in
        the reference, whose Dirichlet-lift functional is commented out (:250-261): the corrector is the solution of the
        neighbourhood problem with homogeneous Dirichlet values on the outer boundary and right-hand side f."""
        return self.solve_for_local_corrections([subdomain], mu, inverse_options=inverse_options)[0]

    def solve_for_local_corrections(self, subdomains, mu=None, inverse_options=None):
        """All corrector problems of one enrichment round in one launch (``lrbms_local_correction_solve``); returns
        one single-vector array per subdomain, on the subdomain's local space."""
        eng = self.engine
        opts = inverse_options if isinstance(inverse_options, dict) else {}
        rtol = min(float(opts.get('precision', 1e-12)), 1e-10)
        marked = [eng.local.index(int(ii)) for ii in subdomains]
        corr, info = eng.local_corrections(self.theta(self.parse_parameter(mu)), marked, rtol=rtol)
        self.last_local_correction_info = info
        out = []
        for k, i in enumerate(marked):
            space = BlockVectorSpace([self.solution_space.subspaces[i]])
            out.append(BlockVectorArray(corr[k].reshape(1, eng.t.n, 1), space))
        return out


def discretize(grid_and_problem_data, solver_options=None, mpi_comm=None, device_index=None, conventions=None, quadrature=None):
    """Reference block_swipdg.py:530-811.  Returns ``(d, data)`` with ``data`` keys as at :631-637."""
    p = grid_and_problem_data
    grid = p['grid']
    lambda_, kappa = p['lambda'], p['kappa']
    if isinstance(lambda_, dict):
        lambda_funcs, lambda_coeffs = lambda_['functions'], lambda_['coefficients']
    else:
        from pylrbms_amd.parameters import ConstantParameterFunctional
        lambda_funcs, lambda_coeffs = [lambda_], [ConstantParameterFunctional(1.)]
        p = dict(p, **{'lambda': {'functions': lambda_funcs, 'coefficients': lambda_coeffs}})
    f = p['f']
    if isinstance(f, dict):                                               # block_swipdg.py:589-595,:739-748,:780-785
        if len(f['functions']) != 1 or f['coefficients'][0] != 1:
            raise NotImplementedError('the residual operators exist only for one f component with coefficient 1')
        f = f['functions'][0]
    mu_bar, mu_hat = p['mu_bar'], p['mu_hat']
    theta_bar = np.array([c.evaluate(mu_bar) for c in lambda_coeffs])
    if device_index is None:
        import torch
        device_index = torch.cuda.current_device() if torch.cuda.is_available() else 0
    # ``conventions``: switches for what the reference tree leaves open (include/lrbms_hip.h
    # LRBMS_OPT_*), e.g. {'oswald_zero_on_subdomain_boundary': True, 'accumulate_coupling_across_q': True}
    engine = Engine(grid, lambda_funcs, kappa, f, p['lambda_bar'], p['lambda_hat'], theta_bar, device_index=device_index,
                    conventions=conventions, quadrature=quadrature)     # quadrature: a QuadratureSpec (default: the reference's orders)
    engine.assemble()

    d = DuneDiscretization(engine, p, solver_options, mpi_comm)
    operators = {}
    for ii in engine.local:
        for kind in ('nc', 'r_fd', 'r_dd', 'df_aa', 'df_bb', 'df_ab', 'local_energy_dg_product'):
            name = '{}_{}'.format(kind, ii)
            operators[name] = OperatorHandle(name, kind, ii, d)
    d.operators = operators
    d.products = {'l2': [OperatorHandle('l2_{}'.format(ii), 'l2', ii, d) for ii in engine.local]}
    oi_op, fr_op = LinearImageOperator(d, 'oswald'), LinearImageOperator(d, 'flux')
    ones = np.ones(engine.S)
    d.estimator = EllipticEstimator(grid, engine.ceps, engine.hdiam * ones, engine.f2, lambda_coeffs, mu_bar, mu_hat,
                                    fr_op, oswald_interpolation_error=oi_op, mpi_comm=d.mpi_comm)
    parameter_range = p['parameter_range'] if 'parameter_range' in p else (0.1, 1.0)
    d.parameter_space = CubicParameterSpace(d.parameter_type, parameter_range[0], parameter_range[1])

    class _BlockSpace:                                                    # data['block_space'] (block_swipdg.py:632)
        num_blocks = grid.num_subdomains

        class mapper:
            size = grid.num_subdomains * engine.t.n

        @staticmethod
        def local_space(ii):
            class _Local:
                @staticmethod
                def size():
                    return engine.t.n
            return _Local

    owned = list(engine.local)
    data = dict(grid=grid, block_space=_BlockSpace,
                local_projections=[BlockProjectionOperator(d, ii) for ii in owned],
                local_rt_projections=[LocalImageProjection(d, ii, 'rt') for ii in owned],
                local_oi_projections=[LocalImageProjection(d, ii, 'oi') for ii in owned],
                local_div_ops=[LocalDivergenceOperator(d, ii) for ii in owned], local_l2_products=d.products['l2'])
    d.data = data
    return d, data
