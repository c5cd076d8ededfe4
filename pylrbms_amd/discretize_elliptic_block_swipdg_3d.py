"""Block SWIPDG P2 discretization in 3D on the HIP path (BASELINE.json config 5) behind the reference's API shape.

The reference's ``discretize`` (python/dune/pylrbms/discretize_elliptic_block_swipdg.py:530-811) binds the 2D / P1 operators
only (:22-23); this module gives the 3D / P2 path the same surface for the calls on the hot path:

    d, data = discretize(grid_and_problem_data)          # :530       offline assembly (lrbms3_assemble_*)
    d.estimate(U, mu)                                     # :205-217   full-order estimate of a block DG vector
    reductor = LRBMSReductor3D(d, bases)                  # reductor.py:17-31
    rd = reductor.reduce()                                # reductor.py:33-73   one lrbms3_project_estimate pass
    u = rd.solve(mu);  rd.estimate(u, mu)                 # online (estimators.py:45-130)

``grid_and_problem_data``: the dict of ``pylrbms_amd.multiscale_problem3d.init_grid_and_problem`` (grid, lambda functions and
coefficient functionals, lambda_bar / lambda_hat, f, mu_bar / mu_hat).  ``d.solve(mu)`` generates snapshots (block-Jacobi CG)."""
import numpy as np

from pylrbms_amd._native import NativeError
from pylrbms_amd.engine3d import Engine3D


class BlockDiscretization3D:
    def __init__(self, p, device_index=0):
        self.grid = p['grid']
        lam = p['lambda']
        self.coefficients = list(lam['coefficients'])
        self.mu_bar, self.mu_hat = p['mu_bar'], p['mu_hat']
        self.engine = Engine3D(self.grid, lam['functions'], p['f'], p['lambda_bar'], p['lambda_hat'],
                               data_degree=p.get('data_degree', 2), device_index=device_index,
                               theta_bar=[float(c(self.mu_bar)) for c in self.coefficients]).assemble()
        self.Q = self.engine.Q
        self.parameter_range = p.get('parameter_range')

    def theta(self, mu):
        return np.array([float(c(mu)) for c in self.coefficients], dtype=np.float64)

    def shape_functions(self, subdomain, order=0):
        """``d.shape_functions(subdomain, order)`` (discretize_elliptic_block_swipdg.py:190-203 in 2D): the constant (order 0)
        and, for order 1, the three coordinate functions relative to the subdomain centre, as P2 nodal vectors [n, 1 or 4]
        (device).  The reductor starts every local basis with them (reductor.py:29-31)."""
        if order not in (0, 1):
            raise NotImplementedError('shape functions of order 0 and 1')
        eng = self.engine
        x = np.asarray(eng.t.node_coordinates(), dtype=np.float64)
        cols = [np.ones(eng.t.n)]
        if order == 1:
            centre = 0.5 * (x.max(axis=0) + x.min(axis=0))
            cols += [x[:, a] - centre[a] for a in range(3)]
        return eng.ctx.from_numpy(np.stack(cols, axis=1))

    def alpha(self, mu, mu2):
        """min_q theta_q(mu) / theta_q(mu2) as written in the reference: the loop returns in its first pass
        (estimators.py:114-121), i.e. the first component only."""
        return self.coefficients[0](mu) / self.coefficients[0](mu2)

    def gamma(self, mu, mu2):
        return max(c(mu) / c(mu2) for c in self.coefficients)

    def combine(self, eta_loc, mu, decompose=False):
        """EstimatorBase._estimate_elliptic (estimators.py:99-112) from the local terms [3, S]."""
        nc, r, df = (np.asarray(x, dtype=np.float64) for x in eta_loc)
        a_bar, a_hat, g_bar = self.alpha(mu, self.mu_bar), self.alpha(mu, self.mu_hat), self.gamma(mu, self.mu_bar)
        if getattr(self.grid, 'world_size', 1) > 1:
            # sharded: eta_loc holds this rank's subdomains only; the two mpi_norm of estimators.py:100-101 as one all-reduce of the
            # sums of squares (parallel.global_norms, as in 2D) -- on the device, RCCL has no host collectives
            import torch
            from pylrbms_amd.parallel import global_norms
            dev = self.engine.ctx.device
            norms = global_norms(torch.as_tensor(nc, device=dev), torch.as_tensor(r + df, device=dev), getattr(self, 'group', None))
            n_nc, n_rdf = float(norms[0]), float(norms[1])
        else:
            n_nc, n_rdf = np.linalg.norm(nc), np.linalg.norm(r + df)
        eta = (1.0 / np.sqrt(a_bar)) * (np.sqrt(g_bar) * n_nc + (1.0 / np.sqrt(a_hat)) * n_rdf)
        if not decompose:
            return eta
        return eta, (nc, r, df), (2.0 / a_bar) * (g_bar * nc ** 2 + (1.0 / a_hat) * (r + df) ** 2)

    def solve(self, mu, rtol=1e-10, max_iter=50000, return_info=False):
        """``d.solve(mu)`` (:219-225): the full-order solution as a block DG vector [S, n] -- CG on the never-assembled block
        operator with a two-level preconditioner (10 x 10 element blocks + P1 per subdomain; ``lrbms3_fom_solve``); snapshot
        generation.  Sharded (one tile of subdomains per rank): as in 2D every rank gathers the block operator once (1.6 GB at
        config 5, against 288 GB of HBM) and solves redundantly through a second context that holds the GLOBAL neighbour table;
        it keeps its own rows."""
        eng = self.engine
        if eng.S_ext == eng.S:
            if not getattr(self, '_fom_kept', False):       # one dense coarse factorisation for all snapshots of this discretization
                eng.ctx.fom_precond_keep(True)
                self._fom_kept = True
            U, info = eng.ctx.fom_solve(self.Q, self.theta(mu), eng.ops['A_diag'], eng.ops['A_cpl'], eng.ops['b'], rtol=rtol,
                                        max_iter=max_iter)
            return (U, info) if return_info else U
        import torch
        ctx, A_d, A_c, b = self._global_fom()
        U, info = ctx.fom_solve(self.Q, self.theta(mu), A_d, A_c, b, rtol=rtol, max_iter=max_iter)
        U = U[torch.as_tensor(eng.local, device=U.device)].contiguous()
        return (U, info) if return_info else U

    def _global_fom(self):
        """(context, A_diag [Q, S_total, n_T, 5, 100], A_cpl [Q, S_total, 6, ncf, 100], b [S_total, n]) in global subdomain order."""
        if getattr(self, '_fom_global', None) is None:
            from pylrbms_amd._native3d import Native3DContext
            from pylrbms_amd.parallel import gather_subdomain_rows
            eng, g = self.engine, self.grid
            owned = [list(g._partition(r, g.world_size)) for r in range(g.world_size)]
            total = g.num_subdomains
            group = getattr(self, 'group', None)
            A_d = gather_subdomain_rows(eng.ops['A_diag'].permute(1, 0, 2, 3, 4).contiguous(), owned, total, group)
            A_c = gather_subdomain_rows(eng.ops['A_cpl'].permute(1, 0, 2, 3, 4).contiguous(), owned, total, group)
            b = gather_subdomain_rows(eng.ops['b'], owned, total, group)
            ctx = Native3DContext(eng.ctx.device.index)
            nbr = np.asarray(g.neighbor_slots, dtype=np.int32).reshape(total, 7)
            ctx.mesh_upload(eng.t, eng.spec, eng.t.tables(eng.spec), nbr, g.phys_mask, total, total)
            ctx.fom_precond_keep(True)
            self._fom_global = (ctx, A_d.permute(1, 0, 2, 3, 4).contiguous(), A_c.permute(1, 0, 2, 3, 4).contiguous(), b.contiguous())
        return self._fom_global

    def apply(self, U, mu):
        """A(mu) U for a block DG array U [S, n, M] (BlockOperator.apply, :500-507)."""
        eng = self.engine
        return eng.ctx.fom_apply(self.Q, self.theta(mu), eng.ops['A_diag'], eng.ops['A_cpl'], U)

    def estimate(self, U, mu, decompose=False):
        """Full-order estimate of the block DG vector U [S, n] (:205-217): the pass with U as a one-column basis, u = 1."""
        eng = self.engine
        V = (U if isinstance(U, eng.ctx.torch.Tensor) else eng.ctx.from_numpy(np.asarray(U))).reshape(eng.S_ext, eng.t.n, 1).contiguous()
        out = eng.project_and_estimate(V)
        ones = eng.ctx.zeros(eng.S_ext, 1) + 1.0
        return self.combine(eng.reduced_estimate(self.theta(mu), ones, out).cpu().numpy(), mu, decompose)


class ReducedDiscretization3D:
    """``rd``: the 7-slot block-sparse reduced system and the projected estimator operators (factored layout), in HBM."""

    def __init__(self, reductor, out):
        self.reductor, self.d, self.out = reductor, reductor.d, out
        self.N = out['rhs_red'].shape[1]

    @property
    def operators(self):
        """Dense blocks of the projected estimator operators (reference: ``rd.operators``), built on request from the factors."""
        from pylrbms_amd.engine3d import expand_factored
        return expand_factored(self.d.engine, self.out, self.d.Q, self.N)

    def solve(self, mu, rtol=1e-12, max_iter=20000, return_info=False):
        """``rd.solve(mu)`` (online_adaptive_lrbms.py:141).  N <= 32: the batched solver with one parameter -- it has the
        two-level preconditioner of this reduced model and the matrix-core panel matvec; larger N: the single-parameter
        block-Jacobi PCG."""
        if self.N <= 32:
            U, info = self.solve_batch([mu], rtol=rtol, max_iter=max_iter, return_info=True)
            return (U[0], info) if return_info else U[0]
        u, info = self.d.engine.reduced_solve(self.d.theta(mu), self.out, rtol=rtol, max_iter=max_iter)
        return (u, info) if return_info else u

    def solve_batch(self, mus, rtol=1e-12, max_iter=20000, return_info=False):
        """Reduced solutions for a list of parameters, <= 64 per native call: [len(mus), S, N]."""
        import torch
        eng, out = self.d.engine, []
        info = (0, 0.0)
        if self.N > 32:                       # the batched kernels take N <= 32: one native solve per parameter
            for mu in mus:
                u, inf = self.solve(mu, rtol=rtol, max_iter=max_iter, return_info=True)
                out.append(u[None])
                info = (max(info[0], inf[0]), max(info[1], inf[1]))
            U = torch.cat(out, dim=0).contiguous()
            return (U, info) if return_info else U
        # two-level preconditioner: inverse diagonal blocks + coarse level on the first local basis vectors, the coarse inverse built
        # once per reduced model at the middle of the parameter range (mu_bar without one); any SPD preconditioner is admissible
        if getattr(self, '_pc', None) is None:
            pr = self.d.parameter_range
            mu_ref = 0.5 * (pr[0] + pr[1]) if pr is not None else self.d.mu_bar
            try:
                self._pc = eng.ctx.reduced_precond_build(self.d.Q, self.d.theta(mu_ref), self.out['B_sys'])
            except NativeError as exc:         # first basis vectors that do not give an SPD coarse matrix (e.g. a zero vector):
                if 'not positive definite' not in str(exc):      # block-Jacobi alone; anything else (HIP errors, bad arguments) is raised
                    raise
                self._pc = False
        eng.ctx.reduced_precond_use(self._pc if self._pc is not False else None)
        try:
            for b0 in range(0, len(mus), 64):          # 64 per native call: four groups of 16 on four streams
                th = np.stack([self.d.theta(mu) for mu in mus[b0:b0 + 64]])
                ub, inf = eng.ctx.reduced_solve_batch(self.d.Q, th, self.out['B_sys'], self.out['rhs_red'], rtol=rtol, max_iter=max_iter)
                out.append(ub.permute(2, 0, 1))
                info = (max(info[0], inf[0]), max(info[1], inf[1]))
        finally:
            eng.ctx.reduced_precond_use(None)
        U = torch.cat(out, dim=0).contiguous()
        return (U, info) if return_info else U

    def estimate_batch(self, U, mus):
        """Estimates of the reduced solutions U [len(mus), S, N] (as ``solve_batch`` returns them): list of eta."""
        eng = self.d.engine
        th = np.stack([self.d.theta(mu) for mu in mus])
        eta = eng.ctx.reduced_estimate_batch(self.d.Q, th, U.permute(1, 2, 0).contiguous(), self.out, eng.ops, eng.hdiam).cpu().numpy()
        return [self.d.combine(eta[:, :, m], mu) for m, mu in enumerate(mus)]

    def estimate(self, u, mu, decompose=False):
        eta_loc = self.d.engine.reduced_estimate(self.d.theta(mu), u.contiguous(), self.out)
        return self.d.combine(eta_loc.cpu().numpy(), mu, decompose)


class ExtensionError3D(Exception):
    """A vector handed to ``extend_basis`` is (numerically) in the span of a local basis (pyMOR's ExtensionError)."""


class LRBMSReductor3D:
    """``LRBMSReductor`` (reference reductor.py:17-78) for the 3D path: the same constructor and methods, local bases as ONE
    device slab [S, n, N_max] (ragged bases: zero columns behind the ``local_sizes()[s]`` vectors of a subdomain).

        LRBMSReductor3D(d, bases=None, products=None, order=None)
            bases      ready-made local bases: a slab [S or S_ext, n, N] / list of [n, N_s] arrays (reductor.py:22-27), or None
            products   {'domain_i': ...} of the reference is the local energy product of every subdomain (reductor.py:19,
                       online_adaptive_lrbms.py:107); here it is ``d.engine.ops['P_diag']`` (lrbms3_assemble_energy_product) and
                       the argument only switches it: None / anything = the energy product, 'euclidean' = plain dot products
            order      0 / 1: start every local basis with the shape functions of that order (reductor.py:29-31); with neither
                       ``bases`` nor ``order`` given: order 0, the reference's default (reductor.py:23-24) -- every local basis then
                       starts with the constant, which the coarse level of the reduced solver's preconditioner builds on.
                       Order 1 deviates from the reference (block_swipdg.py:195: uncentred x0, x1, x0*x1, a 2D list whose
                       projection call is undefined at HEAD): here the constant plus the three coordinates relative to the
                       subdomain centre, the 3D counterpart of "all polynomials of degree <= 1"
        extend_basis(U)            restrict a block DG function [S, n(, L)] to every subdomain, Gram-Schmidt it into the bases
        extend_basis_local(ii, U)  the same for ONE subdomain (reductor.py:31,78)
        reduce()                   one pass of the hot path (reductor.py:33-73)
        reconstruct(u), reconstruct_local(u, ii)

    Gram-Schmidt runs on the device: the product is applied by ``lrbms3_energy_product_apply``, the rest are batched products.
    ``enrich_local`` (reductor.py:75-78) needs the neighbourhood corrector solves, which exist in 2D only."""

    def __init__(self, d, bases=None, products=None, order=None):
        import torch
        self.d, self._torch = d, torch
        eng = d.engine
        self.euclidean = products == 'euclidean'
        self._V, self._nloc = None, None
        if order is None and bases is None:
            order = 0                                                    # reductor.py:23-24
        if bases is not None:
            if isinstance(bases, (list, tuple)):
                blocks = [b if isinstance(b, torch.Tensor) else eng.ctx.from_numpy(np.asarray(b)) for b in bases]
                nmax = max(int(b.shape[1]) for b in blocks)
                self._nloc = np.array([int(b.shape[1]) for b in blocks], dtype=np.int64)
                V = torch.stack([torch.nn.functional.pad(b, (0, nmax - int(b.shape[1]))) for b in blocks])
            else:
                V = bases if isinstance(bases, torch.Tensor) else eng.ctx.from_numpy(np.asarray(bases))
                self._nloc = np.full(V.shape[0], int(V.shape[2]), dtype=np.int64)
            assert V.shape[0] in (eng.S, eng.S_ext) and V.shape[1] == eng.t.n
            self._V = V.contiguous()
        if order is not None:
            if self._V is not None and self._V.shape[0] != eng.S:
                raise NotImplementedError('order= with ready-made bases that already carry their halo')
            sf = d.shape_functions(0, order)                            # the same template for every subdomain
            self._gram_schmidt_extend(sf[None].expand(eng.S, -1, -1).contiguous())

    # ------------------------------------------------------------------ bases
    @property
    def bases(self):
        """The basis slab [S (or S_ext), n, N_max] (device); columns >= local_sizes()[s] of subdomain s are zero."""
        return self._V

    def basis_size(self):
        return 0 if self._V is None else int(self._V.shape[2])

    def local_sizes(self):
        return [] if self._nloc is None else [int(v) for v in self._nloc[:self.d.engine.S]]

    def _product_apply(self, X):
        eng = self.d.engine
        if self.euclidean:
            return X
        return eng.ctx.energy_product_apply(eng.ops['P_diag'], X.contiguous())

    def gram(self):
        """V^T P V per subdomain [S, N_max, N_max]: the identity on the filled columns after Gram-Schmidt (what the 2D pass returns
        as E_red)."""
        V = self._V[:self.d.engine.S].contiguous()
        return self._torch.einsum('snk,snl->skl', V, self._product_apply(V))

    def _orthonormalize(self, v, atol=1e-13, rtol=1e-10):
        """Gram-Schmidt with one re-orthogonalisation of the single-column slab ``v`` [S, n, 1] against the local bases; returns the
        normalised slab and the mask of subdomains whose block was NOT (numerically) in the span of their basis."""
        torch = self._torch
        S = self.d.engine.S
        V = self._V[:S] if self._V is not None and self._V.shape[2] > 0 else None
        v = v.clone()
        norm0 = torch.sqrt(torch.clamp((v * self._product_apply(v)).sum(dim=(1, 2)), min=0.0))
        for _ in range(2):
            if V is not None:
                coef = torch.einsum('snk,snl->skl', V, self._product_apply(v))
                v = v - torch.einsum('snk,skl->snl', V, coef)
        norm = torch.sqrt(torch.clamp((v * self._product_apply(v)).sum(dim=(1, 2)), min=0.0))
        ok = (norm > atol) & (norm > rtol * norm0)
        v = torch.where(ok[:, None, None], v / torch.where(ok, norm, torch.ones_like(norm))[:, None, None], torch.zeros_like(v))
        return v, ok

    def _append(self, v, ok):
        torch = self._torch
        eng = self.d.engine
        ok_host = ok.cpu().numpy().astype(bool)
        if self._V is None:
            self._V = eng.ctx.zeros(eng.S, eng.t.n, 0)
            self._nloc = np.zeros(eng.S, dtype=np.int64)
        if self._V.shape[0] != eng.S:
            raise NotImplementedError('extending bases that carry their halo: extend the local slab and exchange afterwards')
        idx = np.where(ok_host)[0]
        if len(idx) == 0:
            return ok_host
        if int(self._nloc[idx].max()) + 1 > self._V.shape[2]:
            self._V = torch.cat([self._V, eng.ctx.zeros(eng.S, eng.t.n, 1)], dim=2).contiguous()
        rows = torch.as_tensor(idx, device=self._V.device)
        cols = torch.as_tensor(self._nloc[idx], device=self._V.device)
        self._V[rows, :, cols] = v[rows, :, 0]
        self._nloc[idx] += 1
        return ok_host

    def _gram_schmidt_extend(self, U):
        """Extend EVERY local basis by the columns of U [S, n, L], one after the other (all-or-nothing per column)."""
        for k in range(U.shape[2]):
            v, ok = self._orthonormalize(U[:, :, k:k + 1])
            if not bool(ok.all()):
                raise ExtensionError3D('snapshot block is (numerically) in the span of its local basis')
            self._append(v, ok)

    def _as_slab(self, U):
        eng = self.d.engine
        U = U if isinstance(U, self._torch.Tensor) else eng.ctx.from_numpy(np.asarray(U))
        if U.dim() == 2:
            U = U[:, :, None]
        assert tuple(U.shape[:2]) == (eng.S, eng.t.n), 'a block DG function [S, n] or [S, n, L]'
        return U.contiguous()

    def extend_basis(self, U):
        """Restrict the block DG function(s) ``U`` [S, n] / [S, n, L] (e.g. ``d.solve(mu)``) to every subdomain and extend all
        local bases (the fork's ``extend_basis``; online_adaptive_lrbms.py:117-121)."""
        self._gram_schmidt_extend(self._as_slab(U))

    def extend_basis_local(self, subdomain, U):
        """Extend the basis of ONE subdomain by the vector(s) ``U`` [n] / [n, L] (reductor.py:31,78)."""
        eng = self.d.engine
        i = eng.local.index(int(subdomain))
        U = U if isinstance(U, self._torch.Tensor) else eng.ctx.from_numpy(np.asarray(U))
        U = U[:, None] if U.dim() == 1 else U
        mask = self._torch.zeros(eng.S, dtype=self._torch.bool, device=U.device)
        mask[i] = True
        for k in range(U.shape[1]):
            full = eng.ctx.zeros(eng.S, eng.t.n, 1)
            full[i, :, 0] = U[:, k]
            v, ok = self._orthonormalize(full)
            if not bool(ok[i]):
                raise ExtensionError3D('local vector is (numerically) in the span of the local basis')
            self._append(v, ok & mask)

    def enrich_local(self, subdomain, U, mu=None):
        raise NotImplementedError('online enrichment (reductor.py:75-78: neighbourhood corrector solves) is built for the 2D path only')

    # ------------------------------------------------------------------ reduce / reconstruct
    def reduce(self):
        eng = self.d.engine
        if self._V is None or self._V.shape[2] == 0:
            raise RuntimeError('no basis')
        V = self._V
        if V.shape[0] != eng.S_ext:
            raise NotImplementedError('sharded discretization: hand in bases [S_ext, n, N] with the halo filled (HaloExchange)')
        return ReducedDiscretization3D(self, eng.project_and_estimate(V.contiguous()))

    def reconstruct(self, u):
        return self._torch.einsum('snj,sj->sn', self._V, u)

    def reconstruct_local(self, u, subdomain):
        """The block of ``reconstruct(u)`` on ONE subdomain [n] (reductor.py:76)."""
        i = self.d.engine.ext.index(int(subdomain))
        return self._V[i] @ u[i]


def discretize(grid_and_problem_data, device_index=0):
    d = BlockDiscretization3D(grid_and_problem_data, device_index=device_index)
    eng = d.engine
    data = {'grid': d.grid, 'engine': eng, 'operators': eng.ops}
    return d, data
