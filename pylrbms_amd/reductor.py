"""``LRBMSReductor`` (reference python/dune/pylrbms/reductor.py:17-78) on the HIP hot path.

``reduce()`` is the timed region of the benchmark: Oswald image bases (reductor.py:40-43), flux-reconstruction image
bases (:51-60) and the Galerkin projection of the system and of every estimator operator (:70, the fork's
``project_system``) run as HIP kernels over all local subdomains at once; the reduced operators stay block-sparse
(S x 5 blocks) instead of the reference's dense ``unblock``-ed ``(sum N)^2`` matrices.

``extend_basis`` restates the fork's ``GenericRBSystemReductor.extend_basis`` (absent from the tree; SURVEY.md
section 8a row B1): per subdomain Gram-Schmidt against the local basis w.r.t. the supplied product (the local energy
product, online_adaptive_lrbms.py:105-108), with re-orthogonalisation; products are applied with ``lrbms_blockell_apply``.
This implementation keeps the local basis size uniform over subdomains (one HBM slab ``[S][n][N]``): an extension is
all-or-nothing and raises ``ExtensionError`` if any block of the snapshot is numerically in the span of its basis.
"""
import numpy as np

from pylrbms_amd.vectorarrays import BlockVectorArray, BlockVectorSpace, ReducedVectorArray


class ExtensionError(Exception):
    """pymor.core.exceptions.ExtensionError (caught at online_adaptive_lrbms.py:116-119)."""


class ReducedDiscretization:
    """``rd``: block-sparse reduced system + projected estimator operators, resident in HBM."""

    def __init__(self, reductor, buffers, N):
        import torch
        self.reductor, self.d, self.N = reductor, reductor.d, N
        eng = self.d.engine
        self.B_sys, self.rhs_red, self.E_red, self.M_red = (x.clone() for x in buffers['sys'])
        self.grams = tuple(x.clone() for x in buffers['grams'])
        self.estimator = self.d.estimator
        self.parameter_space = self.d.parameter_space

        class _Space:
            dim = eng.grid.num_subdomains * N
        self.solution_space = _Space
        self.operator = type('Op', (), {'source': _Space, 'range': _Space})
        self.operators = {'nc': self.grams[0], 'r_fd': self.grams[1], 'r_dd': self.grams[2], 'df_bb': self.grams[3],
                          'df_ab': self.grams[4], 'df_aa': self.grams[5], 'local_energy_dg_product': self.E_red}
        self.products = {'l2': self.M_red}
        self._torch = torch

    def parse_parameter(self, mu):
        return self.d.parse_parameter(mu)

    def solve(self, mu, inverse_options=None):
        """``rd.solve(mu)`` (online_adaptive_lrbms.py:141): (sum_q theta_q A_q^red) u = b^red.  The reference does a
        dense LU of the unblocked matrix; here block-Jacobi PCG on the block-sparse system (``lrbms_reduced_solve``)."""
        eng = self.d.engine
        if eng.S_ext != eng.S:
            raise NotImplementedError('reduced solve on a sharded discretization: gather B_sys / rhs_red first')
        u, info = eng.reduced_solve(self.d.theta(mu), self.B_sys, self.rhs_red)
        self.last_solve_info = info
        return ReducedVectorArray(u.reshape(eng.S, self.N, 1))

    def _local_estimates(self, U, mu):
        torch = self._torch
        eng = self.d.engine
        theta = self.d.theta(mu)
        cols = []
        for k in range(len(U)):
            u = U.tensor[:, :, k].contiguous()
            if eng.S_ext != eng.S:
                # halo coefficients: every rank contributes its rows to a global [S_total, N] table (one all-reduce
                # of S_total * N doubles), then picks the rows of its ext ordering
                import torch.distributed as dist
                table = torch.zeros(eng.grid.num_subdomains, self.N, dtype=u.dtype, device=u.device)
                table[torch.as_tensor(eng.local, device=u.device)] = u
                dist.all_reduce(table, group=getattr(self.d.mpi_comm, 'group', None))
                u = table[torch.as_tensor(eng.ext, device=u.device)].contiguous()
            cols.append(eng.reduced_estimate(theta, u, self.grams))
        eta = torch.stack(cols, dim=2)
        return eta[0], eta[1], eta[2]

    def estimate(self, U, mu=None, decompose=False):
        """``rd.estimate(u, mu=, decompose=)``: same code path as the full-order estimate (estimators.py:45-112) with
        the reduced operators (reduced OI = identity, reduced FR = theta-weighted identities, reductor.py:44-66)."""
        return self.estimator.estimate(U, self.parse_parameter(mu), self, decompose=decompose)


class LRBMSReductor:

    def __init__(self, d, bases=None, products=None, order=None, num_cpus=1, solver_options=None):
        assert order is None or 0 <= order <= 1
        self.d = d
        self.solver_options = solver_options
        self.products = products            # the local energy products; applied through the engine's P_diag
        self.num_cpus = num_cpus            # accepted and ignored, as in the reference (reductor.py:19)
        eng = d.engine
        self._V = None                      # [S, n, N] device tensor
        if bases is not None:
            self._V = self._bases_to_tensor(bases)
        if order is None and bases is None:
            order = 0
        if order is not None:
            for ii in eng.local:            # reductor.py:29-31
                self.extend_basis_local(d.shape_functions(ii, order), _defer=True)
            self._flush_local()

    # ------------------------------------------------------------------ bases
    def _bases_to_tensor(self, bases):
        import torch
        eng = self.d.engine
        blocks = []
        for ii in eng.local:
            b = bases['domain_{}'.format(ii)]
            t = b.tensor[0] if isinstance(b, BlockVectorArray) else eng.ctx.from_numpy(np.asarray(b).T)
            blocks.append(t)
        if len({tuple(b.shape) for b in blocks}) != 1:
            raise NotImplementedError('local bases of different sizes (uniform N only)')
        return torch.stack(blocks).contiguous()

    @property
    def bases(self):
        """dict space id -> array, as ``reductor.bases`` (keys 'domain_i'; after reduce() also 'OI_i', 'RT_i')."""
        eng = self.d.engine
        out = {}
        for i, ii in enumerate(eng.local):
            space = BlockVectorSpace([self.d.solution_space.subspaces[i]])
            out['domain_{}'.format(ii)] = BlockVectorArray(self._V[i:i + 1], space) if self._V is not None else None
        out.update(getattr(self, '_image_bases', {}))
        return out

    def basis_size(self):
        return 0 if self._V is None else int(self._V.shape[2])

    def _product_apply(self, X):
        """P X with P the local energy product (block-ELL) -- the product handed in at online_adaptive_lrbms.py:107."""
        return self.d.engine.ctx.blockell_apply(self.d.engine.P_diag, X.contiguous())

    def _gram_schmidt_extend(self, U, atol=1e-13, rtol=1e-10):
        """Orthonormalise the columns of U [S, n, L] against the current local bases and each other."""
        import torch
        V = self._V
        for k in range(U.shape[2]):
            v = U[:, :, k:k + 1].clone()
            norm0 = torch.sqrt((v * self._product_apply(v)).sum(dim=(1, 2)))
            for _ in range(2):                                           # re-orthogonalise once
                if V is not None:
                    coef = torch.einsum('snk,snl->skl', V, self._product_apply(v))
                    v = v - torch.einsum('snk,skl->snl', V, coef)
            norm = torch.sqrt(torch.clamp((v * self._product_apply(v)).sum(dim=(1, 2)), min=0.0))
            if bool(((norm <= atol) | (norm <= rtol * norm0)).any()):
                raise ExtensionError('snapshot block is (numerically) in the span of its local basis')
            v = v / norm[:, None, None]
            V = v if V is None else torch.cat([V, v], dim=2)
        self._V = V.contiguous()

    def extend_basis(self, U):
        """Restrict a global snapshot to every subdomain and extend all local bases (fork ``extend_basis``)."""
        self._gram_schmidt_extend(U.tensor)

    def extend_basis_local(self, U, _defer=False):
        """Reference reductor.py:31,78: extend the basis of the one subdomain ``U`` lives on.  Uniform sizes are kept by
        collecting one vector per subdomain before the slab grows (``_defer``); a lone local extension is not
        supported in this round (it belongs to online enrichment)."""
        if not _defer:
            raise NotImplementedError('lone local basis extension (online enrichment) is not in this round')
        self._pending = getattr(self, '_pending', [])
        self._pending.append(U.tensor)

    def _flush_local(self):
        import torch
        pend = getattr(self, '_pending', [])
        if pend:
            self._gram_schmidt_extend(torch.cat(pend, dim=0))
            self._pending = []

    # ------------------------------------------------------------------ reduce
    def reduce(self):
        return self._reduce()

    def _reduce(self):
        d = self.d
        eng = d.engine
        if self._V is None:
            raise RuntimeError('no basis')
        N = self.basis_size()
        V = d._with_halo(self._V)
        if getattr(self, '_buffers', None) is None or self._buffers['N'] != N:
            self._buffers = eng.alloc_reduce_buffers(N)
        buf = eng.project_and_estimate(V, self._buffers)
        self._image_bases = {'OI': buf['Wt'], 'RT': buf['Rt']}          # target-major image bases (device tensors)
        return ReducedDiscretization(self, buf, N)

    def reconstruct(self, u):
        import torch
        coef = u.tensor if hasattr(u, 'tensor') else u
        return BlockVectorArray(torch.einsum('snk,skl->snl', self._V, coef), self.d.solution_space)

    def reconstruct_local(self, u, space_id):
        i = self.d.engine.local.index(int(space_id.split('_')[1]))
        rec = self.reconstruct(u)
        return rec.block(i)

    def enrich_local(self, subdomain, U, mu=None):
        """Reference reductor.py:75-78."""
        raise NotImplementedError('online enrichment is SURVEY.md section 8f "next" #1')


class ParallelLRBMSReductor(LRBMSReductor):
    """Reference reductor.py:81-146: at HEAD ``_reduce`` returns the serial result (the Allreduce code after the
    early ``return`` at :125 is dead), so this is the same reductor with an ``mpi_comm``."""

    def __init__(self, d, bases=None, products=None, order=None, solver_options=None, mpi_comm=None):
        super().__init__(d, bases=bases, products=products, solver_options=solver_options, num_cpus=1, order=order)
        self.mpi_comm = mpi_comm
