"""Host-side driver of the HIP hot path: owns the device arrays of one rank and sequences the C-ABI calls.

Data layout in HBM (all fp64, C-contiguous; S = local subdomains, S_ext = S + halo):

    lam    [Q][S_ext][n_T][LS]   coefficient samples at the points of the quadrature rules (pylrbms_amd/quadrature.py)
    A_diag [Q][S][n_T][4][9]     SWIPDG diagonal blocks, block-ELL over the element adjacency template
    A_cpl  [Q][S][4][ncf][9]     SWIPDG coupling blocks per side face
    V      [S_ext][n][N]         local reduced bases, DoF-major (basis index contiguous)
    Wt     [S][n][5N]            Oswald image bases restricted to the target subdomain (slot-major columns)
    Rt     [S][n_rt][5QN]        RT0 flux-reconstruction image bases (slot, q, basis) columns
    B_sys  [Q][S][5][N][N]       projected system blocks; G_* projected estimator operators (see include/lrbms_hip.h;
                                 G_rdd / G_bb are block-compact [S][9][QN][QN]: only the structurally non-zero blocks)

The reference builds the same quantities as pyMOR operators in ``discretize`` / ``LRBMSReductor._reduce``
(discretize_elliptic_block_swipdg.py:530-811, reductor.py:33-73).
"""
import numpy as np

from pylrbms_amd._native import NativeContext, NativeError
from pylrbms_amd.quadrature import QuadratureSpec, edge_rule, native_quadrature, triangle_rule


def element_points(grid, subdomains):
    """Vertex coordinates of every element of the given subdomains [len, n_T, 3, 2], centres [len, n_T, 2], keys."""
    t = grid.template
    origins = np.stack([grid.subdomain_origin(int(s)) for s in subdomains])           # [len, 2]
    pts = t.points[None] + origins[:, None, None, :]                                    # [len, n_T, 3, 2]
    return pts, pts.mean(axis=2), grid.element_keys(subdomains)


def volume_points(pts, order):
    """Points of the triangle rule of a requested order on every element: [len, n_T, k, 2]."""
    bary, _ = triangle_rule(order)
    return np.einsum('kv,sevd->sekd', bary, pts)


def face_points(pts, f, order):
    """Points of the edge rule on local face f (from local vertex f + 1 to f + 2): [len, n_T, k, 2]."""
    tt, _ = edge_rule(order)
    a, b = pts[:, :, (f + 1) % 3], pts[:, :, (f + 2) % 3]
    return a[:, :, None, :] + tt[None, None, :, None] * (b - a)[:, :, None, :]


def sample_function(fn, x, centers, keys):
    c = np.broadcast_to(centers[:, :, None, :], x.shape)
    k = np.broadcast_to(keys[:, :, None, :], x.shape[:-1] + (3,))
    out = np.asarray(fn(x, c, k), dtype=np.float64)
    return np.broadcast_to(out, x.shape[:-1])


def lambda_record_points(grid, subdomains, spec):
    """x [len, n_T, lam_stride, 2] of the lambda_q sample record (layout: pylrbms_amd/quadrature.py native_quadrature):
    system volume | 3 faces x nfs system face points (inner rule on faces with an in-subdomain neighbour, coupling rule on
    the subdomain boundary; unused slots repeat the first point) | 3 x energy face | 3 x flux face | energy volume."""
    t = grid.template
    qd = native_quadrature(spec)
    pts, centers, keys = element_points(grid, subdomains)
    x = np.empty(pts.shape[:2] + (qd.lam_stride, 2))
    x[:, :, qd.o_sysv:qd.o_sysv + qd.system_volume.n] = volume_points(pts, spec.system_volume)
    inner = t.nb_elem >= 0                                                              # [n_T, 3]
    for f in range(3):
        xi, xc = face_points(pts, f, spec.system_inner_face), face_points(pts, f, spec.system_coupling_face)
        blk = np.repeat(xc[:, :, :1], qd.nfs, axis=2)
        blk[:, :, :xc.shape[2]] = xc
        blk_i = np.repeat(xi[:, :, :1], qd.nfs, axis=2)
        blk_i[:, :, :xi.shape[2]] = xi
        x[:, :, qd.o_sysf + f * qd.nfs:qd.o_sysf + (f + 1) * qd.nfs] = np.where(inner[None, :, f, None, None], blk_i, blk)
        ne, nf = qd.energy_face.n, qd.flux_face.n
        x[:, :, qd.o_enf + f * ne:qd.o_enf + (f + 1) * ne] = face_points(pts, f, spec.energy_face)
        x[:, :, qd.o_flf + f * nf:qd.o_flf + (f + 1) * nf] = face_points(pts, f, spec.flux_face)
    x[:, :, qd.o_env:qd.o_env + qd.energy_volume.n] = volume_points(pts, spec.energy_volume)
    return x, centers, keys


def volume_record_points(grid, subdomains, orders):
    """Concatenated volume points of several rules: x [len, n_T, sum k, 2], centres, keys."""
    pts, centers, keys = element_points(grid, subdomains)
    return np.concatenate([volume_points(pts, o) for o in orders], axis=2), centers, keys


class Engine:
    """All device state of one rank for one discretization."""

    SERIAL_PHASES_FROM = 384      # subdomains per rank from which a sharded pass runs its two halves on ONE stream (project_and_estimate)

    def __init__(self, grid, lambda_funcs, kappa, f, lambda_bar, lambda_hat, theta_bar, device_index=0, conventions=None,
                 quadrature=None):
        """``quadrature``: a ``QuadratureSpec`` (default: the reference's orders for these data functions,
        ``QuadratureSpec.for_problem``; ``QuadratureSpec.uniform(5)`` is the round-1 convention)."""
        self._init_args = (lambda_funcs, kappa, f, lambda_bar, lambda_hat, theta_bar, device_index, conventions, quadrature)
        self.conventions = dict(conventions or {})
        self.quadrature = quadrature if quadrature is not None else QuadratureSpec.for_problem(lambda_funcs, f, lambda_bar,
                                                                                                lambda_hat)
        self.grid = grid
        t = grid.template
        self.t = t
        local = list(grid.subdomains_on_rank)
        lset = set(local)
        # (with the Oswald vertex patch the diagonal neighbours are read too -- one DoF row per element at the shared cross point)
        diagonal = bool(self.conventions.get('oswald_vertex_patch'))
        if hasattr(grid, 'halo_subdomains'):
            halo = grid.halo_subdomains(diagonal=diagonal)
        else:
            halo = sorted({j for s in local for j in grid.neighboring_subdomains(s)} - lset)
        self.local, self.halo = local, halo
        self.ext = local + halo
        pos = {g: i for i, g in enumerate(self.ext)}
        self.ext_pos = pos
        nbr = np.full((len(local), 5), -1, dtype=np.int32)
        for i, s in enumerate(local):
            for slot in range(5):
                g = grid.neighbor_slots[s, slot]
                if g >= 0:
                    nbr[i, slot] = pos[int(g)]
        self.nbr = nbr
        self.S, self.S_ext = len(local), len(self.ext)
        self.Q = len(lambda_funcs)
        kap = np.asarray(getattr(kappa, 'value', kappa), dtype=np.float64).reshape(2, 2)
        self.kappa = kap
        self.ctx = NativeContext(device_index)
        self.ctx.mesh_upload(t, kap, nbr, self.S, self.S_ext)
        for name, value in self.conventions.items():          # conventions the reference leaves open (LRBMS_OPT_*)
            self.ctx.set_option(name, value)
        if diagonal and hasattr(grid, 'diagonal_neighbors'):
            self.nbr_diag = np.array([[pos.get(g, -1) if g >= 0 else -1 for g in grid.diagonal_neighbors(s)] for s in local],
                                     dtype=np.int32).reshape(len(local), 4)
            self.ctx.set_diagonal_neighbours(self.nbr_diag)
        self.ctx.set_quadrature(self.quadrature)
        self.hdiam = grid.subdomain_diameter(0)

        # ---- coefficient sampling on the host at the points of the chosen rules, one H2D copy each
        sp = self.quadrature
        x, c, k = lambda_record_points(grid, self.ext, sp)
        self.lam = self.ctx.from_numpy(np.ascontiguousarray(np.stack([sample_function(fn, x, c, k) for fn in lambda_funcs])))
        xd, cl, kl = volume_record_points(grid, self.local, (sp.df_aa, sp.df_ab))
        self.lam_df = self.ctx.from_numpy(np.ascontiguousarray(np.stack([sample_function(fn, xd, cl, kl) for fn in lambda_funcs])))
        xh, _, _ = volume_record_points(grid, self.local, (sp.df_aa, sp.df_ab, sp.df_bb, sp.ceps))
        self.lhat = self.ctx.from_numpy(np.ascontiguousarray(sample_function(lambda_hat, xh, cl, kl)))
        xf, _, _ = volume_record_points(grid, self.local, (sp.rhs, sp.f2))
        self.f_smp = self.ctx.from_numpy(np.ascontiguousarray(sample_function(f, xf, cl, kl)))
        xb, _, _ = volume_record_points(grid, self.local, (sp.elliptic_bar,))
        self.lbar = self.ctx.from_numpy(np.ascontiguousarray(sample_function(lambda_bar, xb, cl, kl)))
        self.theta_bar = np.asarray(theta_bar, dtype=np.float64)
        self.assembled = False

    # ------------------------------------------------------------------ offline assembly (K1-K6, K8a, K9)
    def assemble(self):
        c = self.ctx
        self.A_diag, self.A_cpl = c.assemble_swipdg(self.lam)
        self.b, self.f2, self.ceps = c.assemble_rhs(self.f_smp, self.lhat)
        self.P_diag, self.ebar, self.caa, self.Aab, self.Bbb = c.assemble_products(self.theta_bar, self.lam, self.lam_df,
                                                                                  self.lbar, self.lhat)
        self.F = c.assemble_flux(self.lam)
        self.assembled = True
        return self

    # ------------------------------------------------------------------ timed region: K7 + K8 + P1 + P2
    def alloc_outputs(self, N, factored=None):
        """The projected system and the projected estimator operators of one pass.  ``factored`` (default: whenever the
        fused pass supports (Q, N)): blocks of df_bb / r_dd / df_ab that involve a neighbour slot are returned as their
        rank-<=ncf factors ``F_side`` and the blocks of nc that do as their rank-<=nvs factors ``F_nc`` (8 tensors,
        include/lrbms_hip.h) -- 0.64 GB of outputs at config 3 instead of 1.75 GB for the dense block-compact layout
        (6 tensors), and the reduced estimate reads 0.3 instead of 1.5 GB."""
        c, S, Q = self.ctx, self.S, self.Q
        W, C, QN = 5 * N, 5 * Q * N, Q * N
        if factored is None:
            factored = c.fused_supported(Q, N, factored=True)
        sys_out = (c.empty(Q, S, 5, N, N), c.empty(S, N), c.empty(S, N, N), c.empty(S, N, N))
        if factored:
            grams = (c.empty(S, N, N), c.empty(S, C), c.empty(S, QN, QN), c.empty(S, QN, QN), c.empty(Q, S, N, QN),
                     c.empty(Q, Q, S, N, N), c.empty(S, 4, self.t.ncf, c.fside_ld(Q, N)), c.empty(S, 4, c.nvs, c.fnc_ld(N)))
        else:
            grams = (c.empty(S, W, W), c.empty(S, C), c.empty(S, 9, QN, QN), c.empty(S, 9, QN, QN), c.empty(Q, S, N, C),
                     c.empty(Q, Q, S, N, N))
        return {'sys': sys_out, 'grams': grams}

    def alloc_reduce_buffers(self, N, images=None, factored=None):
        """Outputs + scratch of ``project_and_estimate``.  The padded image bases ``Wt`` / ``Rt`` (1.3 GB at config 3) are
        only materialised by the unfused kernels: ``images=None`` allocates them iff the fused pass cannot run."""
        c, S, Q, n, n_rt = self.ctx, self.S, self.Q, self.t.n, self.t.n_rt
        W, C = 5 * N, 5 * Q * N
        if images is None:
            images = not c.fused_supported(Q, N, factored=True)
        work = c.empty(max(c.estimator_work_size(Q, N), Q * S * n * N, c.fused_work_size(Q, N)))
        buf = {'N': N, 'Wt': c.empty(S, n, W) if images else None, 'Rt': c.empty(S, n_rt, C) if images else None, 'work': work}
        buf.update(self.alloc_outputs(N, factored=factored))
        return buf

    def touched_targets(self, changed_global):
        """Local indices (ascending) of the target subdomains whose projected operators depend on the basis of one of the
        subdomains ``changed_global`` (global ids, on any rank): the changed subdomains themselves and their face neighbours --
        the operators of target ii are built from the bases of ii and of its neighbourhood (reductor.py:40-60, block_swipdg.py:78)
        -- plus the diagonal neighbours when the Oswald patch is the whole vertex star (conventions oswald_vertex_patch)."""
        g = self.grid
        hit = set()
        for m in changed_global:
            m = int(m)
            hit.add(m)
            nbrs = [int(j) for j in g.neighbor_slots[m] if j >= 0]
            hit.update(nbrs)
            if self.conventions.get('oswald_vertex_patch'):
                for j in nbrs:                                            # diagonal = neighbour of a neighbour across the other axis
                    hit.update(int(k) for k in g.neighbor_slots[j] if k >= 0)
        pos = {gid: i for i, gid in enumerate(self.local)}
        return sorted(pos[gid] for gid in hit if gid in pos)

    def project_and_estimate(self, V, buffers=None, project_system=True, fused=None, halo=None, subset=None):
        """One pass of the hot path over all local subdomains -- or, with ``subset`` (ascending local indices, fused pass only), over
        those subdomains only, writing their rows into ``buffers`` and leaving every other row as it is (incremental
        re-projection after online enrichment: ``touched_targets``; bit-identical to the rows a whole pass writes).  ``V`` [S_ext, n, N] must already hold the halo -- or
        ``halo`` (a ``pylrbms_amd.parallel.HaloExchange``) is given and fills it: with the fused pass the exchange then
        runs on the communication stream while the halo-independent kernels (more than half of the pass) are computed on
        the main stream; the kernels that read neighbour rows start on a side stream as soon as the halo has arrived.
        ``fused=None`` picks the fused pass (csrc/fused.hip) whenever the library supports (Q, N) and falls back to the
        unfused HIP kernels otherwise (both are GPU paths; the unfused one also materialises the image bases Wt, Rt)."""
        if not self.assembled:
            raise NativeError('assemble() must run before project_and_estimate()')
        N = V.shape[2]
        c = self.ctx
        if fused is None:
            cache = self.__dict__.setdefault('_fused_ok', {})
            if N not in cache:
                cache[N] = c.fused_supported(self.Q, N, factored=True)
            fused = cache[N]
        buf = buffers if buffers is not None else self.alloc_reduce_buffers(N, factored=bool(fused))
        if buf['N'] != N:
            raise NativeError('buffers were allocated for N={}'.format(buf['N']))
        if not fused and len(buf['grams']) == 8:
            raise NativeError('the unfused kernels write the dense layout: allocate the buffers with factored=False')
        if subset is not None:
            if not fused or buffers is None:
                raise NativeError('subset= needs the fused pass and the buffers of an earlier whole pass')
            if len(subset) == 0:
                if halo is not None:
                    halo(V)
                return buf
            c.fused_set_subset(subset)
            try:
                return self.project_and_estimate(V, buffers, project_system=project_system, fused=fused, halo=halo)
            finally:
                c.fused_set_subset(None)
        if fused:
            args = (V, self.F, self.A_diag, self.A_cpl, self.P_diag, self.b, self.ebar, self.caa, self.Aab, self.Bbb,
                    buf['work'], buf['sys'], buf['grams'])
            # argument checks / pointer marshalling once per (V, buffers) pair: a sharded step makes three library calls on
            # ~0.2 ms of device work, so the host side of a step matters (tools/phase_time.py)
            key = (V.data_ptr(), buf['work'].data_ptr(), buf['grams'][0].data_ptr(), buf['sys'][0].data_ptr())
            bound = self.__dict__.get('_bound_pass')
            if bound is None or bound[0] != key:
                bound = (key, c.bind_project_estimate_fused(*args))
                self._bound_pass = bound
            run = bound[1]
            if halo is None:
                run(0)
            else:
                torch = c.torch
                if self.S >= self.SERIAL_PHASES_FROM:
                    # many subdomains per rank: every kernel fills the chip, and the halo-dependent half beside the dense kernels
                    # only gets in their way (512 subdomains: 0.408 ms overlapped, 0.385 one phase after the other).  The
                    # exchange still has the whole halo-independent half to complete in.
                    finish = halo.start(V)
                    run(1)
                    finish()
                    run(2)
                    return buf
                main = torch.cuda.current_stream()
                side = c.aux_stream(0)     # a library stream, not a fresh one: HIP maps streams onto few hardware queues
                if main.cuda_stream == side.cuda_stream:    # (the caller works on library stream 0 itself: nothing to run beside)
                    finish = halo.start(V)
                    run(1)
                    finish()
                    run(2)
                    return buf
                # The whole exchange lives on library stream 0: pack (it reads the local slabs, so it waits for what the main stream
                # has queued so far -- one event), asynchronous collective, wait, unpack into V[S:].  The main stream goes straight
                # to the preparation: with the pack in front of it the chain pack -> preparation -> projection kernel was 8 us
                # longer per step at the 8-GPU tile.
                if '_step_events' not in self.__dict__:
                    self._step_events = (torch.cuda.Event(),)
                ready, = self._step_events
                ready.record(main)
                torch.cuda.set_stream(side)                 # (a set_stream pair: the context manager + wait_stream cost 20 us)
                try:
                    side.wait_event(ready)
                    halo.start(V)()
                finally:
                    torch.cuda.set_stream(main)
                # ONE library call for the step: preparation of the own basis + the dense kernels (local slabs only) on the main
                # stream; R_side, Avg_side, the thin kernels and the coupling blocks on library stream 0, behind the unpack and
                # behind the preparation, as soon as the halo is there; joined into the main stream (tools/host_step_time.py: 24 + 19
                # us of host time for the two calls and 5 for the host's own event pair before, against ~30 for this one)
                run(5, main.cuda_stream)
            return buf
        if halo is not None:
            halo(V)
        if buf.get('Wt') is None:      # the unfused kernels materialise the image bases
            buf['Wt'] = c.empty(self.S, self.t.n, 5 * N)
            buf['Rt'] = c.empty(self.S, self.t.n_rt, 5 * self.Q * N)
        c.oswald_apply(V, out=buf['Wt'])
        c.flux_reconstruct(self.F, V, out=buf['Rt'])
        if project_system:
            c.project_system(V, self.A_diag, self.A_cpl, self.P_diag, self.b, work=buf['work'], out=buf['sys'])
        c.estimator_grams(V, buf['Wt'], buf['Rt'], self.ebar, self.caa, self.Aab, self.Bbb, self.b, work=buf['work'],
                          out=buf['grams'])
        return buf

    # ------------------------------------------------------------------ online
    def reduced_estimate(self, theta, u, grams):
        return self.ctx.reduced_estimate(theta, u, grams, self.f2, self.ceps, self.hdiam)

    def reduced_solve(self, theta, B_sys, rhs_red, rtol=1e-13, max_iter=20000):
        return self.ctx.reduced_solve(theta, B_sys, rhs_red, rtol=rtol, max_iter=max_iter)

    # ------------------------------------------------------------------ online enrichment (section 8f "next" #1)
    def local_corrections(self, theta, marked, rtol=1e-12, max_iter=20000):
        """Neighbourhood corrector solves for the subdomains ``marked`` (local indices): [len(marked), n] + info."""
        if not self.assembled:
            raise NativeError('assemble() must run before local_corrections()')
        if self.S_ext != self.S and not getattr(self, '_is_hood', False):
            # sharded: the neighbourhood of a local subdomain reaches into the halo, whose operator blocks this engine does
            # not hold.  The corrector problems run on a second engine whose LOCAL set is this rank's local + halo
            # subdomains (same leading order, so local indices agree); its coefficients are sampled and its blocks assembled
            # here, without communication (assembly needs coefficient samples only).
            if getattr(self, '_hood_engine', None) is None:
                import copy
                g2 = copy.copy(self.grid)
                g2._on_rank = list(self.ext)
                self._hood_engine = Engine(g2, *self._init_args).assemble()
                self._hood_engine._is_hood = True
            corr, info = self._hood_engine.local_corrections(theta, marked, rtol=rtol, max_iter=max_iter)
            return corr, info
        if getattr(self, 'D_corr', None) is None:
            self.D_corr = self.ctx.assemble_dirichlet_correction(self.lam)
        return self.ctx.local_correction_solve(theta, marked, self.A_diag, self.A_cpl, self.D_corr, self.b, rtol=rtol,
                                               max_iter=max_iter)


# ---------------------------------------------------------------------- layout converters (host, for API / tests)
def expand_factored_grams(grams, ncf=None):
    """Factored layout (8 tensors, include/lrbms_hip.h) -> dense layout (6 tensors: G_rdd / G_bb block-compact
    [S, 9, QN, QN], G_ab [Q, S, N, 5QN]) with a few batched products on the device; a 6-tuple is returned unchanged.
    For callers that want the blocks themselves (``rd.operators``, storage, tests) -- the estimate kernels never need it."""
    if len(grams) == 6:
        return tuple(grams)
    import torch
    Gnc_s, r_fd, Gd_s, Gb_s, Gab_s, G_aa, Fs, Fn = grams
    Q, S, N, QN = Gab_s.shape[0], Gab_s.shape[1], Gab_s.shape[2], Gab_s.shape[3]
    nvs = Fn.shape[2]
    if Fn.shape[3] != 2 * N + 4 * nvs:
        raise NotImplementedError('F_nc carries the diagonal subdomains (conventions={"oswald_vertex_patch": True}): the dense '
                                  'block layout has five slots per neighbourhood and cannot hold them -- use the estimates')
    A, Cn, M = Fn[..., :N], Fn[..., N:2 * N], Fn[..., 2 * N:].reshape(S, 4, nvs, 4, nvs)
    G_nc = torch.zeros(S, 5 * N, 5 * N, dtype=Fs.dtype, device=Fs.device)
    G_nc[:, 2 * N:3 * N, 2 * N:3 * N] = Gnc_s
    slots = (0, 1, 3, 4)
    a_s = torch.einsum('sapi,sapj->saij', A, Cn)                      # [a, self] blocks
    a_b = torch.einsum('sapi,sapbr,sbrj->sabij', A, M, A)             # [a, b] blocks
    for a, sl in enumerate(slots):
        G_nc[:, sl * N:(sl + 1) * N, 2 * N:3 * N] = a_s[:, a]
        G_nc[:, 2 * N:3 * N, sl * N:(sl + 1) * N] = a_s[:, a].transpose(1, 2)
        for b, sl2 in enumerate(slots):
            G_nc[:, sl * N:(sl + 1) * N, sl2 * N:(sl2 + 1) * N] = a_b[:, a, b]
    Ra, Yb, Dp = Fs[..., :QN], Fs[..., QN:2 * QN], Fs[..., 2 * QN:3 * QN]
    Xab = Fs[..., 3 * QN:4 * QN].reshape(S, 4, Fs.shape[2], Q, N)
    sc0, sc1 = Fs[..., 4 * QN], Fs[..., 4 * QN + 1]
    G_bb = torch.empty(S, 9, QN, QN, dtype=Fs.dtype, device=Fs.device)
    G_rdd = torch.empty_like(G_bb)
    G_bb[:, 0], G_rdd[:, 0] = Gb_s, Gd_s
    G_bb[:, 1:5] = torch.einsum('sapr,sapc->sarc', Ra, Yb)
    G_rdd[:, 1:5] = torch.einsum('sapr,sapc->sarc', Ra, Dp)
    G_bb[:, 5:9] = torch.einsum('sapr,sap,sapc->sarc', Ra, sc0, Ra)
    G_rdd[:, 5:9] = torch.einsum('sapr,sap,sapc->sarc', Ra, sc1, Ra)
    G_ab = torch.zeros(Q, S, N, 5 * QN, dtype=Fs.dtype, device=Fs.device)
    G_ab[..., 2 * QN:3 * QN] = Gab_s
    side = torch.einsum('sapqi,sapc->qsaic', Xab, Ra)
    for a, slot in enumerate((0, 1, 3, 4)):
        G_ab[..., slot * QN:(slot + 1) * QN] = side[:, :, a]
    return G_nc, r_fd, G_rdd, G_bb, G_ab, G_aa



def blockell_to_dense(template, vals):
    """[n_T][4][9] block-ELL values of one subdomain -> dense [n, n] (inspection / ``.matrix()`` of the API shim)."""
    t = template
    vals = np.asarray(vals).reshape(t.n_T, 4, 3, 3)
    out = np.zeros((t.n, t.n))
    for e in range(t.n_T):
        out[3 * e:3 * e + 3, 3 * e:3 * e + 3] += vals[e, 0]
        for f in range(3):
            nb = t.nb_elem[e, f]
            if nb >= 0:
                out[3 * e:3 * e + 3, 3 * nb:3 * nb + 3] += vals[e, 1 + f]
    return out


def coupling_to_dense(template, vals, side):
    """[ncf][9] coupling blocks of one (subdomain, side) -> dense [n, n] block (rows: own DoFs, cols: neighbour's)."""
    t = template
    vals = np.asarray(vals).reshape(t.ncf, 3, 3)
    out = np.zeros((t.n, t.n))
    for p in range(t.side_count[side]):
        ei, eo = t.side_elem[side, p], t.side_elem_out[side, p]
        out[3 * ei:3 * ei + 3, 3 * eo:3 * eo + 3] += vals[p]
    return out
