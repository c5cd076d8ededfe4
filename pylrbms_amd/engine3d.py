"""Host-side driver of the 3D / P2 HIP path (BASELINE.json config 5): owns the device arrays of one rank and sequences the
C-ABI calls of include/lrbms3d_hip.h.  3D counterpart of pylrbms_amd/engine.py (which cites the reference lines it replaces).

Data layout in HBM (fp64, C-contiguous; S local subdomains, S_ext = S + halo):

    lam    [Q][S_ext][n_T][LS]    coefficient samples at the points of the rules (grid3d.QuadratureSpec3D)
    A_diag [Q][S][n_T][5][100]    SWIPDG blocks, block-ELL over the element adjacency template (10 x 10 blocks, P2)
    P_diag [S][n_T][5][100]       local energy product at mu_bar (the same pattern; local to every subdomain)
    A_cpl  [Q][S][6][ncf][100]    coupling blocks per side face
    V      [S_ext][n][N]          local reduced bases, DoF-major
    outputs of a pass: Native3DContext.out_shapes (factored layout, include/lrbms3d_hip.h)
"""
import numpy as np

from pylrbms_amd._native import NativeError
from pylrbms_amd._native3d import Native3DContext
from pylrbms_amd.grid3d import SELF_SLOT, SIDE_TO_SLOT, QuadratureSpec3D


def sample(fn, x):
    return np.broadcast_to(np.asarray(fn(x), dtype=np.float64), x.shape[:-1])


class Engine3D:
    def __init__(self, grid, lambda_funcs, f, lambda_bar, lambda_hat, data_degree=2, device_index=0, theta_bar=None):
        """``theta_bar`` [Q]: theta_q(mu_bar), the weights of the local energy product (reference: assembled at mu_bar,
        discretize_elliptic_block_swipdg.py:676); default: all ones."""
        self.grid, self.t = grid, grid.template
        self.theta_bar = np.ones(len(lambda_funcs)) if theta_bar is None else np.asarray(theta_bar, dtype=np.float64).reshape(-1)
        t = self.t
        self.spec = QuadratureSpec3D(data_degree)
        local = list(grid.subdomains_on_rank)
        halo = sorted({j for s in local for j in grid.neighboring_subdomains(s)} - set(local))
        self.local, self.halo, self.ext = local, halo, local + halo
        pos = {g: i for i, g in enumerate(self.ext)}
        self.ext_pos = pos
        nbr = np.full((len(local), 7), -1, dtype=np.int32)
        for i, s in enumerate(local):
            for slot in range(7):
                g = grid.neighbor_slots[s, slot]
                if g >= 0:
                    nbr[i, slot] = pos[int(g)]
        self.nbr = nbr
        self.S, self.S_ext, self.Q = len(local), len(self.ext), len(lambda_funcs)
        self.hdiam = grid.subdomain_diameter(0)
        self.ctx = Native3DContext(device_index)
        self.ctx.mesh_upload(t, self.spec, t.tables(self.spec), nbr, grid.phys_mask[self.ext], self.S, self.S_ext)
        # ---- coefficient sampling on the host (as in 2D), one H2D copy per record
        xl, xh, xb = t.record_points(self.spec)
        org_ext = np.stack([grid.subdomain_origin(s) for s in self.ext])
        org = org_ext[:self.S]
        c = self.ctx
        self.lam = c.from_numpy(np.stack([np.stack([sample(fn, xl + o) for o in org_ext]) for fn in lambda_funcs]))
        self.lhat = c.from_numpy(np.stack([sample(lambda_hat, xh + o) for o in org]))
        self.f_smp = c.from_numpy(np.stack([sample(f, xh + o) for o in org]))
        self.lbar = c.from_numpy(np.stack([sample(lambda_bar, xb + o) for o in org]))
        self.ops = None

    def assemble(self):
        c = self.ctx
        A_diag, A_cpl = c.assemble_system(self.lam)
        b, f2, ceps, bdiv = c.assemble_rhs(self.f_smp, self.lhat)
        ebar, Aaa, Aab, Bbb = c.assemble_products(self.lam, self.lbar, self.lhat)
        Cf = c.assemble_flux(self.lam)
        P_diag = c.assemble_energy_product(self.theta_bar, self.lam)          # K6: local energy product at mu_bar
        self.ops = dict(A_diag=A_diag, A_cpl=A_cpl, b=b, f2=f2, ceps=ceps, bdiv=bdiv, ebar=ebar, Aaa=Aaa, Aab=Aab, Bbb=Bbb, Cf=Cf,
                        P_diag=P_diag)
        return self

    def alloc_outputs(self, N):
        shp = self.ctx.out_shapes(self.Q, N)
        return {k: self.ctx.empty(*shp[k]) for k in self.ctx.OUT_NAMES}

    def alloc_work(self, N):
        return self.ctx.empty(self.ctx.work_size(self.Q, N))

    def project_and_estimate(self, V, out=None, work=None, halo=None):
        """One pass of the hot path over all local subdomains; V [S_ext, n, N] with the halo filled -- or ``halo`` (a
        ``pylrbms_amd.parallel.HaloExchange`` on this rank's 3D tile) fills it: one exchange step per pass (the rows of the
        cube layer next to every foreign side), started first and waited for only by the kernels that read neighbour rows."""
        if self.ops is None:
            raise NativeError('assemble() must run before project_and_estimate()')
        N = V.shape[2]
        out = out if out is not None else self.alloc_outputs(N)
        work = work if work is not None else self.alloc_work(N)
        if halo is None:
            return self.ctx.project_estimate(self.Q, V, self.ops, work, out)
        # sharded: the collective runs while everything that reads rank-local slabs only is computed (all but the neighbours'
        # shares of the flux image and of the node averages and the coupling blocks: ~95 % of the pass)
        finish = halo.start(V)
        self.ctx.project_estimate(self.Q, V, self.ops, work, out, phase=1)
        finish()
        return self.ctx.project_estimate(self.Q, V, self.ops, work, out, phase=2)

    def reduced_estimate(self, theta, u, out):
        return self.ctx.reduced_estimate(self.Q, theta, u, out, self.ops, self.hdiam)

    def reduced_solve(self, theta, out, rtol=1e-13, max_iter=5000):
        return self.ctx.reduced_solve(self.Q, theta, out['B_sys'], out['rhs_red'], rtol=rtol, max_iter=max_iter)

    def interpolate(self, fn):
        """P2 nodal interpolant as a block DG vector [S, n] (host)."""
        x = self.t.node_coordinates()
        return np.stack([sample(fn, x + self.grid.subdomain_origin(s)) for s in self.local])


# ---------------------------------------------------------------------- layout converter (host / tests)
def expand_factored(engine, out, Q, N):
    """Factored outputs of a pass -> the dense operators of the neighbourhood formulation (what the reference's LRBMSReductor
    would hold after projecting nc_i, r_dd_i, df_bb_i, df_ab_i, r_fd_i through the image bases; columns slot-major, then q,
    then basis index), by a few batched products on the device.  For tests and for callers that want the blocks."""
    import torch
    t, S, QN = engine.t, engine.S, Q * N
    dev = out['G_nc'].device
    ops = engine.ops
    slots = SIDE_TO_SLOT
    # ---- nonconformity: w = W_self u_s - P z,  z = sum_a As_a u_a
    nb, nvs = t.nb, t.nvs
    Z = torch.zeros(S, nb, 7 * N, dtype=torch.float64, device=dev)              # z as a linear map of the slot-major coefficients
    bs = torch.as_tensor(t.bnode_sides, device=dev).long()
    As = out['As'].reshape(S, 6 * nvs, N)
    for k in range(3):
        sp = bs[:, k]
        ok = sp >= 0
        for a in range(6):
            sel = ok & (torch.div(sp, nvs, rounding_mode='floor') == a)
            if sel.any():
                Z[:, sel, slots[a] * N:(slots[a] + 1) * N] += As[:, sp[sel]]
    G_nc = torch.zeros(S, 7 * N, 7 * N, dtype=torch.float64, device=dev)
    G_nc[:, 3 * N:4 * N, 3 * N:4 * N] = out['G_nc']
    cross = torch.einsum('sbi,sbc->sic', out['Cn'], Z)                          # u_s^T Cn^T z
    G_nc[:, 3 * N:4 * N, :] += cross
    G_nc[:, :, 3 * N:4 * N] += cross.transpose(1, 2)
    ebar = ops['ebar'].reshape(S, t.n_T, 10, 10)
    bel = torch.as_tensor(t.bel_elem, device=dev).long()
    bb = torch.as_tensor(t.bel_bnode, device=dev).long()                         # [nbel, 10]
    Ze = Z[:, bb.clamp(min=0)] * (bb >= 0)[None, :, :, None]                      # [S, nbel, 10, 7N]
    G_nc += torch.einsum('seic,seij,sejd->scd', Ze, ebar[:, bel], Ze)
    # ---- flux: Re = R_self ur_s + Zf,  Zf = Rb ur_a on the side faces
    C = 7 * QN
    Zf = torch.zeros(S, t.nbf, C, dtype=torch.float64, device=dev)
    for a in range(6):
        Zf[:, a * t.ncf:(a + 1) * t.ncf, slots[a] * QN:(slots[a] + 1) * QN] = out['Rb'][:, a * t.ncf:(a + 1) * t.ncf]
    sel_e = torch.as_tensor(t.sel_elem, device=dev).long()
    sf = torch.as_tensor(t.sel_sf, device=dev).long()                            # [nsel, 4]
    Zfe = Zf[:, sf.clamp(min=0)] * (sf >= 0)[None, :, :, None]                    # [S, nsel, 4, C]
    Bbb = ops['Bbb'].reshape(S, t.n_T, 4, 4)[:, sel_e]
    sgn = torch.ones(S, t.n_T, 4, dtype=torch.float64, device=dev) * torch.as_tensor(t.tsign, device=dev)[None]
    nbe = torch.as_tensor(t.nb_elem, device=dev)
    phys = torch.as_tensor(engine.grid.phys_mask[engine.local], device=dev)
    side = (-(nbe + 1)).clamp(min=0)
    on_phys = (nbe < 0)[None] & (((phys[:, None, None] >> side[None]) & 1) == 1)
    sgn = torch.where(on_phys, torch.ones_like(sgn), sgn)
    divc = torch.as_tensor(t.divc[t.elem_type], device=dev)[None] * sgn          # [S, nT, 4]
    dZ = torch.einsum('sef,sefc->sec', divc[:, sel_e], Zfe)                      # div of the side part per side element
    s0 = slice(3 * QN, 4 * QN)

    def full(self_block, cross_rows, quad):
        G = torch.zeros(S, C, C, dtype=torch.float64, device=dev)
        G[:, s0, s0] = self_block
        cr = torch.einsum('sfc,sfd->scd', cross_rows, Zf)                        # ur^T rows^T Zf
        G[:, s0, :] += cr
        G[:, :, s0] += cr.transpose(1, 2)
        return G + quad
    G_bb = full(out['G_bb'], out['Yb'], torch.einsum('sefc,sefg,segd->scd', Zfe, Bbb, Zfe))
    G_rdd = full(out['G_rdd'], out['Dp'], t.volume * torch.einsum('sec,sed->scd', dZ, dZ))
    r_fd = torch.zeros(S, C, dtype=torch.float64, device=dev)
    r_fd[:, s0] = out['r_fd']
    r_fd += torch.einsum('se,sec->sc', ops['bdiv'][:, sel_e], dZ)
    G_ab = torch.zeros(Q, S, N, C, dtype=torch.float64, device=dev)
    G_ab[..., s0] = out['G_ab']
    G_ab += torch.einsum('qsfi,sfc->qsic', out['Xab'], Zf)
    return dict(G_nc=G_nc, G_bb=G_bb, G_rdd=G_rdd, r_fd=r_fd, G_ab=G_ab, G_aa=out['G_aa'], B_sys=out['B_sys'], rhs_red=out['rhs_red'])
