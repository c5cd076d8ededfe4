"""3D domain-decomposition grid of BASELINE.json config 5 (8 x 8 x 8 subdomains, SWIPDG p = 2): host index layer and
reference-element tables of the HIP kernels in csrc/lrbms3d.hip.

The reference binds the 2D / P1 operators only (python/dune/pylrbms/discretize_elliptic_block_swipdg.py:22-23, ``x[0], x[1]``
at :195); ``make_grid`` (python/dune/pylrbms/grid.py:18-30) is dimension independent in spirit (cube -> simplices ->
Cartesian subdomains), so the 3D grid keeps its queries (``num_subdomains``, ``neighborhood_of``, ``neighboring_subdomains``,
``boundary_subdomains``, ``subdomains_on_rank``) on the Kuhn (Freudenthal) triangulation: six tetrahedra per cube, translation
invariant, hence every subdomain is a translate of ONE template and -- new in 3D -- every element is a translate of one of
SIX reference tetrahedra.  That second invariance is what the kernels are built on: every local integral on the path is

    block[e] = sum_k  (coefficient sample at point k of e)  x  TABLE[type(e)][k]

with tables that depend on the element type only (basis values / gradients / normals / weights at the quadrature points, own and
neighbour side), i.e. assembly is a small dense contraction of the sample records with a table that stays in L2 / LDS.

Conventions (identical to the CPU restatement in oracle/mesh3d.py, which was written first):
* cubes of a subdomain x-fastest (``((cz ky + cy) kx + cx)``), the six tetrahedra of a cube in the order of
  ``itertools.permutations(range(3))`` (vertex path from the cube's lower corner, one unit step per axis in that order),
  vertices 1 and 2 swapped where that makes the orientation positive; local face f opposite vertex f; P2 local DoFs:
  4 vertices, then the midpoints of the edges (0,1) (0,2) (0,3) (1,2) (1,3) (2,3);
* subdomain id ``sx + Px (sy + Py sz)``; sides of a subdomain sorted by neighbour id: 0 = z-, 1 = y-, 2 = x-, 3 = x+, 4 = y+,
  5 = z+; neighbourhood slots 0..6 = sides 0..2, self (slot 3), sides 3..5;
* face quadrature points are generated from the face's vertices in ascending lattice order (x slowest) -- both elements of a
  face see the same points in the same order; RT0 orientation: outward from the element with the lower global index
  (side faces: the subdomain with the lower id; physical boundary: outward);
* RT0 / side-face / side-node numbering: first appearance over (element, local face) resp. ascending node id.
"""
import itertools

import numpy as np

PERMS = list(itertools.permutations(range(3)))
FACE_VERTS = np.array([[1, 2, 3], [0, 2, 3], [0, 1, 3], [0, 1, 2]])
EDGE_VERTS = np.array([[0, 1], [0, 2], [0, 3], [1, 2], [1, 3], [2, 3]])
NLOC = 10
SIDE_AXIS = (2, 1, 0, 0, 1, 2)          # axis normal to side a
SIDE_DIR = (-1, -1, -1, 1, 1, 1)
SIDE_TO_SLOT = (0, 1, 2, 4, 5, 6)
SELF_SLOT = 3
SIGMA_INNER_P2 = 20.0                    # dune-gdt swipdg inner_sigma(polorder <= 2)      [UPSTREAM-RECALL, SURVEY App. A.2]
SIGMA_BOUNDARY_P2 = 38.0
BETA_3D = 0.5                            # 1 / (d - 1)


# ------------------------------------------------------------------------------------------------- quadrature (host tables)
def gauss_jacobi01(n, alpha):
    """Nodes / weights on [0, 1] for the weight (1 - x)^alpha (Golub-Welsch on the Jacobi matrix, beta = 0)."""
    from math import gamma
    k = np.arange(n, dtype=np.float64)
    a = np.empty(n)
    a[0] = -alpha / (alpha + 2.0)
    if n > 1:
        kk = k[1:]
        a[1:] = -alpha ** 2 / ((2 * kk + alpha) * (2 * kk + alpha + 2))
    kk = np.arange(1, n, dtype=np.float64)
    b = np.sqrt(4 * kk * (kk + alpha) * kk * (kk + alpha) / ((2 * kk + alpha) ** 2 * (2 * kk + alpha + 1) * (2 * kk + alpha - 1)))
    J = np.diag(a) + np.diag(b, 1) + np.diag(b, -1)
    x, v = np.linalg.eigh(J)
    mu0 = 2.0 ** (alpha + 1) * gamma(alpha + 1) / gamma(alpha + 2)
    w = mu0 * v[0] ** 2
    return 0.5 * (x + 1.0), w / 2.0 ** (alpha + 1.0)


def rule_size(degree):
    return max(1, (int(degree) + 2) // 2)


def tet_rule(degree):
    """Stroud conical product rule on the tetrahedron: (barycentric points [k, 4], weights [k] summing to 1), exact for
    polynomials of the given degree with n^3 points, n = (degree + 2) // 2."""
    n = rule_size(degree)
    x0, w0 = gauss_jacobi01(n, 2.0)
    x1, w1 = gauss_jacobi01(n, 1.0)
    x2, w2 = gauss_jacobi01(n, 0.0)
    a, b, c = np.meshgrid(x0, x1, x2, indexing='ij')
    w = (w0[:, None, None] * w1[None, :, None] * w2[None, None, :]).ravel()
    x = a.ravel()
    y = (b * (1.0 - a)).ravel()
    z = (c * (1.0 - a) * (1.0 - b)).ravel()
    return np.stack([1.0 - x - y - z, x, y, z], axis=1), w * 6.0


def tri_rule(degree):
    n = rule_size(degree)
    x0, w0 = gauss_jacobi01(n, 1.0)
    x1, w1 = gauss_jacobi01(n, 0.0)
    a, b = np.meshgrid(x0, x1, indexing='ij')
    w = (w0[:, None] * w1[None, :]).ravel()
    x = a.ravel()
    y = (b * (1.0 - a)).ravel()
    return np.stack([1.0 - x - y, x, y], axis=1), w * 2.0


def p2_basis(lam):
    """P2 Lagrange basis at barycentric points [..., 4]: values [..., 10], derivatives w.r.t. the barycentrics [..., 10, 4]."""
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


class QuadratureSpec3D:
    """Degrees of the rules per integrand from the polynomial degree ``deg`` declared for the data functions (polynomial data
    are then integrated exactly): system volume deg + 2, system faces deg + 4, flux faces deg + 3, rhs / products deg + 4,
    estimator operators 3 deg + 4."""

    def __init__(self, data_degree=2):
        d = int(data_degree)
        self.data_degree = d
        self.system_volume, self.system_face, self.flux_face = d + 2, d + 4, d + 3
        self.product_volume, self.estimator_volume = d + 4, 3 * d + 4
        self.nA, self.nB, self.nC = (rule_size(x) ** 3 for x in (self.system_volume, self.product_volume, self.estimator_volume))
        self.nFs, self.nFf = rule_size(self.system_face) ** 2, rule_size(self.flux_face) ** 2
        # lambda_q record: volA | 4 x Fs | 4 x Ff | volC ;  lambda_hat: volB | volC ;  lambda_bar: volB ;  f: volB | volC
        self.o_fs, self.o_ff = self.nA, self.nA + 4 * self.nFs
        self.o_c = self.o_ff + 4 * self.nFf
        self.lam_stride = self.o_c + self.nC
        self.hat_stride = self.nB + self.nC
        self.f_stride = self.nB + self.nC


# ------------------------------------------------------------------------------------------------- template
def _kuhn_offsets():
    """[6][4][3] integer vertex offsets of the six tetrahedra of the unit cube, positively oriented."""
    out = np.zeros((6, 4, 3), dtype=np.int64)
    for t, perm in enumerate(PERMS):
        p = np.zeros(3, dtype=np.int64)
        out[t, 0] = p
        for i, d in enumerate(perm):
            p = p.copy()
            p[d] += 1
            out[t, i + 1] = p
        J = (out[t, 1:] - out[t, :1]).astype(np.float64)
        if np.linalg.det(J) < 0:
            out[t, [1, 2]] = out[t, [2, 1]]
    return out


class SubdomainTemplate3D:
    """Connectivity (int32) and reference-tetrahedron geometry (float64) shared by all subdomains."""

    def __init__(self, kc, h, kappa=None):
        kx, ky, kz = (int(v) for v in kc)
        self.kc = (kx, ky, kz)
        self.h = np.asarray(h, dtype=np.float64).reshape(3)
        self.kappa = np.eye(3) if kappa is None else np.asarray(kappa, dtype=np.float64).reshape(3, 3)
        off = _kuhn_offsets()
        self.kuhn = off
        nT = 6 * kx * ky * kz
        self.n_T, self.n = nT, NLOC * nT

        # ---- elements of the extended box (one layer of cubes around the subdomain), lattice vertex coordinates
        ex = np.array([(cx, cy, cz) for cz in range(-1, kz + 1) for cy in range(-1, ky + 1) for cx in range(-1, kx + 1)])
        lat = (ex[:, None, None, :] + off[None]).reshape(-1, 4, 3)                     # [nE, 4, 3]
        cube = np.repeat(ex, 6, axis=0)
        typ = np.tile(np.arange(6), len(ex))
        inside = np.all((cube >= 0) & (cube < np.array([kx, ky, kz])), axis=1)
        kvec = np.array([kx, ky, kz])
        cl = np.mod(cube, kvec)
        local_index = ((cl[:, 2] * ky + cl[:, 1]) * kx + cl[:, 0]) * 6 + typ          # index inside its OWN subdomain
        sub_off = np.floor_divide(cube, kvec)                                          # offset of the owning subdomain
        own = np.nonzero(inside)[0]
        assert np.array_equal(local_index[own], np.arange(nT))
        self.elem_lattice = lat[own]
        self.elem_type = typ[own].astype(np.int32)
        self.elem_cube = cube[own]

        def vkey(v):                                                                   # ascending = lattice order, x slowest
            return ((v[..., 0] + 1) * (ky + 3) + (v[..., 1] + 1)) * (kz + 3) + (v[..., 2] + 1)

        vk = vkey(lat)                                                                 # [nE, 4]
        fk = np.sort(vk[:, FACE_VERTS], axis=2)                                        # [nE, 4, 3]
        big = (ky + 3) * (kz + 3) * (kx + 3) + 1
        fkey = (fk[..., 0] * big + fk[..., 1]) * big + fk[..., 2]
        table = {}
        for E in range(len(lat)):
            for f in range(4):
                table.setdefault(int(fkey[E, f]), []).append((E, f))

        nb_elem = np.full((nT, 4), -1, dtype=np.int32)
        nb_face = np.full((nT, 4), -1, dtype=np.int32)
        nb_out = np.full((nT, 4), -1, dtype=np.int32)
        face_pos = np.full((nT, 4), -1, dtype=np.int32)
        side_lists = [[] for _ in range(6)]
        tsign = np.ones((nT, 4), dtype=np.int32)
        nb_type = np.zeros((6, 4), dtype=np.int32)
        nb_face_of_type = np.zeros((6, 4), dtype=np.int32)
        nb_cube_off = np.zeros((6, 4, 3), dtype=np.int64)
        for e, E in enumerate(own):
            for f in range(4):
                pair = table[int(fkey[E, f])]
                assert len(pair) == 2
                E2, f2 = pair[0] if pair[1][0] == E else pair[1]
                t = typ[E]
                nb_type[t, f], nb_face_of_type[t, f] = typ[E2], f2
                nb_cube_off[t, f] = cube[E2] - cube[E]
                nb_face[e, f] = f2
                if inside[E2]:
                    nb_elem[e, f] = local_index[E2]
                    tsign[e, f] = 1 if e < local_index[E2] else -1
                else:
                    so = sub_off[E2]
                    ax = int(np.nonzero(so)[0][0])
                    assert np.count_nonzero(so) == 1
                    side = {(2, -1): 0, (1, -1): 1, (0, -1): 2, (0, 1): 3, (1, 1): 4, (2, 1): 5}[(ax, int(so[ax]))]
                    nb_elem[e, f] = -(1 + side)
                    nb_out[e, f] = local_index[E2]
                    face_pos[e, f] = len(side_lists[side])
                    side_lists[side].append((e, f, int(local_index[E2]), f2))
                    tsign[e, f] = -1 if side < 3 else 1                               # the neighbour with the lower id is "minus"
        self.nb_elem, self.nb_face, self.nb_out, self.face_pos, self.tsign = nb_elem, nb_face, nb_out, face_pos, tsign
        # faces whose inner neighbour has the HIGHER element index (compact list, -1 padded): the symmetric form of the system
        # projection reads the diagonal block and these blocks only (A[e', e] = A[e, e']^T)
        up = np.full((nT, 4), -1, dtype=np.int32)
        for e in range(nT):
            fs = [f for f in range(4) if nb_elem[e, f] > e]
            up[e, :len(fs)] = fs
        self.up_face = up
        self.nb_type, self.nb_face_of_type, self.nb_cube_off = nb_type, nb_face_of_type, nb_cube_off
        ncf = max(len(s) for s in side_lists)
        self.ncf, self.nbf = ncf, 6 * ncf
        self.side_count = np.array([len(s) for s in side_lists], dtype=np.int32)
        self.side_elem = np.full((6, ncf), -1, dtype=np.int32)
        self.side_face = np.full((6, ncf), -1, dtype=np.int32)
        self.side_elem_out = np.full((6, ncf), -1, dtype=np.int32)
        self.side_face_out = np.full((6, ncf), -1, dtype=np.int32)
        for a in range(6):
            for p, (e, f, eo, fo) in enumerate(side_lists[a]):
                self.side_elem[a, p], self.side_face[a, p], self.side_elem_out[a, p], self.side_face_out[a, p] = e, f, eo, fo

        # ---- RT0 numbering: first appearance over (element, face)
        elem_rt = np.full((nT, 4), -1, dtype=np.int32)
        rt = []
        for e in range(nT):
            for f in range(4):
                if elem_rt[e, f] >= 0:
                    continue
                idx = len(rt)
                elem_rt[e, f] = idx
                e2 = nb_elem[e, f]
                if e2 >= 0:
                    elem_rt[e2, nb_face[e, f]] = idx
                    rt.append((e, f, e2, nb_face[e, f], -1, -1))
                else:
                    rt.append((e, f, -1, -1, -(e2 + 1), face_pos[e, f]))
        rt = np.array(rt, dtype=np.int32)
        self.elem_rt, self.n_rt = elem_rt, len(rt)
        self.rt_e0, self.rt_f0, self.rt_e1, self.rt_f1, self.rt_side, self.rt_pos = (np.ascontiguousarray(rt[:, i]) for i in range(6))

        # ---- P2 nodes on the doubled lattice
        own_lat = lat[own]
        nd = np.concatenate([2 * own_lat, own_lat[:, EDGE_VERTS[:, 0]] + own_lat[:, EDGE_VERTS[:, 1]]], axis=1)   # [nT, 10, 3]
        self.nodes_per_dim = (2 * kx + 1, 2 * ky + 1, 2 * kz + 1)
        nnx, nny, nnz = self.nodes_per_dim
        self.n_nodes = nnx * nny * nnz
        self.dof_node = ((nd[..., 0] * nny + nd[..., 1]) * nnz + nd[..., 2]).reshape(-1).astype(np.int32)
        self.dof_node_lattice = nd.reshape(-1, 3)
        order = np.argsort(self.dof_node, kind='stable')
        self.node_ptr = np.searchsorted(self.dof_node[order], np.arange(self.n_nodes + 1)).astype(np.int32)
        self.node_dofs = order.astype(np.int32)
        X, Y, Z = np.meshgrid(np.arange(nnx), np.arange(nny), np.arange(nnz), indexing='ij')
        X, Y, Z = X.ravel(), Y.ravel(), Z.ravel()
        mask = ((Z == 0) * 1 + (Y == 0) * 2 + (X == 0) * 4 + (X == nnx - 1) * 8 + (Y == nny - 1) * 16 + (Z == nnz - 1) * 32)
        self.node_mask = mask.astype(np.int32)
        # elements of the extended box at every own node, by the subdomain they belong to
        nd_all = np.concatenate([2 * lat, lat[:, EDGE_VERTS[:, 0]] + lat[:, EDGE_VERTS[:, 1]]], axis=1)            # [nE, 10, 3]
        ok = np.all((nd_all >= 0) & (nd_all <= np.array([2 * kx, 2 * ky, 2 * kz])), axis=2)                       # node of the own box
        nid_all = (nd_all[..., 0] * nny + nd_all[..., 1]) * nnz + nd_all[..., 2]
        nside = np.count_nonzero(sub_off, axis=1)
        side_of = np.full(len(lat), -2)                                                # -1 own, 0..5 face neighbour, -2 diagonal
        side_of[nside == 0] = -1
        for a in range(6):
            sel = (nside == 1) & (sub_off[:, SIDE_AXIS[a]] == SIDE_DIR[a])
            side_of[sel] = a
        cnt = np.zeros(self.n_nodes, dtype=np.int64)
        nvs = max(nnx * nny, nnx * nnz, nny * nnz)
        self.nvs = nvs
        self.side_nodes = np.full((6, nvs), -1, dtype=np.int32)
        self.side_node_count = np.zeros(6, dtype=np.int32)
        node_side_pos = np.full((self.n_nodes, 6), -1, dtype=np.int32)
        for a in range(6):
            on = np.nonzero(mask & (1 << a))[0]
            self.side_nodes[a, :len(on)] = on
            self.side_node_count[a] = len(on)
            node_side_pos[on, a] = np.arange(len(on))
        self.node_side_pos = node_side_pos
        sn_lists = [[[] for _ in range(nvs)] for _ in range(6)]
        for E in range(len(lat)):
            a = side_of[E]
            if a == -2:
                continue
            for i in range(NLOC):
                if not ok[E, i]:
                    continue
                g = int(nid_all[E, i])
                if a == -1:
                    cnt[g] += 1
                elif mask[g] & (1 << a):
                    cnt[g] += 1
                    sn_lists[a][node_side_pos[g, a]].append(NLOC * int(local_index[E]) + i)
        self.node_count = cnt.astype(np.int32)                                         # |patch| of a non-boundary node
        ptr, idx = [0], []
        for a in range(6):
            for p in range(nvs):
                idx.extend(sn_lists[a][p])
                ptr.append(len(idx))
        self.sn_ptr, self.sn_dofs = np.array(ptr, dtype=np.int32), np.array(idx, dtype=np.int32)
        # boundary nodes of the subdomain and the (side, pos) pairs they belong to
        bn = np.nonzero(mask)[0]
        self.nb = len(bn)
        self.bnodes = bn.astype(np.int32)
        self.node_bnode = np.full(self.n_nodes, -1, dtype=np.int32)
        self.node_bnode[bn] = np.arange(len(bn))
        bs = np.full((len(bn), 3), -1, dtype=np.int32)
        for i, g in enumerate(bn):
            sides = [a for a in range(6) if mask[g] & (1 << a)]
            for j, a in enumerate(sides):
                bs[i, j] = a * nvs + node_side_pos[g, a]
        self.bnode_sides = bs
        # elements with a node on the subdomain boundary (nonconformity side term) / with a side face (flux side terms)
        dofb = self.node_bnode[self.dof_node].reshape(nT, NLOC)
        bel = np.nonzero(np.any(dofb >= 0, axis=1))[0]
        self.bel_elem, self.bel_bnode = bel.astype(np.int32), np.ascontiguousarray(dofb[bel])
        # compact numbering of the DoFs that sit on a boundary node (rows of E W_self the side-node factors are summed from)
        onb = dofb.reshape(-1) >= 0
        self.nbd = int(onb.sum())
        self.dof_bslot = np.where(onb, np.cumsum(onb) - 1, -1).astype(np.int32)
        ptr, idx = [0], []
        for g in bn:                                     # boundary node -> compact slots of its own DoFs (k3_side_nc)
            idx.extend(int(self.dof_bslot[d]) for d in self.node_dofs[self.node_ptr[g]:self.node_ptr[g + 1]])
            ptr.append(len(idx))
        self.bn_ptr, self.bn_slots = np.array(ptr, dtype=np.int32), np.array(idx, dtype=np.int32)
        sf = np.where(nb_elem < 0, (-(nb_elem + 1)) * ncf + face_pos, -1)
        sel = np.nonzero(np.any(sf >= 0, axis=1))[0]
        self.sel_elem, self.sel_sf = sel.astype(np.int32), np.ascontiguousarray(sf[sel].astype(np.int32))
        # traversal order of the element loops of the pass: cubes in 2 x 2 x 2 blocks, so that most face neighbours of an element
        # (whose basis rows it reads) are visited within the next few dozen items and still sit in the L2
        cb = self.elem_cube
        key = (((cb[:, 2] // 2) * ((ky + 1) // 2) + cb[:, 1] // 2) * ((kx + 1) // 2) + cb[:, 0] // 2) * 8 + \
            ((cb[:, 2] % 2) * 2 + cb[:, 1] % 2) * 2 + cb[:, 0] % 2
        self.order = np.argsort(key * 6 + self.elem_type, kind='stable').astype(np.int32)
        self._geometry()

    # ------------------------------------------------------------------ reference tetrahedra
    def _geometry(self):
        h = self.h
        X = self.kuhn.astype(np.float64) * h                                           # [6, 4, 3] physical, cube at the origin
        J = X[:, 1:] - X[:, :1]
        self.volume = float(np.linalg.det(J[0]) / 6.0)
        assert np.allclose(np.linalg.det(J) / 6.0, self.volume) and self.volume > 0
        Jinv = np.linalg.inv(J)
        g = np.empty((6, 4, 3))
        g[:, 1:] = np.transpose(Jinv, (0, 2, 1))
        g[:, 0] = -g[:, 1:].sum(axis=1)
        self.type_vertices, self.grad_lambda = X, g
        Vf = X[:, FACE_VERTS]                                                          # [6, 4(face), 3, 3]
        nrm = np.cross(Vf[:, :, 1] - Vf[:, :, 0], Vf[:, :, 2] - Vf[:, :, 0])
        self.face_area = 0.5 * np.linalg.norm(nrm, axis=2)                             # [6, 4]
        nrm = nrm / np.linalg.norm(nrm, axis=2)[..., None]
        opp = X                                                                        # vertex f is opposite face f
        flip = np.einsum('tfa,tfa->tf', nrm, Vf[:, :, 0] - opp) < 0
        nrm[flip] *= -1.0
        self.face_normal = nrm                                                         # outward
        self.divc = self.face_area / self.volume                                       # |f| / |T|
        self.psic = self.face_area / (3.0 * self.volume)
        self.diam_sub = float(np.linalg.norm(h * np.array(self.kc)))

    def bary(self, t, x, cube_off=(0, 0, 0)):
        """Barycentric coordinates of physical points x [..., 3] in the type-t tetrahedron of the cube at lattice offset."""
        x0 = self.type_vertices[t, 0] + np.asarray(cube_off) * self.h
        lam = np.einsum('ia,...a->...i', self.grad_lambda[t], x - x0)
        lam[..., 0] += 1.0
        return lam

    def face_points(self, t, f, degree):
        """Physical points [k, 3] of the face rule on face f of the type-t tetrahedron (cube at the origin): generated from the
        face's vertices in ascending lattice order, so that both elements of a face see the same points."""
        lat = self.kuhn[t][FACE_VERTS[f]]
        key = (lat[:, 0] * 4 + lat[:, 1]) * 4 + lat[:, 2]
        V = (lat[np.argsort(key)] * self.h)
        bary, w = tri_rule(degree)
        return bary @ V, w

    def tables(self, spec):
        """Reference tables of the assembly contraction (module docstring), float64:
          TV   [6][nA][100]          w |T| grad phi_i . kappa grad phi_j                     (system volume, rule A)
          TE   [6][nB][100]          the same at rule B                                       (lambda_bar product E)
          TAA  [6][nC][100]          the same at rule C                                       (df_aa)
          TFo  [6][4][nFs][100]      inner face, (own, own) block      TFn: (own, neighbour) block      TFb: Dirichlet face
          TPo, TPn, TPb              the PENALTY parts of TFo, TFn, TFb alone (local energy product, block_swipdg.py:651-677)
          TC   [6][4][nFf][10]       flux coefficients of the own element on an inner face   TCb: on a Dirichlet face
          TPH  [6][nB][10]           w |T| phi_i   (rhs)               TM [6][100]  mass
          TB   [6][nC][16]           w |T| psi_f . kappa^-1 psi_g   (unsigned)               TAB [6][nC][40]  w |T| grad phi_i . psi_f
        """
        K, Kinv = self.kappa, np.linalg.inv(self.kappa)
        vol = self.volume
        out = {}

        def vol_tab(degree):
            bary, w = tet_rule(degree)
            phi, dphi = p2_basis(bary)
            tab = np.empty((6, len(w), NLOC, NLOC))
            for t in range(6):
                grad = np.einsum('kiv,va->kia', dphi, self.grad_lambda[t])
                tab[t] = np.einsum('k,kia,ab,kjb->kij', w * vol, grad, K, grad)
            return tab.reshape(6, len(w), 100), bary, w, phi, dphi

        out['TV'] = vol_tab(spec.system_volume)[0]
        TE, baryB, wB, phiB, _ = vol_tab(spec.product_volume)
        out['TE'] = TE
        out['TPH'] = np.broadcast_to((wB[:, None] * vol * phiB)[None], (6, len(wB), NLOC)).copy()
        out['TM'] = np.broadcast_to(np.einsum('k,ki,kj->ij', wB * vol, phiB, phiB).reshape(1, 100), (6, 100)).copy()
        out['WB'] = wB * vol
        TAA, baryC, wC, phiC, dphiC = vol_tab(spec.estimator_volume)
        out['TAA'] = TAA
        out['WC'] = wC * vol
        TB, TAB = np.empty((6, len(wC), 4, 4)), np.empty((6, len(wC), NLOC, 4))
        for t in range(6):
            X = self.type_vertices[t]
            x = baryC @ X
            psi = self.psic[t][None, :, None] * (x[:, None, :] - X[None, :, :])        # [k, 4, 3] unsigned
            grad = np.einsum('kiv,va->kia', dphiC, self.grad_lambda[t])
            TB[t] = np.einsum('k,kfa,ab,kgb->kfg', wC * vol, psi, Kinv, psi)
            TAB[t] = np.einsum('k,kia,kfa->kif', wC * vol, grad, psi)
        out['TB'], out['TAB'] = TB.reshape(6, len(wC), 16), TAB.reshape(6, len(wC), 40)

        nFs, nFf = spec.nFs, spec.nFf
        TFo, TFn, TFb = (np.empty((6, 4, nFs, NLOC, NLOC)) for _ in range(3))
        TPo, TPn, TPb = (np.empty((6, 4, nFs, NLOC, NLOC)) for _ in range(3))
        TC, TCb = np.empty((6, 4, nFf, NLOC)), np.empty((6, 4, nFf, NLOC))
        for t in range(6):
            for f in range(4):
                n, area = self.face_normal[t, f], self.face_area[t, f]
                delta = n @ K @ n
                s_in = SIGMA_INNER_P2 * 0.5 * delta / area ** BETA_3D
                s_bd = SIGMA_BOUNDARY_P2 * delta / area ** BETA_3D
                t2, off2 = self.nb_type[t, f], self.nb_cube_off[t, f]
                for degree, which in ((spec.system_face, 'sys'), (spec.flux_face, 'flux')):
                    xq, w = self.face_points(t, f, degree)
                    po, do_ = p2_basis(self.bary(t, xq))
                    pn, dn_ = p2_basis(self.bary(t2, xq, off2))
                    d_o = np.einsum('kiv,va,ab,b->ki', do_, self.grad_lambda[t], K.T, n)      # kappa grad phi . n_own
                    d_n = np.einsum('kiv,va,ab,b->ki', dn_, self.grad_lambda[t2], K.T, n)
                    if which == 'sys':
                        ww = (w * area)[:, None, None]
                        TFo[t, f] = ww * (-0.5 * po[:, :, None] * d_o[:, None, :] - 0.5 * d_o[:, :, None] * po[:, None, :]
                                          + s_in * po[:, :, None] * po[:, None, :])
                        TFn[t, f] = ww * (-0.5 * po[:, :, None] * d_n[:, None, :] + 0.5 * d_o[:, :, None] * pn[:, None, :]
                                          - s_in * po[:, :, None] * pn[:, None, :])
                        TFb[t, f] = ww * (-po[:, :, None] * d_o[:, None, :] - d_o[:, :, None] * po[:, None, :]
                                          + s_bd * po[:, :, None] * po[:, None, :])
                        TPo[t, f] = ww * (s_in * po[:, :, None] * po[:, None, :])
                        TPn[t, f] = ww * (-s_in * po[:, :, None] * pn[:, None, :])
                        TPb[t, f] = ww * (s_bd * po[:, :, None] * po[:, None, :])
                    else:
                        TC[t, f] = w[:, None] * (-0.5 * d_o + s_in * po)
                        TCb[t, f] = w[:, None] * (-d_o + s_bd * po)
        out['TFo'], out['TFn'], out['TFb'] = (x.reshape(6, 4, nFs, 100) for x in (TFo, TFn, TFb))
        out['TPo'], out['TPn'], out['TPb'] = (x.reshape(6, 4, nFs, 100) for x in (TPo, TPn, TPb))
        out['TC'], out['TCb'] = TC, TCb
        return out

    # ------------------------------------------------------------------ sample points of one subdomain (origin at 0)
    def record_points(self, spec):
        """Physical points (relative to the subdomain origin) of the three sample records:
        lambda [n_T, lam_stride, 3], hat [n_T, hat_stride, 3] (also the f record), bar [n_T, nB, 3]."""
        org = self.elem_cube * self.h                                                  # [nT, 3]
        X = self.type_vertices[self.elem_type] + org[:, None, :]                       # [nT, 4, 3]
        bA, bB, bC = tet_rule(spec.system_volume)[0], tet_rule(spec.product_volume)[0], tet_rule(spec.estimator_volume)[0]
        xA, xB, xC = (np.einsum('kv,evd->ekd', b, X) for b in (bA, bB, bC))
        lam = np.empty((self.n_T, spec.lam_stride, 3))
        lam[:, :spec.nA] = xA
        lam[:, spec.o_c:] = xC
        for t in range(6):
            sel = self.elem_type == t
            for f in range(4):
                lam[sel, spec.o_fs + f * spec.nFs:spec.o_fs + (f + 1) * spec.nFs] = \
                    self.face_points(t, f, spec.system_face)[0][None] + org[sel][:, None, :]
                lam[sel, spec.o_ff + f * spec.nFf:spec.o_ff + (f + 1) * spec.nFf] = \
                    self.face_points(t, f, spec.flux_face)[0][None] + org[sel][:, None, :]
        return lam, np.concatenate([xB, xC], axis=1), xB

    def node_coordinates(self):
        """Physical coordinates [n, 3] of the Lagrange node of every local DoF (P2 nodal interpolation)."""
        return self.dof_node_lattice * (0.5 * self.h)


# ------------------------------------------------------------------------------------------------- grid
def tile_grid3d(world_size, P):
    """Factorisation (tx, ty, tz) of world_size into a tile grid that divides P and minimises the halo surface."""
    best = None
    for tx in range(1, world_size + 1):
        if world_size % tx or P[0] % tx:
            continue
        for ty in range(1, world_size // tx + 1):
            if (world_size // tx) % ty or P[1] % ty:
                continue
            tz = world_size // tx // ty
            if P[2] % tz:
                continue
            a, b, c = P[0] // tx, P[1] // ty, P[2] // tz
            cost = a * b + b * c + a * c
            if best is None or cost < best[0]:
                best = (cost, (tx, ty, tz))
    if best is None:
        raise ValueError('cannot tile {} subdomains over {} ranks'.format(tuple(P), world_size))
    return best[1]


class DDSubdomainsGrid3D:
    """``K`` cubes per direction cut into ``P`` subdomains per direction (the queries of reference grid.py:8-69)."""

    def __init__(self, lower_left, upper_right, num_cubes, num_partitions, rank=0, world_size=1, kappa=None):
        self.lower_left = np.asarray(lower_left, dtype=np.float64)
        self.upper_right = np.asarray(upper_right, dtype=np.float64)
        self.K = np.asarray(num_cubes, dtype=np.int64)
        self.P = np.asarray(num_partitions, dtype=np.int64)
        assert self.K.shape == (3,) and self.P.shape == (3,) and np.all(self.K % self.P == 0)
        self.kc = self.K // self.P
        self.h = (self.upper_right - self.lower_left) / self.K
        self.template = SubdomainTemplate3D(self.kc, self.h, kappa)
        self.num_subdomains = int(self.P.prod())
        self.dim = 3
        Px, Py, Pz = (int(v) for v in self.P)
        s = np.arange(self.num_subdomains)
        sx, sy, sz = s % Px, (s // Px) % Py, s // (Px * Py)
        self.sub_coords = np.stack([sx, sy, sz], axis=1)
        slots = np.full((self.num_subdomains, 7), -1, dtype=np.int64)
        slots[:, SELF_SLOT] = s
        phys = np.zeros(self.num_subdomains, dtype=np.int32)
        for a in range(6):
            c = self.sub_coords.copy()
            c[:, SIDE_AXIS[a]] += SIDE_DIR[a]
            ok = np.all((c >= 0) & (c < self.P), axis=1)
            slots[ok, SIDE_TO_SLOT[a]] = (c[ok, 0] + Px * (c[ok, 1] + Py * c[ok, 2]))
            phys |= np.where(ok, 0, 1 << a).astype(np.int32)
        self.neighbor_slots = slots
        self.phys_mask = phys                                                          # bit a: side a lies on the physical boundary
        self.rank, self.world_size = int(rank), int(world_size)
        self._on_rank = self._partition(self.rank, self.world_size)

    def _partition(self, rank, world_size):
        if world_size == 1:
            return list(range(self.num_subdomains))
        tx, ty, tz = tile_grid3d(world_size, [int(v) for v in self.P])
        a, b, c = int(self.P[0]) // tx, int(self.P[1]) // ty, int(self.P[2]) // tz
        rx, ry, rz = rank % tx, (rank // tx) % ty, rank // (tx * ty)
        sel = np.all((self.sub_coords // np.array([a, b, c])) == np.array([rx, ry, rz]), axis=1)
        return [int(v) for v in np.nonzero(sel)[0]]

    @property
    def subdomains_on_rank(self):
        return list(self._on_rank)

    def neighboring_subdomains(self, ii):
        r = self.neighbor_slots[ii]
        return [int(v) for k, v in enumerate(r) if v >= 0 and k != SELF_SLOT]

    def neighborhood_of(self, ii):
        return sorted(int(v) for v in self.neighbor_slots[ii] if v >= 0)

    @property
    def boundary_subdomains(self):
        return [int(v) for v in np.nonzero(self.phys_mask)[0]]

    def subdomain_origin(self, ii):
        return self.lower_left + self.sub_coords[ii] * self.kc * self.h

    def subdomain_diameter(self, ii=0):
        return float(np.linalg.norm((self.upper_right - self.lower_left) / self.P))


def make_grid3d(domain=([0, 0, 0], [1, 1, 1]), num_subdomains=(2, 2, 2), cubes_per_subdomain_and_dim=4, rank=0, world_size=1,
                kappa=None):
    """3D counterpart of ``make_grid`` (reference grid.py:18-30): ``num_subdomains`` Cartesian subdomains of
    ``cubes_per_subdomain_and_dim``^3 cubes (six tetrahedra each)."""
    P = np.asarray(num_subdomains, dtype=np.int64)
    kc = np.broadcast_to(np.asarray(cubes_per_subdomain_and_dim, dtype=np.int64), (3,))
    return DDSubdomainsGrid3D(domain[0], domain[1], P * kc, P, rank=rank, world_size=world_size, kappa=kappa)
