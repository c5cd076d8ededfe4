"""Structured domain-decomposition grid: host index layer.

Mirrors ``dune.pylrbms.grid`` (reference python/dune/pylrbms/grid.py:8-69): ``make_grid`` returns an
object answering the queries the hot path uses -- ``num_subdomains``, ``subdomains_on_rank``,
``neighborhood_of`` / ``neighboring_subdomains`` (discretize_elliptic_block_swipdg.py:78,421),
``boundary_subdomains`` (:393) -- for the cube grid with two conforming refinements
(8 triangles per coarse square) cut into ``num_partitions`` Cartesian subdomains.

MI355X-first layout decision: every subdomain of such a grid is a translate of ONE template
(``k x k`` coarse squares).  All connectivity (element adjacency, face / RT0 numbering, vertex
stars for the Oswald interpolation, coupling-face pairing with the four neighbours) is therefore
stored once as a few KB of int32 that the HIP kernels keep in LDS / scalar cache, and every
per-subdomain array in HBM is a dense, fixed-stride slab.  The index maps are integer-exact and
are compared bit for bit with the generic (hash-based) oracle in tests/test_grid.py.

Conventions (documented in DESIGN.md section 3)
* lattice: spacing = half a coarse square; every lattice point is a vertex.
* elements of a subdomain: coarse squares row-major (x fastest), 8 triangles per square going
  counter-clockwise round the square's boundary ring starting at its lower-left corner;
  triangle = (centre, ring[t], ring[t+1]); local DoF v sits at vertex v; local face f is
  opposite vertex f (so face 0 lies on the boundary of the coarse square).
* subdomain id = sx + Px * sy; neighbourhood slots in sorted order: 0=S, 1=W, 2=self, 3=E, 4=N.
* sides of a subdomain: 0=S, 1=W, 2=E, 3=N  (slot = side if side < 2 else side + 1).
* RT0 / face numbering of a subdomain: first appearance over (element, local face).
* face orientation: normal with n_x > 0, or n_x == 0 and n_y > 0; domain-boundary faces point outward.
"""
import numpy as np

RING = np.array([(0, 0), (1, 0), (2, 0), (2, 1), (2, 2), (1, 2), (0, 2), (0, 1)], dtype=np.int64)
SIDE_TO_SLOT = (0, 1, 3, 4)
SLOT_TO_SIDE = (0, 1, -1, 2, 3)
INNER_BOUNDARY_SEGMENT_INDEX = 18446744073709551573  # 2**64 - 43, grid.py:11


class SubdomainTemplate:
    """Connectivity + reference geometry shared by all subdomains (int32 / float64 arrays)."""

    def __init__(self, kx, ky, hx, hy):
        self.kx, self.ky, self.hx, self.hy = int(kx), int(ky), float(hx), float(hy)
        kx, ky = self.kx, self.ky
        nT = 8 * kx * ky
        self.n_T, self.n = nT, 3 * nT
        nvx, nvy = 2 * kx + 1, 2 * ky + 1
        self.nvx, self.nvy = nvx, nvy
        self.n_vertices = nvx * nvy

        # ---- elements: lattice coordinates of the three vertices, closed form
        e = np.arange(nT)
        sq, t = e // 8, e % 8
        cx, cy = sq % kx, sq // kx
        lat = np.zeros((nT, 3, 2), dtype=np.int64)
        lat[:, 0, 0], lat[:, 0, 1] = 2 * cx + 1, 2 * cy + 1
        lat[:, 1, :] = np.stack([2 * cx, 2 * cy], axis=1) + RING[t]
        lat[:, 2, :] = np.stack([2 * cx, 2 * cy], axis=1) + RING[(t + 1) % 8]
        self.tri_lattice = lat
        self.elem_square = np.stack([cx, cy, t], axis=1)
        self.dof_vertex = (lat[:, :, 0] + nvx * lat[:, :, 1]).reshape(-1).astype(np.int32)

        # ---- reference geometry (translation invariant)
        p = lat.astype(np.float64) * np.array([self.hx, self.hy])
        self.points = p
        e1, e2 = p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]
        det = e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]
        assert np.all(det > 0)
        self.area = 0.5 * det
        g = np.zeros((nT, 3, 2))
        for i in range(3):
            d = p[:, (i + 2) % 3] - p[:, (i + 1) % 3]
            g[:, i, 0], g[:, i, 1] = -d[:, 1] / det, d[:, 0] / det
        self.grad = g
        gn = np.linalg.norm(g, axis=2)
        self.normal = -g / gn[:, :, None]                 # outward unit normal of face f
        self.face_len = 2.0 * self.area[:, None] * gn     # |e_f|

        # ---- faces: sort edge keys, pair up; first-appearance numbering via np.unique
        a = self.dof_vertex.reshape(nT, 3)[:, [1, 2, 0]]  # face f joins vertices f+1, f+2
        b = self.dof_vertex.reshape(nT, 3)[:, [2, 0, 1]]
        key = (np.minimum(a, b).astype(np.int64) * self.n_vertices + np.maximum(a, b)).reshape(-1)
        uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
        rank = np.empty(len(uniq), dtype=np.int64)
        rank[np.argsort(first, kind='stable')] = np.arange(len(uniq))
        self.elem_rt = rank[inv].reshape(nT, 3).astype(np.int32)
        self.n_rt = len(uniq)
        # neighbour across each face: the other (element, face) with the same key
        order = np.argsort(key, kind='stable')
        ks = key[order]
        same_next = np.zeros(len(ks), dtype=bool)
        same_next[:-1] = ks[1:] == ks[:-1]
        partner = np.full(3 * nT, -1, dtype=np.int64)
        i0 = np.nonzero(same_next)[0]
        partner[order[i0]] = order[i0 + 1]
        partner[order[i0 + 1]] = order[i0]
        nb_elem = np.where(partner >= 0, partner // 3, -1).reshape(nT, 3)
        nb_face = np.where(partner >= 0, partner % 3, -1).reshape(nT, 3)

        # ---- faces on the four sides of the subdomain (always local face 0 of a ring triangle)
        mid = 0.5 * (lat[:, [1, 2, 0], :] + lat[:, [2, 0, 1], :])          # face midpoints (lattice units)
        side = np.full((nT, 3), -1, dtype=np.int64)
        side[(partner.reshape(nT, 3) < 0) & (mid[:, :, 1] == 0)] = 0
        side[(partner.reshape(nT, 3) < 0) & (mid[:, :, 0] == 0)] = 1
        side[(partner.reshape(nT, 3) < 0) & (mid[:, :, 0] == 2 * kx)] = 2
        side[(partner.reshape(nT, 3) < 0) & (mid[:, :, 1] == 2 * ky)] = 3
        assert np.all((partner.reshape(nT, 3) >= 0) | (side >= 0))
        self.face_side = side.astype(np.int32)
        # pairing with the neighbouring subdomain: same face midpoint, shifted by one subdomain
        shift = {0: (0, 2 * ky), 1: (2 * kx, 0), 2: (-2 * kx, 0), 3: (0, -2 * ky)}
        opposite = {0: 3, 1: 2, 2: 1, 3: 0}
        self.ncf = 2 * max(kx, ky)
        self.side_elem = np.full((4, self.ncf), -1, dtype=np.int32)        # our element on that side, face idx
        self.side_face = np.full((4, self.ncf), -1, dtype=np.int32)
        self.side_elem_out = np.full((4, self.ncf), -1, dtype=np.int32)    # element in the neighbour
        self.side_face_out = np.full((4, self.ncf), -1, dtype=np.int32)
        self.side_count = np.zeros(4, dtype=np.int32)
        elem_side_pos = np.full((nT, 3), -1, dtype=np.int64)
        midkey = {}
        for sd in range(4):
            es, fs = np.nonzero(side == sd)
            self.side_count[sd] = len(es)
            for pos, (el, f) in enumerate(zip(es, fs)):
                self.side_elem[sd, pos], self.side_face[sd, pos] = el, f
                elem_side_pos[el, f] = pos
                midkey[(sd, float(mid[el, f, 0]), float(mid[el, f, 1]))] = (el, f)
        for sd in range(4):
            for pos in range(self.side_count[sd]):
                el, f = self.side_elem[sd, pos], self.side_face[sd, pos]
                mx, my = mid[el, f, 0] + shift[sd][0], mid[el, f, 1] + shift[sd][1]
                eo, fo = midkey[(opposite[sd], float(mx), float(my))]
                self.side_elem_out[sd, pos], self.side_face_out[sd, pos] = eo, fo
        # element-adjacency table used by the block-ELL matrices and the kernels:
        #   nb_elem >= 0: inner neighbour (local element);  < 0: -(1 + side)
        self.nb_elem = np.where(nb_elem >= 0, nb_elem, -(1 + side)).astype(np.int32)
        self.nb_face = np.where(nb_face >= 0, nb_face, 0).astype(np.int32)
        self.elem_side_pos = elem_side_pos.astype(np.int32)
        # for side faces: element / face in the neighbouring subdomain
        nb_out = np.full((nT, 3), -1, dtype=np.int64)
        nbf_out = np.full((nT, 3), -1, dtype=np.int64)
        for sd in range(4):
            c = self.side_count[sd]
            nb_out[self.side_elem[sd, :c], self.side_face[sd, :c]] = self.side_elem_out[sd, :c]
            nbf_out[self.side_elem[sd, :c], self.side_face[sd, :c]] = self.side_face_out[sd, :c]
        self.nb_elem_out = nb_out.astype(np.int32)
        self.nb_face_out = nbf_out.astype(np.int32)

        # ---- orientation sign of (element, face): +1 if the outward normal is the face normal.
        # Side faces S/W get -1 here; a kernel overrides it with +1 where the side is domain boundary.
        nx, ny = self.normal[:, :, 0], self.normal[:, :, 1]
        pos = (nx > 1e-12) | ((np.abs(nx) <= 1e-12) & (ny > 0))
        self.face_sign = np.where(pos, 1, -1).astype(np.int32)

        # ---- RT face -> (primary element, face, secondary element, face, side)
        rt_e0 = np.full(self.n_rt, -1, dtype=np.int64)
        rt_f0 = np.full(self.n_rt, -1, dtype=np.int64)
        rt_e1 = np.full(self.n_rt, -1, dtype=np.int64)
        rt_f1 = np.full(self.n_rt, -1, dtype=np.int64)
        rt_side = np.full(self.n_rt, -1, dtype=np.int64)
        for el in range(nT):
            for f in range(3):
                r = self.elem_rt[el, f]
                if rt_e0[r] < 0:
                    rt_e0[r], rt_f0[r] = el, f
                    if side[el, f] >= 0:
                        rt_side[r] = side[el, f]
                        rt_e1[r], rt_f1[r] = nb_out[el, f], nbf_out[el, f]
                else:
                    rt_e1[r], rt_f1[r] = el, f
        self.rt_e0, self.rt_f0 = rt_e0.astype(np.int32), rt_f0.astype(np.int32)
        self.rt_e1, self.rt_f1 = rt_e1.astype(np.int32), rt_f1.astype(np.int32)
        self.rt_side = rt_side.astype(np.int32)

        # ---- vertex stars (Oswald): CSR lattice vertex -> local DoFs, sorted by DoF
        order = np.argsort(self.dof_vertex, kind='stable')
        counts = np.bincount(self.dof_vertex, minlength=self.n_vertices)
        self.vdof_ptr = np.concatenate(([0], np.cumsum(counts))).astype(np.int32)
        self.vdof_idx = order.astype(np.int32)

        # ---- elements having at least one vertex on a side (support of the neighbours' Oswald images), padded table
        on = [lat[:, :, 1] == 0, lat[:, :, 0] == 0, lat[:, :, 0] == 2 * kx, lat[:, :, 1] == 2 * ky]
        lists = [np.nonzero(o.any(axis=1))[0] for o in on]
        self.ntouch = max(len(l) for l in lists)
        self.touch_elem = np.full((4, self.ntouch), -1, dtype=np.int32)
        self.touch_count = np.array([len(l) for l in lists], dtype=np.int32)
        for sd in range(4):
            self.touch_elem[sd, :len(lists[sd])] = lists[sd]


class DDSubdomainsGrid:
    """What ``make_cube_dd_subdomains_grid__*`` returns in the reference, for the structured case."""

    def __init__(self, lower_left, upper_right, num_elements, num_partitions,
                 inner_boundary_segment_index=INNER_BOUNDARY_SEGMENT_INDEX, rank=0, world_size=1):
        self.lower_left = np.asarray(lower_left, dtype=np.float64)
        self.upper_right = np.asarray(upper_right, dtype=np.float64)
        Kx, Ky = int(num_elements[0]), int(num_elements[1])
        Px, Py = int(num_partitions[0]), int(num_partitions[1])
        if Kx % Px or Ky % Py:
            raise ValueError('num_elements {} must be divisible by num_partitions {}'.format((Kx, Ky), (Px, Py)))
        self.K, self.P = (Kx, Ky), (Px, Py)
        self.inner_boundary_segment_index = inner_boundary_segment_index
        self.num_subdomains = Px * Py
        self.hx = (self.upper_right[0] - self.lower_left[0]) / (2 * Kx)
        self.hy = (self.upper_right[1] - self.lower_left[1]) / (2 * Ky)
        self.template = SubdomainTemplate(Kx // Px, Ky // Py, self.hx, self.hy)
        self.num_elements = 8 * Kx * Ky
        s = np.arange(self.num_subdomains)
        sx, sy = s % Px, s // Px
        nb = np.full((self.num_subdomains, 5), -1, dtype=np.int64)
        nb[:, 2] = s
        nb[sy > 0, 0] = (s - Px)[sy > 0]
        nb[sx > 0, 1] = (s - 1)[sx > 0]
        nb[sx < Px - 1, 3] = (s + 1)[sx < Px - 1]
        nb[sy < Py - 1, 4] = (s + Px)[sy < Py - 1]
        self.neighbor_slots = nb                      # [S, 5] global subdomain id per slot or -1
        self.rank, self.world_size = rank, world_size
        self._on_rank = self._partition(rank, world_size)

    # -- rank ownership: contiguous 2D tiles of subdomains
    def _partition(self, rank, world_size):
        Px, Py = self.P
        if world_size == 1:
            return list(range(self.num_subdomains))
        gx, gy = tile_grid(world_size, Px, Py)
        rx, ry = rank % gx, rank // gx
        xs = np.array_split(np.arange(Px), gx)[rx]
        ys = np.array_split(np.arange(Py), gy)[ry]
        return [int(x + Px * y) for y in ys for x in xs]

    @property
    def subdomains_on_rank(self):
        return list(self._on_rank)

    def neighboring_subdomains(self, ii):
        r = self.neighbor_slots[ii]
        return [int(j) for k, j in enumerate(r) if j >= 0 and k != 2]

    def neighborhood_of(self, ii):
        return [int(j) for j in self.neighbor_slots[ii] if j >= 0]

    def diagonal_neighbors(self, ii):
        """Global ids of the subdomains that touch ``ii`` in ONE vertex, by corner 0 SW, 1 SE, 2 NW, 3 NE (-1: none).  Not part of
        ``neighborhood_of`` (HEAD's grid.neighborhood_of returns face neighbours); the Oswald vertex patch reads them
        (conventions={'oswald_vertex_patch': True})."""
        Px, Py = self.P
        sx, sy = int(ii) % Px, int(ii) // Px
        out = []
        for dy in (-1, 1):
            for dx in (-1, 1):
                x, y = sx + dx, sy + dy
                out.append(int(x + Px * y) if 0 <= x < Px and 0 <= y < Py else -1)
        return out

    def halo_subdomains(self, diagonal=False):
        """Sorted global ids of the subdomains other ranks own whose basis rows this rank's subdomains read: the face neighbours
        and, with ``diagonal`` (Oswald vertex patch), the diagonal ones.  Engine and HaloPlan order the halo slabs this way."""
        local = set(self.subdomains_on_rank)
        halo = {j for s in local for j in self.neighboring_subdomains(s)}
        if diagonal:
            halo |= {j for s in local for j in self.diagonal_neighbors(s) if j >= 0}
        return sorted(int(j) for j in halo - local)

    def boundary_subdomains(self):
        return [int(i) for i in np.nonzero((self.neighbor_slots < 0).any(axis=1))[0]]

    def subdomain_origin(self, ii):
        Px = self.P[0]
        sx, sy = ii % Px, ii // Px
        t = self.template
        return self.lower_left + np.array([sx * 2 * t.kx * self.hx, sy * 2 * t.ky * self.hy])

    def subdomain_diameter(self, ii):
        t = self.template
        return float(np.hypot(2 * t.kx * self.hx, 2 * t.ky * self.hy))

    def max_entity_diameter(self):
        return float(max(2 * self.hx, 2 * self.hy, np.hypot(self.hx, self.hy)))

    def element_keys(self, subdomains):
        """Canonical (global coarse cx, cy, t) key of every element of the given subdomains [len, n_T, 3]."""
        t = self.template
        Px = self.P[0]
        s = np.asarray(subdomains, dtype=np.int64)
        key = np.broadcast_to(t.elem_square[None], (len(s), t.n_T, 3)).copy()
        key[:, :, 0] += ((s % Px) * t.kx)[:, None]
        key[:, :, 1] += ((s // Px) * t.ky)[:, None]
        return key

    def visualize(self, name, with_coupling=False):
        """grid.py:35 ``grid.visualize('grid', False)``: the mesh with the subdomain index as cell data (legacy VTK)."""
        from pylrbms_amd.visualize import write_vtk
        return write_vtk(name, self, range(self.num_subdomains))


def tile_grid(world_size, Px, Py):
    """Factor ``world_size`` into gx * gy process tiles, as square as the subdomain grid allows."""
    best = None
    for gx in range(1, world_size + 1):
        if world_size % gx:
            continue
        gy = world_size // gx
        if gx > Px or gy > Py:
            continue
        per = (Px / gx) + (Py / gy)                   # perimeter proxy
        if best is None or per < best[0]:
            best = (per, gx, gy)
    if best is None:
        raise ValueError('cannot tile {}x{} subdomains over {} ranks'.format(Px, Py, world_size))
    return best[1], best[2]


def make_grid(domain=([0, 0], [1, 1]), num_subdomains=None, half_num_fine_elements_per_subdomain_and_dim=4,
              inner_boundary_segment_index=INNER_BOUNDARY_SEGMENT_INDEX, mpi_comm=None):
    """Reference grid.py:8-42.  As there, ``half_num_fine_elements_per_subdomain_and_dim`` is handed on as the
    GLOBAL number of coarse squares per direction (grid.py:24-25).  ``mpi_comm`` may be ``None`` (serial) or any
    object with ``rank`` / ``size`` attributes (e.g. pylrbms_amd.parallel.Communicator)."""
    assert half_num_fine_elements_per_subdomain_and_dim > 3
    h = half_num_fine_elements_per_subdomain_and_dim
    if not num_subdomains:
        num_subdomains = [1, 1]
    rank = getattr(mpi_comm, 'rank', 0) if mpi_comm is not None else 0
    size = getattr(mpi_comm, 'size', 1) if mpi_comm is not None else 1
    return DDSubdomainsGrid(domain[0], domain[1], [h, h], num_subdomains, inner_boundary_segment_index, rank, size)


def make_multiscale_grid(num_subdomains, coarse_per_subdomain=4, domain=([0, 0], [1, 1]), mpi_comm=None):
    """Synthetic benchmark grid of SURVEY section 8(d): ``k_c x k_c`` coarse squares per subdomain."""
    Px, Py = num_subdomains
    rank = getattr(mpi_comm, 'rank', 0) if mpi_comm is not None else 0
    size = getattr(mpi_comm, 'size', 1) if mpi_comm is not None else 1
    return DDSubdomainsGrid(domain[0], domain[1], [Px * coarse_per_subdomain, Py * coarse_per_subdomain],
                            [Px, Py], rank=rank, world_size=size)


def make_boundary_info(grid, config):
    """Reference grid.py:45-53.  Only the all-Dirichlet boundary info used by every problem file is supported."""
    if config.get('type') != 'xt.grid.boundaryinfo.alldirichlet':
        raise NotImplementedError(config)
    return dict(config)


def grid_info(log, grid, mpi_comm=None):
    """Reference grid.py:56-69 (banner; the allreduce there only sums an int)."""
    tpl = ('\n**************************************************************\n* Grid Type {}\n* # Subdomains {}\n'
           '* Process subdomains {}\n* First Neighbors {}\n* Boundary Subdomains {}\n'
           '**************************************************************\n')
    log(tpl.format(type(grid).__name__, grid.num_subdomains, grid.subdomains_on_rank,
                   grid.neighboring_subdomains(grid.subdomains_on_rank[0]), grid.boundary_subdomains()))
