"""Oracle grid + index maps (row K0 of SURVEY.md section 8a) -- TEST INFRASTRUCTURE ONLY.

Restates what the reference obtains from dune-xt-grid through
``make_grid`` (python/dune/pylrbms/grid.py:8-42): a cube grid with
``num_elements`` coarse squares, two conforming refinements (8 triangles per
coarse square), a Cartesian partition into ``num_partitions`` subdomains, and
the queries used by the hot path: ``neighborhood_of`` / ``neighboring_subdomains``
/ ``boundary_subdomains`` (discretize_elliptic_block_swipdg.py:78,393,421) and
the block DG mapper (``offset[ii] + local``, :115-116, :304-305).

Everything here is derived *generically* (hashing edges, bucketing element
centres, python loops) on purpose: the product computes the same maps with
closed-form structured-grid formulas and the tests compare both bit for bit.

PARITY UNPINNED: dune-xt-grid is absent; element / face numbering inside a
subdomain is this oracle's documented convention.
"""
import numpy as np

# counter-clockwise ring of the 8 boundary lattice points of a coarse square
# (lattice spacing = half a coarse square), starting at the lower-left corner
RING = np.array([(0, 0), (1, 0), (2, 0), (2, 1), (2, 2), (1, 2), (0, 2), (0, 1)], dtype=np.int64)


class OracleMesh:
    """Triangulated cube with a Cartesian subdomain partition.

    Parameters mirror ``make_cube_dd_subdomains_grid`` as called at grid.py:21-30:
    ``num_elements`` is the GLOBAL number of coarse squares per direction.
    """

    def __init__(self, lower_left, upper_right, num_elements, num_partitions):
        ll = np.asarray(lower_left, dtype=np.float64)
        ur = np.asarray(upper_right, dtype=np.float64)
        Kx, Ky = int(num_elements[0]), int(num_elements[1])
        Px, Py = int(num_partitions[0]), int(num_partitions[1])
        self.ll, self.ur, self.K, self.P = ll, ur, (Kx, Ky), (Px, Py)
        self.num_subdomains = Px * Py
        nvx, nvy = 2 * Kx + 1, 2 * Ky + 1
        self.lattice_shape = (nvx, nvy)
        hx = (ur[0] - ll[0]) / (2 * Kx)
        hy = (ur[1] - ll[1]) / (2 * Ky)
        self.h = (hx, hy)

        # ---- vertices: every lattice point, id = j * nvx + i
        jj, ii = np.meshgrid(np.arange(nvy), np.arange(nvx), indexing='ij')
        self.vertex_lattice = np.stack([ii.ravel(), jj.ravel()], axis=1).astype(np.int64)
        self.vertices = ll[None, :] + self.vertex_lattice * np.array([hx, hy])[None, :]

        # ---- elements in generation order: coarse squares row-major, 8 triangles each
        tris = []
        keys = []
        for cy in range(Ky):
            for cx in range(Kx):
                c = (2 * cx + 1) + nvx * (2 * cy + 1)
                for t in range(8):
                    a = RING[t]
                    b = RING[(t + 1) % 8]
                    va = (2 * cx + a[0]) + nvx * (2 * cy + a[1])
                    vb = (2 * cx + b[0]) + nvx * (2 * cy + b[1])
                    tris.append((c, va, vb))
                    keys.append((cx, cy, t))
        gen_tris = np.asarray(tris, dtype=np.int64)
        gen_keys = np.asarray(keys, dtype=np.int64)
        nE = gen_tris.shape[0]

        # ---- subdomain of an element: bucket of its centre
        centers = self.vertices[gen_tris].mean(axis=1)
        sub = np.zeros(nE, dtype=np.int64)
        stride = 1
        for d, Pd in enumerate((Px, Py)):
            b = np.floor(Pd * (centers[:, d] - ll[d]) / (ur[d] - ll[d])).astype(np.int64)
            b = np.minimum(b, Pd - 1)
            sub += b * stride
            stride *= Pd

        # ---- block ordering: subdomain-major, inside a subdomain the generation order
        order = np.argsort(sub, kind='stable')
        self.triangles = gen_tris[order]          # [nE, 3] vertex ids, CCW
        self.elem_key = gen_keys[order]           # [nE, 3] (cx, cy, t) canonical key
        self.elem_subdomain = sub[order]          # [nE]
        counts = np.bincount(sub, minlength=self.num_subdomains)
        assert np.all(counts == counts[0]), 'subdomains must hold equally many elements'
        self.elements_per_subdomain = int(counts[0])
        self.elem_offset = np.concatenate(([0], np.cumsum(counts))).astype(np.int64)
        self.elem_local = np.arange(nE, dtype=np.int64) - self.elem_offset[self.elem_subdomain]
        self.num_elements = nE
        self.elem_center = self.vertices[self.triangles].mean(axis=1)
        self.dof_offset = 3 * self.elem_offset      # block DG mapper offsets

        # ---- geometry
        p = self.vertices[self.triangles]           # [nE, 3, 2]
        e1 = p[:, 1] - p[:, 0]
        e2 = p[:, 2] - p[:, 0]
        det = e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]
        assert np.all(det > 0), 'triangles must be counter-clockwise'
        self.area = 0.5 * det
        # grad phi_i = rot90(p_{i+2} - p_{i+1}) / (2|T|)  (inward normal of opposite edge)
        grads = np.zeros((nE, 3, 2))
        for i in range(3):
            d = p[:, (i + 2) % 3] - p[:, (i + 1) % 3]
            grads[:, i, 0] = -d[:, 1] / det
            grads[:, i, 1] = d[:, 0] / det
        self.grads = grads
        self.points = p

        # ---- faces: hash sorted vertex pairs; local face f is opposite local vertex f
        face_of = {}
        face_list = []
        for E in range(nE):
            tri = self.triangles[E]
            for f in range(3):
                a, b = int(tri[(f + 1) % 3]), int(tri[(f + 2) % 3])
                key = (a, b) if a < b else (b, a)
                idx = face_of.get(key)
                if idx is None:
                    face_of[key] = len(face_list)
                    face_list.append([(E, f)])
                else:
                    face_list[idx].append((E, f))
        nF = len(face_list)
        self.num_faces = nF
        f_minus = np.zeros((nF, 2), dtype=np.int64)
        f_plus = np.full((nF, 2), -1, dtype=np.int64)
        normals = np.zeros((nF, 2))
        lengths = np.zeros(nF)
        elem_face = np.full((nE, 3), -1, dtype=np.int64)
        elem_face_sign = np.zeros((nE, 3), dtype=np.int64)
        for idx, sides in enumerate(face_list):
            (E0, f0) = sides[0]
            g = grads[E0, f0]
            n0 = -g / np.linalg.norm(g)             # outward normal of E0 on that face
            a = self.points[E0, (f0 + 1) % 3]
            b = self.points[E0, (f0 + 2) % 3]
            lengths[idx] = np.linalg.norm(b - a)
            # orientation rule: the face normal is the lexicographically positive one
            positive = (n0[0] > 1e-12) or (abs(n0[0]) <= 1e-12 and n0[1] > 0)
            if len(sides) == 1:
                f_minus[idx] = (E0, f0)
                normals[idx] = n0                   # domain boundary: outward
                elem_face[E0, f0] = idx
                elem_face_sign[E0, f0] = 1
            else:
                (E1, f1) = sides[1]
                if positive:
                    f_minus[idx], f_plus[idx], normals[idx] = (E0, f0), (E1, f1), n0
                else:
                    f_minus[idx], f_plus[idx], normals[idx] = (E1, f1), (E0, f0), -n0
                Em, fm = f_minus[idx]
                Ep, fp = f_plus[idx]
                elem_face[Em, fm] = idx
                elem_face_sign[Em, fm] = 1
                elem_face[Ep, fp] = idx
                elem_face_sign[Ep, fp] = -1
        self.face_minus, self.face_plus = f_minus, f_plus
        self.face_normal, self.face_length = normals, lengths
        self.elem_face, self.elem_face_sign = elem_face, elem_face_sign
        sm = self.elem_subdomain[f_minus[:, 0]]
        sp = np.where(f_plus[:, 0] >= 0, self.elem_subdomain[np.maximum(f_plus[:, 0], 0)], -1)
        self.face_kind = np.where(f_plus[:, 0] < 0, 2, np.where(sm == sp, 0, 1))  # 0 inner 1 coupling 2 boundary
        self.face_sub_minus, self.face_sub_plus = sm, sp

        # ---- subdomain neighbour relations (share a codim-1 intersection)
        nbrs = [set() for _ in range(self.num_subdomains)]
        for idx in np.nonzero(self.face_kind == 1)[0]:
            nbrs[sm[idx]].add(int(sp[idx]))
            nbrs[sp[idx]].add(int(sm[idx]))
        self._neighbors = [sorted(s) for s in nbrs]
        self._boundary_subdomains = sorted(set(int(s) for s in sm[self.face_kind == 2]))

        # ---- per-subdomain RT0 (face) numbering: first appearance over (element, local face)
        self.rt_faces = []      # list of arrays: global face ids in local RT order
        self.rt_local = []      # dict global face -> local RT index
        nT = self.elements_per_subdomain
        for s in range(self.num_subdomains):
            seen = {}
            lst = []
            for E in range(self.elem_offset[s], self.elem_offset[s] + nT):
                for f in range(3):
                    gf = int(elem_face[E, f])
                    if gf not in seen:
                        seen[gf] = len(lst)
                        lst.append(gf)
            self.rt_faces.append(np.asarray(lst, dtype=np.int64))
            self.rt_local.append(seen)

        # ---- vertex -> (element, local vertex) adjacency (Oswald)
        adj = [[] for _ in range(self.vertices.shape[0])]
        for E in range(nE):
            for v in range(3):
                adj[int(self.triangles[E, v])].append((E, v))
        self.vertex_adjacency = adj
        lat = self.vertex_lattice
        self.vertex_on_boundary = ((lat[:, 0] == 0) | (lat[:, 0] == nvx - 1) |
                                   (lat[:, 1] == 0) | (lat[:, 1] == nvy - 1))

    # queries named as the reference uses them
    def neighboring_subdomains(self, ii):
        return list(self._neighbors[ii])

    def neighborhood_of(self, ii):
        return sorted([ii] + self._neighbors[ii])

    def vertex_neighborhood_of(self, ii):
        """ii and every subdomain sharing at least a vertex with it (face neighbours + diagonal ones), sorted."""
        cache = self.__dict__.setdefault('_vertex_hoods', {})
        if ii not in cache:
            E0 = self.elem_offset[ii]
            verts = np.unique(self.triangles[E0:E0 + self.elements_per_subdomain])
            cache[ii] = sorted({int(self.elem_subdomain[E]) for g in verts for (E, _) in self.vertex_adjacency[int(g)]})
        return list(cache[ii])

    def boundary_subdomains(self):
        return list(self._boundary_subdomains)

    @property
    def subdomains_on_rank(self):
        return list(range(self.num_subdomains))

    def local_size(self, ii):
        return 3 * self.elements_per_subdomain

    def subdomain_diameter(self, ii):
        E0 = self.elem_offset[ii]
        pts = self.points[E0:E0 + self.elements_per_subdomain].reshape(-1, 2)
        lo, hi = pts.min(axis=0), pts.max(axis=0)
        return float(np.linalg.norm(hi - lo))
