"""Tetrahedral subdomain mesh of the 3D / P2 oracle (BASELINE.json config 5) -- TEST INFRASTRUCTURE ONLY.

The 3D analogue of oracle/mesh.py: the unit-cube-type domain is cut into ``K`` cubes per direction, every cube into the
six tetrahedra of the Kuhn (Freudenthal) triangulation -- translation invariant, hence conforming across cubes -- and
the cubes into ``P`` subdomains per direction (an element belongs to the subdomain its centre falls into, as
dune-xt-grid's ``dd_subdomains_cube`` does in 2D).  Elements are numbered subdomain-major, so the
block DG mapper is ``dof = ndof_local * element + local`` with ``element = ii * n_T + e_local``.

The reference binds the 2D P1 operators only (discretize_elliptic_block_swipdg.py:22-23, ``x[0], x[1]`` at :195): there
is NO reference counterpart for this mesh; its conventions are the dimension-independent ones of SURVEY.md App. A.
"""
import itertools

import numpy as np

# local faces: face f is opposite vertex f
FACE_VERTS = np.array([[1, 2, 3], [0, 2, 3], [0, 1, 3], [0, 1, 2]])
# local edges of the P2 element, in the order of the edge DoFs 4 .. 9
EDGE_VERTS = np.array([[0, 1], [0, 2], [0, 3], [1, 2], [1, 3], [2, 3]])


class KuhnMesh3D:
    def __init__(self, num_cubes, num_subdomains, lower_left=(0.0, 0.0, 0.0), upper_right=(1.0, 1.0, 1.0)):
        K = np.asarray(num_cubes, dtype=np.int64)
        P = np.asarray(num_subdomains, dtype=np.int64)
        assert K.shape == (3,) and P.shape == (3,) and np.all(K % P == 0)
        self.K, self.P = K, P
        self.lower_left = np.asarray(lower_left, dtype=np.float64)
        self.upper_right = np.asarray(upper_right, dtype=np.float64)
        kc = K // P
        self.kc = kc
        nvx = K + 1
        ii, jj, kk = np.meshgrid(np.arange(nvx[0]), np.arange(nvx[1]), np.arange(nvx[2]), indexing='ij')
        h = (self.upper_right - self.lower_left) / K
        self.vertices = np.stack([ii, jj, kk], axis=-1).reshape(-1, 3) * h + self.lower_left

        def vid(i, j, k):
            return (i * nvx[1] + j) * nvx[2] + k

        # elements, subdomain-major (x fastest subdomain numbering as in 2D: s = sx + Px (sy + Py sz))
        perms = list(itertools.permutations(range(3)))
        tets, sub = [], []
        for sz in range(P[2]):
            for sy in range(P[1]):
                for sx in range(P[0]):
                    s = sx + P[0] * (sy + P[1] * sz)
                    for cz in range(kc[2]):
                        for cy in range(kc[1]):
                            for cx in range(kc[0]):
                                base = np.array([sx * kc[0] + cx, sy * kc[1] + cy, sz * kc[2] + cz])
                                for perm in perms:
                                    p = base.copy()
                                    verts = [vid(*p)]
                                    for d in perm:
                                        p = p.copy()
                                        p[d] += 1
                                        verts.append(vid(*p))
                                    tets.append(verts)
                                    sub.append(s)
        tets = np.array(tets, dtype=np.int64)
        # positive orientation
        X = self.vertices[tets]
        vol6 = np.linalg.det(X[:, 1:] - X[:, :1])
        neg = vol6 < 0
        tets[neg] = tets[neg][:, [0, 2, 1, 3]]
        self.elements = tets
        self.elem_subdomain = np.array(sub, dtype=np.int64)
        self.num_elements = len(tets)
        self.num_subdomains = int(P.prod())
        self.elements_per_subdomain = self.num_elements // self.num_subdomains
        assert np.all(self.elem_subdomain == np.repeat(np.arange(self.num_subdomains), self.elements_per_subdomain))
        self._geometry()
        self._faces()
        self._nodes()
        self._subdomain_graph()

    # ------------------------------------------------------------------ geometry
    def _geometry(self):
        X = self.vertices[self.elements]                         # [nT, 4, 3]
        J = X[:, 1:] - X[:, :1]                                  # rows: edge vectors
        self.volume = np.linalg.det(J) / 6.0
        assert np.all(self.volume > 0)
        Jinv = np.linalg.inv(J)                                  # x - x0 = l[1:] @ J  ->  l[1:] = (x - x0) @ Jinv
        g = np.empty((self.num_elements, 4, 3))
        g[:, 1:] = np.transpose(Jinv, (0, 2, 1))                 # grad lambda_i = column i-1 of Jinv
        g[:, 0] = -g[:, 1:].sum(axis=1)
        self.grad_lambda = g
        self.centers = X.mean(axis=1)
        self.diameter = np.max(np.linalg.norm(X[:, :, None] - X[:, None, :], axis=-1), axis=(1, 2))

    def barycentric(self, e, x):
        """Barycentric coordinates of the points ``x [..., 3]`` in element(s) ``e`` (broadcast over the leading axes)."""
        x0 = self.vertices[self.elements[e, 0]]
        g = self.grad_lambda[e]                                  # [..., 4, 3]
        lam = np.einsum('...ia,...a->...i', g, x - x0)
        lam[..., 0] += 1.0
        return lam

    # ------------------------------------------------------------------ faces
    def _faces(self):
        nT = self.num_elements
        keys = np.sort(self.elements[:, FACE_VERTS], axis=2).reshape(nT * 4, 3)
        uniq, inv, counts = np.unique(keys, axis=0, return_inverse=True, return_counts=True)
        inv = inv.reshape(-1)
        self.elem_face = inv.reshape(nT, 4)
        nF = len(uniq)
        self.num_faces = nF
        self.face_vertices = uniq
        minus = np.full((nF, 2), -1, dtype=np.int64)
        plus = np.full((nF, 2), -1, dtype=np.int64)
        for e in range(nT):                                      # increasing element index: the lower element is "minus"
            for f in range(4):
                fid = self.elem_face[e, f]
                if minus[fid, 0] < 0:
                    minus[fid] = (e, f)
                else:
                    plus[fid] = (e, f)
        self.face_minus, self.face_plus = minus, plus
        V = self.vertices[uniq]
        nrm = np.cross(V[:, 1] - V[:, 0], V[:, 2] - V[:, 0])
        self.face_area = 0.5 * np.linalg.norm(nrm, axis=1)
        nrm /= np.linalg.norm(nrm, axis=1)[:, None]
        # outward from the minus element: pointing away from its vertex opposite the face
        opp = self.vertices[self.elements[minus[:, 0], minus[:, 1]]]
        flip = np.einsum('fa,fa->f', nrm, V[:, 0] - opp) < 0
        nrm[flip] *= -1.0
        self.face_normal = nrm
        self.face_is_boundary = plus[:, 0] < 0
        # orientation sign of (element, local face): +1 for the minus side
        sign = -np.ones((nT, 4))
        sign[minus[:, 0], minus[:, 1]] = 1.0
        self.elem_face_sign = sign
        sm = self.elem_subdomain[minus[:, 0]]
        sp_ = np.where(plus[:, 0] >= 0, self.elem_subdomain[np.maximum(plus[:, 0], 0)], -1)
        self.face_sub_minus, self.face_sub_plus = sm, sp_
        self.face_is_coupling = (~self.face_is_boundary) & (sm != sp_)

    # ------------------------------------------------------------------ P2 Lagrange nodes
    def _nodes(self):
        nT, nv = self.num_elements, len(self.vertices)
        ekeys = np.sort(self.elements[:, EDGE_VERTS], axis=2).reshape(nT * 6, 2)
        uniq, inv = np.unique(ekeys, axis=0, return_inverse=True)
        self.edges = uniq
        self.elem_nodes = np.concatenate([self.elements, nv + inv.reshape(nT, 6)], axis=1)    # [nT, 10]
        self.num_nodes = nv + len(uniq)
        self.node_coords = np.concatenate([self.vertices, self.vertices[uniq].mean(axis=1)], axis=0)
        lo, hi = self.lower_left, self.upper_right
        tol = 1e-12 * np.max(hi - lo)
        self.node_on_boundary = np.any((np.abs(self.node_coords - lo) < tol) | (np.abs(self.node_coords - hi) < tol), axis=1)

    # ------------------------------------------------------------------ subdomains
    def _subdomain_graph(self):
        nb = [set() for _ in range(self.num_subdomains)]
        for a, b in zip(self.face_sub_minus[self.face_is_coupling], self.face_sub_plus[self.face_is_coupling]):
            nb[a].add(int(b))
            nb[b].add(int(a))
        self._neighbors = [sorted(x) for x in nb]
        lo = self.lower_left
        hs = (self.upper_right - lo) / self.P
        self.subdomain_diameter = float(np.linalg.norm(hs))

    def neighboring_subdomains(self, ii):
        return list(self._neighbors[ii])

    def neighborhood_of(self, ii):
        return sorted([ii] + self._neighbors[ii])

    def elements_of(self, ii):
        n = self.elements_per_subdomain
        return np.arange(ii * n, (ii + 1) * n)
