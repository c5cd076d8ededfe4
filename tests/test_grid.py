"""Index maps: structured closed forms of pylrbms_amd.grid vs the generic,
hash-based oracle mesh -- bit-exact."""
import numpy as np
import pytest

from oracle.mesh import OracleMesh
from pylrbms_amd.grid import DDSubdomainsGrid, make_grid, tile_grid, SIDE_TO_SLOT

CASES = [((4, 4), (2, 2)), ((8, 8), (4, 4)), ((12, 8), (3, 2)), ((4, 4), (1, 1)), ((4, 4), (4, 4)), ((6, 2), (3, 1))]


@pytest.mark.parametrize('K,P', CASES)
def test_index_maps_bit_exact(K, P):
    ll, ur = [-1.0, -1.0], [1.0, 1.0]
    m = OracleMesh(ll, ur, K, P)
    g = DDSubdomainsGrid(ll, ur, K, P)
    t = g.template
    assert g.num_subdomains == m.num_subdomains
    assert t.n_T == m.elements_per_subdomain
    nvx = m.lattice_shape[0]
    for ii in range(g.num_subdomains):
        assert g.neighborhood_of(ii) == m.neighborhood_of(ii)
        assert g.neighboring_subdomains(ii) == m.neighboring_subdomains(ii)
        E0 = m.elem_offset[ii]
        sl = slice(E0, E0 + t.n_T)
        # element keys and vertices (global lattice ids)
        assert np.array_equal(g.element_keys([ii])[0], m.elem_key[sl])
        sx, sy = ii % P[0], ii // P[0]
        glat = t.tri_lattice + np.array([2 * t.kx * sx, 2 * t.ky * sy])
        gid = glat[:, :, 0] + nvx * glat[:, :, 1]
        assert np.array_equal(gid, m.triangles[sl])
        # per-subdomain RT numbering
        loc = np.array([[m.rt_local[ii][int(f)] for f in m.elem_face[E]] for E in range(E0, E0 + t.n_T)])
        assert np.array_equal(loc, t.elem_rt)
        assert len(m.rt_faces[ii]) == t.n_rt
        # element adjacency / coupling pairing / orientation
        for el in range(t.n_T):
            for f in range(3):
                gf = m.elem_face[E0 + el, f]
                other = m.face_plus[gf] if m.face_minus[gf, 0] == E0 + el else m.face_minus[gf]
                kind = m.face_kind[gf]
                if kind == 0:
                    assert t.nb_elem[el, f] == other[0] - E0 and t.nb_face[el, f] == other[1]
                    assert t.face_sign[el, f] == m.elem_face_sign[E0 + el, f]
                else:
                    side = -(t.nb_elem[el, f]) - 1
                    assert 0 <= side < 4
                    jj = g.neighbor_slots[ii, SIDE_TO_SLOT[side]]
                    if kind == 2:
                        assert jj < 0 and m.elem_face_sign[E0 + el, f] == 1
                    else:
                        assert jj == m.elem_subdomain[other[0]]
                        assert t.nb_elem_out[el, f] == m.elem_local[other[0]]
                        assert t.nb_face_out[el, f] == other[1]
                        assert t.face_sign[el, f] == m.elem_face_sign[E0 + el, f]
    assert g.boundary_subdomains() == m.boundary_subdomains()
    # geometry
    assert np.allclose(t.area, m.area[:t.n_T], rtol=0, atol=1e-15)
    assert np.allclose(t.grad, m.grads[:t.n_T], rtol=1e-14, atol=0)
    assert abs(g.subdomain_diameter(0) - m.subdomain_diameter(0)) < 1e-14


def test_vertex_stars():
    g = DDSubdomainsGrid([0, 0], [1, 1], (8, 8), (2, 2))
    t = g.template
    for v in range(t.n_vertices):
        dofs = t.vdof_idx[t.vdof_ptr[v]:t.vdof_ptr[v + 1]]
        assert np.all(t.dof_vertex[dofs] == v)
    assert t.vdof_ptr[-1] == t.n


def test_make_grid_matches_reference_call():
    g = make_grid(([-1, -1], [1, 1]), [2, 2], 4)
    assert g.num_subdomains == 4 and g.template.n == 96
    with pytest.raises(AssertionError):
        make_grid(([-1, -1], [1, 1]), [2, 2], 3)             # grid.py:13


def test_rank_tiles_partition_all_subdomains():
    for ws in (1, 2, 4, 8):
        owned = []
        for r in range(ws):
            g = DDSubdomainsGrid([0, 0], [1, 1], (32, 32), (32, 32), rank=r, world_size=ws)
            owned += g.subdomains_on_rank
        assert sorted(owned) == list(range(1024))
    assert tile_grid(8, 32, 32) in ((2, 4), (4, 2))
