"""Host logic of the 3D / P2 path (BASELINE.json config 5) on the CPU: the subdomain template, the quadrature rules and the
reference-tetrahedron tables of pylrbms_amd/grid3d.py against the generic (hash-based, global sparse) oracle.  The tables are
contracted with the coefficient samples in NumPy here -- exactly the arithmetic the HIP assembly kernels perform -- so the
conventions (element order, face point order, RT0 orientation, penalty scaling) are checked without a GPU."""
import numpy as np
import pytest

import common3d as c3
from oracle import lrbms3d as o3
from pylrbms_amd import grid3d as g3


def test_rules_match_the_oracle():
    for deg in range(1, 11):
        for mine, ref in ((g3.tet_rule(deg), o3.tet_rule(deg)), (g3.tri_rule(deg), o3.tri_rule(deg))):
            assert np.abs(mine[0] - ref[0]).max() < 1e-14 and np.abs(mine[1] - ref[1]).max() < 1e-14


def test_template_index_maps_are_consistent():
    t = g3.make_grid3d(num_subdomains=(1, 1, 1), cubes_per_subdomain_and_dim=4).template
    assert (t.n_T, t.n, t.n_rt, t.ncf, t.nvs, t.nb) == (384, 3840, 864, 32, 81, 386)          # config 5
    assert t.n_rt == (4 * t.n_T - 6 * t.ncf) // 2 + 6 * t.ncf
    # inner neighbours are mutual, across the same RT0 face, with opposite orientation
    for e in range(t.n_T):
        for f in range(4):
            nb = t.nb_elem[e, f]
            if nb >= 0:
                f2 = t.nb_face[e, f]
                assert t.nb_elem[nb, f2] == e and t.elem_rt[nb, f2] == t.elem_rt[e, f] and t.tsign[nb, f2] == -t.tsign[e, f]
    # a side face meets the mirrored face of the neighbour: opposite sides list each other's elements
    for a in range(6):
        assert sorted(t.side_elem[a]) == sorted(t.side_elem_out[5 - a])
    # every node's patch: own elements + the face neighbours' elements at it
    assert t.node_count.max() == 24 and t.node_count[t.node_mask == 0].min() >= 4
    assert len(t.sn_dofs) == t.sn_ptr[-1]
    t2 = g3.make_grid3d(num_subdomains=(3, 2, 2), cubes_per_subdomain_and_dim=(2, 1, 1)).template
    assert t2.ncf == 4 and sorted(t2.side_count) == [2, 2, 4, 4, 4, 4]


def test_grid_queries_match_the_oracle_mesh():
    from oracle.mesh3d import KuhnMesh3D
    g = g3.make_grid3d(num_subdomains=(3, 3, 3), cubes_per_subdomain_and_dim=1)
    m = KuhnMesh3D([3, 3, 3], [3, 3, 3])
    for ii in range(27):
        assert g.neighborhood_of(ii) == m.neighborhood_of(ii)
        assert g.neighboring_subdomains(ii) == m.neighboring_subdomains(ii)
    assert g.neighborhood_of(13) == [4, 10, 12, 13, 14, 16, 22]
    assert abs(g.subdomain_diameter() - m.subdomain_diameter) < 1e-15
    assert sorted(g3.make_grid3d(num_subdomains=(8, 8, 8), cubes_per_subdomain_and_dim=1, rank=3, world_size=8).subdomains_on_rank)[:3] \
        == [36, 37, 38]
    parts = [g3.DDSubdomainsGrid3D([0] * 3, [1] * 3, [8] * 3, [8] * 3, rank=r, world_size=8).subdomains_on_rank for r in range(8)]
    assert sorted(sum(parts, [])) == list(range(512)) and all(len(p) == 64 for p in parts)


@pytest.mark.parametrize('name', ['aniso_2x2x1', 'q3_2x1x2'])
def test_table_contraction_reproduces_the_oracle_assembly(name):
    p = c3.make_problem(name)
    d = c3.oracle_of(p)
    ref = c3.oracle_assembled(p, d)
    grid, t = p['grid'], p['grid'].template
    spec = g3.QuadratureSpec3D(2)
    T = t.tables(spec)
    xl, xh, xb = t.record_points(spec)
    Q, nT = d.Q, t.n_T
    worst = {}

    def upd(k, got, want):
        worst[k] = max(worst.get(k, 0.0), c3.rel(got, want))
    for s in range(grid.num_subdomains):
        org, phys = grid.subdomain_origin(s), int(grid.phys_mask[s])
        lams = [fn(xl + org) for fn in p['lambdas']]
        lh, fs, lbs = p['lambda_hat'](xh + org), p['f'](xh + org), p['lambda_bar'](xb + org)
        sg = np.where((t.nb_elem < 0) & (((phys >> np.maximum(-(t.nb_elem + 1), 0)) & 1) == 1), 1, t.tsign)     # [nT, 4]
        th_bar = c3.theta_of(p, p['mu_bar'])
        for e in range(nT):
            ty = t.elem_type[e]
            # local energy product (block_swipdg.py:651-677): lambda at mu_bar, penalty parts of the face tables, every face of the
            # subdomain boundary a Dirichlet face, no coupling
            lamb = sum(th_bar[q] * lams[q][e] for q in range(Q))
            pblk = lamb[:spec.nA] @ T['TV'][ty]
            for f in range(4):
                lf = lamb[spec.o_fs + f * spec.nFs:spec.o_fs + (f + 1) * spec.nFs]
                nb = t.nb_elem[e, f]
                pblk = pblk + lf @ (T['TPb'] if nb < 0 else T['TPo'])[ty, f]
                upd('P_nb', (lf @ T['TPn'][ty, f]).reshape(10, 10) if nb >= 0 else np.zeros((10, 10)), ref['P_diag'][s, e, 1 + f])
            upd('P_diag', pblk.reshape(10, 10), ref['P_diag'][s, e, 0])
            for q in range(Q):
                lam = lams[q][e]
                blk = lam[:spec.nA] @ T['TV'][ty]
                for f in range(4):
                    lf = lam[spec.o_fs + f * spec.nFs:spec.o_fs + (f + 1) * spec.nFs]
                    lc = lam[spec.o_ff + f * spec.nFf:spec.o_ff + (f + 1) * spec.nFf]
                    nb = t.nb_elem[e, f]
                    bnd = nb < 0 and (phys >> (-(nb + 1))) & 1
                    blk = blk + lf @ (T['TFb'] if bnd else T['TFo'])[ty, f]
                    if nb >= 0:
                        upd('A_nb', (lf @ T['TFn'][ty, f]).reshape(10, 10), ref['A_diag'][q, s, e, 1 + f])
                    elif not bnd:
                        upd('A_cpl', (lf @ T['TFn'][ty, f]).reshape(10, 10), ref['A_cpl'][q, s, -(nb + 1), t.face_pos[e, f]])
                    upd('Cf', lc @ (T['TCb'] if bnd else T['TC'])[ty, f], ref['Cf'][q, s, e, f])
                upd('A_diag', blk.reshape(10, 10), ref['A_diag'][q, s, e, 0])
                lq = lam[spec.o_c:] / lh[e, spec.nB:]
                upd('Aab', (lq @ T['TAB'][ty]).reshape(10, 4) * sg[e][None, :], ref['Aab'][q, s, e])
                for q2 in range(Q):
                    upd('Aaa', ((lq * lams[q2][e][spec.o_c:]) @ T['TAA'][ty]).reshape(10, 10), ref['Aaa'][q, q2, s, e])
            upd('Bbb', ((1.0 / lh[e, spec.nB:]) @ T['TB'][ty]).reshape(4, 4) * sg[e][:, None] * sg[e][None, :], ref['Bbb'][s, e])
            upd('b', fs[e, :spec.nB] @ T['TPH'][ty], ref['b'][s, 10 * e:10 * e + 10])
            upd('ebar', (lbs[e] @ T['TE'][ty]).reshape(10, 10), ref['ebar'][s, e])
        upd('bdiv', fs[:, spec.nB:] @ T['WC'], ref['bdiv'][s])
        upd('f2', (fs[:, :spec.nB] ** 2 @ T['WB']).sum(), ref['f2'][s])
        upd('ceps', lh[:, :spec.nB].min() * np.linalg.eigvalsh(0.5 * (p['kappa'] + p['kappa'].T)).min(), ref['ceps'][s])
    assert all(v < 1e-12 for v in worst.values()), worst
