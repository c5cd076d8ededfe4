"""BASELINE.json config 5 at full size (3D diffusion, 8 x 8 x 8 subdomains, SWIPDG p = 2, k_c = 4, local basis dim 30): the workload
`bench.py --config cfg5` times, on the kernel path it times (512 subdomains: one workgroup per (subdomain, operator), in-kernel
epilogues, no K-split).  The oracle needs minutes for this size, so the checks are size-independent properties (3D counterpart of
tests/test_parity_gpu.py::test_full_size_properties_config3); the same template and basis size against the oracle on four
subdomains is the 'cfg5_template' case of tests/test_parity3d_gpu.py, forced through the same ksplit = 1 path.

PARITY UNPINNED beyond the oracle: the reference has no 3D / P2 counterpart (discretize_elliptic_block_swipdg.py:22-23)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def cfg5():
    import torch
    from pylrbms_amd import multiscale_problem3d
    from pylrbms_amd.engine3d import Engine3D
    p = multiscale_problem3d.init_grid_and_problem({'num_subdomains': (8, 8, 8), 'cubes_per_subdomain': 4})
    lam = p['lambda']
    eng = Engine3D(p['grid'], lam['functions'], p['f'], p['lambda_bar'], p['lambda_hat'], data_degree=p['data_degree']).assemble()
    N = 30
    g = torch.Generator(device='cuda').manual_seed(11)
    V = torch.randn(eng.S_ext, eng.t.n, N, dtype=torch.float64, device='cuda', generator=g)
    V[:, :, 0] = 1.0
    V = torch.linalg.qr(V)[0].contiguous()                     # well-conditioned reduced systems; first column stays the constant
    out = eng.project_and_estimate(V)
    torch.cuda.synchronize()
    yield dict(p=p, eng=eng, V=V, out=out, N=N)
    del eng, V, out
    torch.cuda.empty_cache()


def _rel(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-300)


def test_outputs_are_finite_and_the_automatic_launch_is_the_unsplit_one(cfg5):
    """At 512 subdomains the launcher takes ksplit = 1 for every projection kernel: forcing it changes no bit; forcing the
    K-split (partial tiles + k3_pg_combine) agrees to summation-order rounding; one stream instead of three changes no bit."""
    import torch
    eng, V, out, N = cfg5['eng'], cfg5['V'], cfg5['out'], cfg5['N']
    for k, v in out.items():
        assert bool(torch.isfinite(v).all()), k
    work = eng.alloc_work(N)
    try:
        for opt, val, exact in (('ksplit', 1, True), ('serial', 1, True), ('ksplit', 2, False)):
            eng.ctx.set_option(opt, val)
            o2 = eng.alloc_outputs(N)
            for v in o2.values():
                v.fill_(float('nan'))
            eng.project_and_estimate(V, o2, work)
            torch.cuda.synchronize()
            eng.ctx.set_option(opt, 0)
            for k in out:
                if exact:
                    assert torch.equal(out[k], o2[k]), (opt, val, k)
                else:
                    assert _rel(o2[k], out[k]) <= 1e-13, (opt, val, k, _rel(o2[k], out[k]))
            del o2
    finally:
        eng.ctx.set_option('ksplit', 0)
        eng.ctx.set_option('serial', 0)


def test_phased_pass_is_bit_identical(cfg5):
    import torch
    eng, V, out, N = cfg5['eng'], cfg5['V'], cfg5['out'], cfg5['N']
    o2, work = eng.alloc_outputs(N), eng.alloc_work(N)
    for v in o2.values():
        v.fill_(float('nan'))
    work.fill_(float('nan'))      # a recycled allocation may still hold the intermediates of the whole pass
    eng.ctx.project_estimate(eng.Q, V, eng.ops, work, o2, phase=1)
    eng.ctx.project_estimate(eng.Q, V, eng.ops, work, o2, phase=2)
    torch.cuda.synchronize()
    for k in out:
        assert torch.equal(out[k], o2[k]), k


def test_symmetries_of_the_projected_operators(cfg5):
    import torch
    eng, out = cfg5['eng'], cfg5['out']
    Q = eng.Q
    B = out['B_sys']                                            # [Q, S, 7, N, N]
    tol = 1e-12
    assert _rel(B[:, :, 3].transpose(2, 3), B[:, :, 3]) <= tol
    for k in ('G_nc', 'G_bb', 'G_rdd'):
        assert _rel(out[k].transpose(1, 2), out[k]) <= tol, k
    for q in range(Q):
        for q2 in range(Q):
            assert _rel(out['G_aa'][q, q2].transpose(1, 2), out['G_aa'][q2, q]) <= tol
    # coupling blocks: block [s, slot] is the transpose of block [neighbour, 6 - slot]
    nbr = torch.as_tensor(eng.nbr.astype(np.int64), device=B.device)
    scale = float(B.abs().max())
    for slot in (0, 1, 2):
        s_idx = torch.nonzero(nbr[:, slot] >= 0)[:, 0]
        other = nbr[s_idx, slot]
        for q in range(Q):
            a, b = B[q, s_idx, slot], B[q, other, 6 - slot].transpose(1, 2)
            assert float((a - b).abs().max()) <= tol * scale, (slot, q)
    # no neighbour: the block is zero
    for slot in (0, 1, 2, 4, 5, 6):
        s_idx = torch.nonzero(nbr[:, slot] < 0)[:, 0]
        assert float(B[:, s_idx, slot].abs().max()) == 0.0


def test_projected_system_against_the_block_operator(cfg5):
    """B_sys and rhs_red against kernels that share nothing with the projection: u^T B(mu) u' == (V u)^T A(mu) (V u') through
    lrbms3_fom_apply (three columns at once), rhs_red == V^T b through torch."""
    import torch
    eng, V, out, N = cfg5['eng'], cfg5['V'], cfg5['out'], cfg5['N']
    S = eng.S
    th = np.array([1.0, 0.37])
    g = torch.Generator(device='cuda').manual_seed(3)
    U = torch.randn(S, N, 3, dtype=torch.float64, device='cuda', generator=g)
    X = torch.einsum('snj,sjm->snm', V, U).contiguous()         # [S, n, 3] reconstructions
    Y = eng.ctx.fom_apply(eng.Q, th, eng.ops['A_diag'], eng.ops['A_cpl'], X)
    lhs = torch.einsum('snj,snm->sjm', V, Y)                    # V^T A V U, per subdomain rows
    Bm = torch.einsum('q,qsabc->sabc', torch.as_tensor(th, device=V.device), out['B_sys'])
    nbr = torch.as_tensor(eng.nbr.astype(np.int64), device=V.device)
    rhs = torch.zeros_like(lhs)
    for slot in range(7):
        ok = nbr[:, slot] >= 0
        rhs[ok] += torch.einsum('sab,sbm->sam', Bm[ok, slot], U[nbr[ok, slot]])
    assert _rel(rhs, lhs) <= 1e-11, _rel(rhs, lhs)
    assert _rel(out['rhs_red'], torch.einsum('snj,sn->sj', V, eng.ops['b'])) <= 1e-12


def test_reduced_estimate_is_the_estimate_of_the_reconstruction(cfg5):
    """The defining property of the projection at full size, through two different instantiations of the kernels: the estimate from
    the N = 30 operators at coefficients u == the estimate from the pass over the ONE-column basis V u at coefficient 1."""
    import torch
    eng, V, out, N = cfg5['eng'], cfg5['V'], cfg5['out'], cfg5['N']
    th = np.array([1.0, 0.62])
    g = torch.Generator(device='cuda').manual_seed(5)
    u = torch.randn(eng.S_ext, N, dtype=torch.float64, device='cuda', generator=g)
    eta = eng.reduced_estimate(th, u, out)
    Uv = torch.einsum('snj,sj->sn', V, u)[:, :, None].contiguous()
    out1 = eng.project_and_estimate(Uv)
    eta1 = eng.reduced_estimate(th, torch.ones(eng.S_ext, 1, dtype=torch.float64, device='cuda'), out1)
    scale = eta1.abs().max(dim=1, keepdim=True).values
    assert float(((eta - eta1).abs() / scale).max()) < 1e-9
    assert bool((eta >= 0).all())
    # ... and the batched estimate of 16 parameters is the single one, column by column
    mus = np.linspace(0.1, 1.0, 16)
    thetas = np.stack([np.array([1.0, m]) for m in mus])
    Ub = u[:, :, None].repeat(1, 1, 16).contiguous()
    etab = eng.ctx.reduced_estimate_batch(eng.Q, thetas, Ub, out, eng.ops, eng.hdiam)
    for m in (0, 7, 15):
        one = eng.reduced_estimate(thetas[m], u, out)
        assert _rel(etab[:, :, m], one) <= 1e-11, m


def test_batched_reduced_solve_at_full_size(cfg5):
    """64 parameters in one call (four groups of 16 on four streams) with the prebuilt two-level preconditioner: residuals of the
    returned solutions recomputed on the host side of the C ABI (torch, from the projected blocks) <= 1e-11, columns equal to
    single-parameter solves to 1e-10, and the coarse level does reduce the iteration count at 8 x 8 x 8 subdomains."""
    import torch
    eng, out, N = cfg5['eng'], cfg5['out'], cfg5['N']
    Q, S = eng.Q, eng.S
    mus = np.linspace(0.1, 1.0, 64)
    thetas = np.stack([np.array([1.0, m]) for m in mus])
    u0, (it0, res0) = eng.ctx.reduced_solve_batch(Q, thetas[:16], out['B_sys'], out['rhs_red'], rtol=1e-12)
    pc = eng.ctx.reduced_precond_build(Q, np.array([1.0, 0.55]), out['B_sys'])
    eng.ctx.reduced_precond_use(pc)
    try:
        ub, (it, res) = eng.ctx.reduced_solve_batch(Q, thetas, out['B_sys'], out['rhs_red'], rtol=1e-12)
    finally:
        eng.ctx.reduced_precond_use(None)
    assert res <= 1e-12 and res0 <= 1e-12
    assert it < it0, (it, it0)                                  # 24 against 48 iterations when this was written
    nbr = torch.as_tensor(eng.nbr.astype(np.int64), device=ub.device)
    bn = float(out['rhs_red'].norm())
    for m in (0, 15, 16, 33, 63):
        Bm = torch.einsum('q,qsabc->sabc', torch.as_tensor(thetas[m], device=ub.device), out['B_sys'])
        r = -out['rhs_red'].clone()
        for slot in range(7):
            ok = nbr[:, slot] >= 0
            r[ok] += torch.einsum('sab,sb->sa', Bm[ok, slot], ub[nbr[ok, slot], :, m])
        assert float(r.norm()) <= 1e-11 * bn, (m, float(r.norm()) / bn)
    for m in (0, 15):
        assert _rel(ub[:, :, m], u0[:, :, m]) <= 1e-10
    us, _ = eng.reduced_solve(thetas[5], out, rtol=1e-13, max_iter=20000)
    assert _rel(ub[:, :, 5], us) <= 1e-10
