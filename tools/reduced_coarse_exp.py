"""Iteration counts of the 3D reduced PCG with block-Jacobi alone and with a coarse level on the first local basis vectors (torch, dense blocks)."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from pylrbms_amd import multiscale_problem3d  # noqa: E402
from pylrbms_amd.engine3d import Engine3D  # noqa: E402

P, kc, N = 8, 4, 30
p = multiscale_problem3d.init_grid_and_problem({'num_subdomains': (P, P, P), 'cubes_per_subdomain': kc})
eng = Engine3D(p['grid'], p['lambda']['functions'], p['f'], p['lambda_bar'], p['lambda_hat']).assemble()
g = torch.Generator(device='cuda').manual_seed(0)
V = torch.randn(eng.S_ext, eng.t.n, N, dtype=torch.float64, device='cuda', generator=g)
V[:, :, 0] = 1.0
V = torch.linalg.qr(V)[0].contiguous()
out = eng.project_and_estimate(V)
B, rhs = out['B_sys'], out['rhs_red']                   # [Q][S][7][N][N], [S][N]
S = eng.S
nbr = torch.as_tensor(eng.nbr, device='cuda').long()    # [S][7]
for mu in (0.1, 0.5, 1.0):
    th = torch.tensor([1.0, mu], dtype=torch.float64, device='cuda')
    A = torch.einsum('q,qsabc->sabc', th, B)            # [S][7][N][N]
    def mv(x):
        y = torch.zeros_like(x)
        for slot in range(7):
            t = nbr[:, slot]
            ok = t >= 0
            y[ok] += torch.einsum('sij,sj->si', A[ok, slot], x[t[ok]])
        return y
    Dinv = torch.linalg.inv(A[:, 3])
    A0 = torch.zeros(S, S, dtype=torch.float64, device='cuda')
    for slot in range(7):
        t = nbr[:, slot]
        ok = t >= 0
        A0[torch.arange(S, device='cuda')[ok], t[ok]] = A[ok, slot, 0, 0]
    A0i = torch.linalg.inv(A0)
    def pc1(r): return torch.einsum('sij,sj->si', Dinv, r)
    def pc2(r):
        z = pc1(r)
        z[:, 0] += A0i @ r[:, 0]
        return z
    for name, M in (('block-Jacobi', pc1), ('two-level', pc2)):
        x = torch.zeros_like(rhs); r = rhs.clone(); z = M(r); pp = z.clone(); rz = (r * z).sum(); b0 = rhs.norm()
        for it in range(1, 500):
            Ap = mv(pp); a = rz / (pp * Ap).sum(); x += a * pp; r -= a * Ap
            if r.norm() < 1e-12 * b0: break
            z = M(r); rzn = (r * z).sum(); pp = z + (rzn / rz) * pp; rz = rzn
        print('mu {:.2f} {:13s} iterations {}'.format(mu, name, it), flush=True)
