"""ctypes binding of the 3D / P2 entry points of liblrbms_hip.so (C ABI: include/lrbms3d_hip.h).  No CPU fallback."""
import ctypes

import numpy as np

from pylrbms_amd import _native
from pylrbms_amd._native import NativeError, c_dbl, c_i32, c_i64, c_vp, _P_DBL, _P_I32

_INT_FIELDS = ('n_T', 'n_rt', 'ncf', 'nvs', 'n_nodes', 'nb', 'nbel', 'nsel', 'nbd', 'nA', 'nB', 'nC', 'nFs', 'nFf', 'o_fs', 'o_ff', 'o_c',
               'lam_stride', 'hat_stride', 'f_stride')
_I32_TABLES = ('elem_type', 'up_face', 'order', 'nb_elem', 'nb_out', 'face_pos', 'tsign', 'elem_rt', 'rt_e0', 'rt_f0', 'rt_e1', 'rt_f1', 'side_elem',
               'side_face', 'side_elem_out', 'side_face_out', 'dof_node', 'node_ptr', 'node_dofs', 'node_mask', 'node_count',
               'side_nodes', 'sn_ptr', 'sn_dofs', 'dof_bslot', 'bn_ptr', 'bn_slots', 'bnodes', 'bnode_sides', 'bel_elem', 'bel_bnode', 'sel_elem', 'sel_sf')
_DBL_TABLES = ('divc', 'TV', 'TE', 'TAA', 'TFo', 'TFn', 'TFb', 'TPo', 'TPn', 'TPb', 'TC', 'TCb', 'TPH', 'TM', 'TB', 'TAB', 'WB', 'WC')


class MeshDesc3D(ctypes.Structure):
    _fields_ = ([(k, c_i32) for k in _INT_FIELDS] + [('volume', c_dbl), ('kmin', c_dbl)] +
                [(k, _P_I32) for k in _I32_TABLES] + [(k, _P_DBL) for k in _DBL_TABLES])


SIGNATURES3 = {
    'lrbms3_ctx_create': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(c_vp)]),
    'lrbms3_ctx_destroy': (ctypes.c_int, [c_vp]),
    'lrbms3_last_error': (ctypes.c_char_p, [c_vp]),
    'lrbms3_ctx_set_option': (ctypes.c_int, [c_vp, c_i32, c_i32]),
    'lrbms3_mesh_upload': (ctypes.c_int, [c_vp, ctypes.POINTER(MeshDesc3D), c_i32, c_i32, _P_I32, _P_I32]),
    'lrbms3_assemble_system': (ctypes.c_int, [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp]),
    'lrbms3_assemble_rhs': (ctypes.c_int, [c_vp] + [c_vp] * 7),
    'lrbms3_assemble_products': (ctypes.c_int, [c_vp, c_i32] + [c_vp] * 8),
    'lrbms3_assemble_flux': (ctypes.c_int, [c_vp, c_i32, c_vp, c_vp, c_vp]),
    'lrbms3_assemble_energy_product': (ctypes.c_int, [c_vp, c_i32, _P_DBL, c_vp, c_vp, c_vp]),
    'lrbms3_energy_product_apply': (ctypes.c_int, [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp]),
    'lrbms3_work_size': (c_i64, [c_vp, c_i32, c_i32]),
    'lrbms3_project_estimate': (ctypes.c_int, [c_vp, c_i32, c_i32] + [c_vp] * 26),
    'lrbms3_project_estimate_phase': (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32] + [c_vp] * 26),
    'lrbms3_kernel_timing': (ctypes.c_int, [c_vp, c_i32]),
    'lrbms3_kernel_timing_read': (ctypes.c_int, [c_vp, ctypes.c_char_p, c_i64, _P_DBL, c_i32, _P_I32]),
    'lrbms3_reduced_estimate': (ctypes.c_int, [c_vp, c_i32, c_i32, _P_DBL] + [c_vp] * 18 + [c_dbl, c_vp, c_vp]),
    'lrbms3_reduced_estimate_batch': (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, _P_DBL] + [c_vp] * 18 + [c_dbl, c_vp, c_vp]),
    'lrbms3_reduced_solve_work_size': (c_i64, [c_vp, c_i32]),
    'lrbms3_reduced_solve': (ctypes.c_int, [c_vp, c_i32, c_i32, _P_DBL, c_vp, c_vp, c_vp, c_vp, c_dbl, c_i32, _P_DBL, c_vp]),
    'lrbms3_reduced_solve_batch_work_size': (c_i64, [c_vp, c_i32, c_i32]),
    'lrbms3_reduced_solve_batch': (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, _P_DBL, c_vp, c_vp, c_vp, c_vp, c_dbl, c_i32, _P_DBL, c_vp]),
    'lrbms3_fom_coarse_space': (ctypes.c_int, [c_vp, c_i32, _P_DBL]),
    'lrbms3_fom_precond_keep': (ctypes.c_int, [c_vp, c_i32]),
    'lrbms3_reduced_precond_size': (c_i64, [c_vp, c_i32]),
    'lrbms3_reduced_precond_work_size': (c_i64, [c_vp, c_i32]),
    'lrbms3_reduced_precond_build': (ctypes.c_int, [c_vp, c_i32, c_i32, _P_DBL, c_vp, c_vp, c_vp, c_vp]),
    'lrbms3_reduced_precond_use': (ctypes.c_int, [c_vp, c_i32, c_vp]),
    'lrbms3_fom_solve_work_size': (c_i64, [c_vp]),
    'lrbms3_fom_solve': (ctypes.c_int, [c_vp, c_i32, _P_DBL, c_vp, c_vp, c_vp, c_vp, c_vp, c_dbl, c_i32, _P_DBL, c_vp]),
    'lrbms3_fom_apply': (ctypes.c_int, [c_vp, c_i32, c_i32, _P_DBL, c_vp, c_vp, c_vp, c_vp, c_vp]),
}

_bound = None


def load_library(path=None):
    global _bound
    if _bound is not None and path is None:
        return _bound
    lib = _native.load_library(path)
    for name, (restype, argtypes) in SIGNATURES3.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = restype, argtypes
    _bound = lib
    return lib


class Native3DContext:
    """One lrbms3_ctx per (process, device); tensors are contiguous float64 CUDA tensors, shapes checked on the host."""

    def __init__(self, device_index=0):
        import os
        import torch
        self.torch = torch
        if not torch.cuda.is_available():
            raise NativeError('no HIP device visible: the LRBMS hot path has no CPU fallback')
        self.lib = load_library()
        self.device = torch.device('cuda', device_index)
        handle = c_vp()
        if self.lib.lrbms3_ctx_create(device_index, ctypes.byref(handle)) != 0:
            raise NativeError('lrbms3_ctx_create failed')
        self.handle, self._pid = handle, os.getpid()

    def close(self):
        import os
        if getattr(self, 'handle', None):
            if self._pid == os.getpid():
                self.lib.lrbms3_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            msg = self.lib.lrbms3_last_error(self.handle)
            raise NativeError('{} failed ({}): {}'.format(what, rc, msg.decode() if msg else ''))

    def _stream(self):
        return c_vp(self.torch.cuda.current_stream(self.device).cuda_stream)

    OPTIONS = {'ksplit': 1, 'serial': 2, 'waves': 3, 'estimate_valu': 4, 'solve_valu': 5, 'fom_coarse': 6}

    def set_option(self, name, value):
        """Launch policy of the library (include/lrbms3d_hip.h, LRBMS3_OPT_*); the library reads no environment variable."""
        if name not in self.OPTIONS:
            raise NativeError('unknown option {!r} (known: {})'.format(name, sorted(self.OPTIONS)))
        self._check(self.lib.lrbms3_ctx_set_option(self.handle, self.OPTIONS[name], int(value)), 'lrbms3_ctx_set_option')

    def _ptr(self, t, shape, name):
        torch = self.torch
        if not isinstance(t, torch.Tensor) or t.dtype != torch.float64 or t.device != self.device:
            raise NativeError('{}: expected a float64 tensor on {}'.format(name, self.device))
        if tuple(t.shape) != tuple(shape):
            raise NativeError('{}: expected shape {}, got {}'.format(name, tuple(shape), tuple(t.shape)))
        if not t.is_contiguous():
            raise NativeError('{}: tensor must be contiguous'.format(name))
        return c_vp(t.data_ptr())

    def empty(self, *shape):
        return self.torch.empty(*shape, dtype=self.torch.float64, device=self.device)

    def zeros(self, *shape):
        return self.torch.zeros(*shape, dtype=self.torch.float64, device=self.device)

    def from_numpy(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(self.device)

    # ------------------------------------------------------------------ mesh
    def mesh_upload(self, template, spec, tables, nbr, phys, S, S_ext):
        t = template
        self.t, self.spec, self.S, self.S_ext = t, spec, int(S), int(S_ext)
        d = MeshDesc3D()
        vals = dict(n_T=t.n_T, n_rt=t.n_rt, ncf=t.ncf, nvs=t.nvs, n_nodes=t.n_nodes, nb=t.nb, nbel=len(t.bel_elem),
                    nsel=len(t.sel_elem), nbd=t.nbd, nA=spec.nA, nB=spec.nB, nC=spec.nC, nFs=spec.nFs, nFf=spec.nFf, o_fs=spec.o_fs,
                    o_ff=spec.o_ff, o_c=spec.o_c, lam_stride=spec.lam_stride, hat_stride=spec.hat_stride, f_stride=spec.f_stride)
        for k, v in vals.items():
            setattr(d, k, int(v))
        d.volume = float(t.volume)
        d.kmin = float(np.linalg.eigvalsh(0.5 * (t.kappa + t.kappa.T)).min())
        keep = {}
        for k in _I32_TABLES:
            keep[k] = np.ascontiguousarray(getattr(t, k), dtype=np.int32).reshape(-1)
            if keep[k].size == 0:
                keep[k] = np.zeros(1, dtype=np.int32)
            setattr(d, k, keep[k].ctypes.data_as(_P_I32))
        for k in _DBL_TABLES:
            src = t.divc if k == 'divc' else tables[k]
            keep[k] = np.ascontiguousarray(src, dtype=np.float64).reshape(-1)
            setattr(d, k, keep[k].ctypes.data_as(_P_DBL))
        nb = np.ascontiguousarray(nbr, dtype=np.int32)
        ph = np.ascontiguousarray(phys, dtype=np.int32)
        assert nb.shape == (S, 7) and ph.shape == (S_ext,)
        rc = self.lib.lrbms3_mesh_upload(self.handle, ctypes.byref(d), S, S_ext, nb.ctypes.data_as(_P_I32), ph.ctypes.data_as(_P_I32))
        self._check(rc, 'lrbms3_mesh_upload')
        self.n_T, self.n, self.n_rt, self.ncf, self.nbf, self.nvs, self.nb, self.n_nodes = (t.n_T, t.n, t.n_rt, t.ncf, t.nbf, t.nvs,
                                                                                              t.nb, t.n_nodes)
        # coarse space of the full-order solver's preconditioner: P1 per subdomain in the local coordinates, centred and scaled
        x = np.asarray(t.node_coordinates(), dtype=np.float64)
        ext = x.max(axis=0) - x.min(axis=0)
        self.fom_coarse_space(np.concatenate([np.ones((t.n, 1)), (x - 0.5 * (x.max(axis=0) + x.min(axis=0))) / ext], axis=1))

    def fom_precond_keep(self, keep=True):
        """The next ``fom_solve`` leaves its coarse inverse in the context, the following ones reuse it for every parameter
        (``False``: drop it, one factorisation per solve again)."""
        self._check(self.lib.lrbms3_fom_precond_keep(self.handle, 1 if keep else 0), 'lrbms3_fom_precond_keep')

    def fom_coarse_space(self, Phi):
        """Phi [n, nc] (nc <= 4) values of the coarse functions of ``fom_solve``'s two-level preconditioner at the local DoFs,
        the same for every subdomain; ``None`` switches the coarse level off."""
        if Phi is None:
            self._check(self.lib.lrbms3_fom_coarse_space(self.handle, 0, None), 'lrbms3_fom_coarse_space')
            return
        Phi = np.ascontiguousarray(Phi, dtype=np.float64)
        assert Phi.ndim == 2 and Phi.shape[0] == self.t.n and 1 <= Phi.shape[1] <= 4
        self._check(self.lib.lrbms3_fom_coarse_space(self.handle, int(Phi.shape[1]), Phi.ctypes.data_as(_P_DBL)), 'lrbms3_fom_coarse_space')

    # ------------------------------------------------------------------ assembly
    def assemble_system(self, lam):
        Q, sp = lam.shape[0], self.spec
        A_diag, A_cpl = self.empty(Q, self.S, self.n_T, 5, 100), self.empty(Q, self.S, 6, self.ncf, 100)
        rc = self.lib.lrbms3_assemble_system(self.handle, Q, self._ptr(lam, (Q, self.S_ext, self.n_T, sp.lam_stride), 'lam'),
                                             c_vp(A_diag.data_ptr()), c_vp(A_cpl.data_ptr()), self._stream())
        self._check(rc, 'lrbms3_assemble_system')
        return A_diag, A_cpl

    def assemble_rhs(self, f_smp, lhat):
        sp = self.spec
        b, f2, ceps, bdiv = self.empty(self.S, self.n), self.empty(self.S), self.empty(self.S), self.empty(self.S, self.n_T)
        rc = self.lib.lrbms3_assemble_rhs(self.handle, self._ptr(f_smp, (self.S, self.n_T, sp.f_stride), 'f_smp'),
                                          self._ptr(lhat, (self.S, self.n_T, sp.hat_stride), 'lhat'), c_vp(b.data_ptr()),
                                          c_vp(f2.data_ptr()), c_vp(ceps.data_ptr()), c_vp(bdiv.data_ptr()), self._stream())
        self._check(rc, 'lrbms3_assemble_rhs')
        return b, f2, ceps, bdiv

    def assemble_products(self, lam, lbar, lhat):
        Q, sp = lam.shape[0], self.spec
        ebar, Aaa = self.empty(self.S, self.n_T, 100), self.empty(Q, Q, self.S, self.n_T, 100)
        Aab, Bbb = self.empty(Q, self.S, self.n_T, 40), self.empty(self.S, self.n_T, 16)
        rc = self.lib.lrbms3_assemble_products(self.handle, Q, self._ptr(lam, (Q, self.S_ext, self.n_T, sp.lam_stride), 'lam'),
                                               self._ptr(lbar, (self.S, self.n_T, sp.nB), 'lbar'),
                                               self._ptr(lhat, (self.S, self.n_T, sp.hat_stride), 'lhat'), c_vp(ebar.data_ptr()),
                                               c_vp(Aaa.data_ptr()), c_vp(Aab.data_ptr()), c_vp(Bbb.data_ptr()), self._stream())
        self._check(rc, 'lrbms3_assemble_products')
        return ebar, Aaa, Aab, Bbb

    def assemble_flux(self, lam):
        Q, sp = lam.shape[0], self.spec
        Cf = self.empty(Q, self.S_ext, self.n_T, 4, 10)
        rc = self.lib.lrbms3_assemble_flux(self.handle, Q, self._ptr(lam, (Q, self.S_ext, self.n_T, sp.lam_stride), 'lam'),
                                           c_vp(Cf.data_ptr()), self._stream())
        self._check(rc, 'lrbms3_assemble_flux')
        return Cf

    def assemble_energy_product(self, theta_bar, lam):
        """P_diag [S, n_T, 5, 100]: the local energy product at mu_bar (block-ELL, local to every subdomain)."""
        Q, sp = lam.shape[0], self.spec
        th = np.ascontiguousarray(theta_bar, dtype=np.float64)
        assert th.shape == (Q,)
        P = self.empty(self.S, self.n_T, 5, 100)
        rc = self.lib.lrbms3_assemble_energy_product(self.handle, Q, th.ctypes.data_as(_P_DBL),
                                                     self._ptr(lam, (Q, self.S_ext, self.n_T, sp.lam_stride), 'lam'),
                                                     c_vp(P.data_ptr()), self._stream())
        self._check(rc, 'lrbms3_assemble_energy_product')
        return P

    def energy_product_apply(self, P_diag, X):
        """P X for X [S, n, M] (M vectors per subdomain)."""
        M = X.shape[2]
        Y = self.empty(self.S, self.n, M)
        rc = self.lib.lrbms3_energy_product_apply(self.handle, M, self._ptr(P_diag, (self.S, self.n_T, 5, 100), 'P_diag'),
                                                  self._ptr(X, (self.S, self.n, M), 'X'), c_vp(Y.data_ptr()), self._stream())
        self._check(rc, 'lrbms3_energy_product_apply')
        return Y

    # ------------------------------------------------------------------ pass
    OUT_NAMES = ('B_sys', 'rhs_red', 'G_nc', 'G_bb', 'G_rdd', 'G_ab', 'G_aa', 'r_fd', 'Rb', 'Yb', 'Dp', 'Xab', 'As', 'Cn')

    def out_shapes(self, Q, N):
        S, QN = self.S, Q * N
        return dict(B_sys=(Q, S, 7, N, N), rhs_red=(S, N), G_nc=(S, N, N), G_bb=(S, QN, QN), G_rdd=(S, QN, QN), G_ab=(Q, S, N, QN),
                    G_aa=(Q, Q, S, N, N), r_fd=(S, QN), Rb=(S, self.nbf, QN), Yb=(S, self.nbf, QN), Dp=(S, self.nbf, QN),
                    Xab=(Q, S, self.nbf, N), As=(S, 6, self.nvs, N), Cn=(S, self.nb, N))

    def work_size(self, Q, N):
        return int(self.lib.lrbms3_work_size(self.handle, Q, N))

    def project_estimate(self, Q, V, ops, work, out, phase=0):
        N = V.shape[2]
        S, nT = self.S, self.n_T
        shp = self.out_shapes(Q, N)
        args = [self._ptr(V, (self.S_ext, self.n, N), 'V'), self._ptr(ops['A_diag'], (Q, S, nT, 5, 100), 'A_diag'),
                self._ptr(ops['A_cpl'], (Q, S, 6, self.ncf, 100), 'A_cpl'), self._ptr(ops['b'], (S, self.n), 'b'),
                self._ptr(ops['ebar'], (S, nT, 100), 'ebar'), self._ptr(ops['Aaa'], (Q, Q, S, nT, 100), 'Aaa'),
                self._ptr(ops['Aab'], (Q, S, nT, 40), 'Aab'), self._ptr(ops['Bbb'], (S, nT, 16), 'Bbb'),
                self._ptr(ops['bdiv'], (S, nT), 'bdiv'), self._ptr(ops['Cf'], (Q, self.S_ext, nT, 4, 10), 'Cf')]
        if work.numel() < self.work_size(Q, N):
            raise NativeError('work buffer too small')
        args.append(c_vp(work.data_ptr()))
        args += [self._ptr(out[k], shp[k], k) for k in self.OUT_NAMES]
        rc = self.lib.lrbms3_project_estimate_phase(self.handle, int(phase), Q, N, *args, self._stream())
        self._check(rc, 'lrbms3_project_estimate_phase')
        return out

    def kernel_timing(self, enable):
        self._check(self.lib.lrbms3_kernel_timing(self.handle, int(bool(enable))), 'lrbms3_kernel_timing')

    def kernel_timing_read(self, cap=4096):
        names = ctypes.create_string_buffer(64 * cap)
        ms = (c_dbl * cap)()
        count = c_i32(0)
        self._check(self.lib.lrbms3_kernel_timing_read(self.handle, names, 64 * cap, ms, cap, ctypes.byref(count)),
                    'lrbms3_kernel_timing_read')
        nm = names.value.decode().split('\n')[:count.value]
        return list(zip(nm, [ms[i] for i in range(count.value)]))

    # ------------------------------------------------------------------ online
    def reduced_estimate(self, Q, theta, u, out, ops, hdiam):
        N = u.shape[1]
        S = self.S
        shp = self.out_shapes(Q, N)
        th = np.ascontiguousarray(theta, dtype=np.float64)
        assert th.shape == (Q,)
        eta = self.empty(3, S)
        names = ('G_nc', 'G_bb', 'G_rdd', 'G_ab', 'G_aa', 'r_fd', 'Rb', 'Yb', 'Dp', 'Xab', 'As', 'Cn')
        args = [self._ptr(u, (self.S_ext, N), 'u')] + [self._ptr(out[k], shp[k], k) for k in names]
        args += [self._ptr(ops['ebar'], (S, self.n_T, 100), 'ebar'), self._ptr(ops['Bbb'], (S, self.n_T, 16), 'Bbb'),
                 self._ptr(ops['bdiv'], (S, self.n_T), 'bdiv'), self._ptr(ops['f2'], (S,), 'f2'), self._ptr(ops['ceps'], (S,), 'ceps')]
        rc = self.lib.lrbms3_reduced_estimate(self.handle, Q, N, th.ctypes.data_as(_P_DBL), *args, float(hdiam), c_vp(eta.data_ptr()),
                                              self._stream())
        self._check(rc, 'lrbms3_reduced_estimate')
        return eta

    def reduced_estimate_batch(self, Q, thetas, u, out, ops, hdiam):
        """thetas [nmu, Q], u [S_ext, N, nmu] (parameter fastest) -> eta_loc [3, S, nmu]."""
        N, nmu, S = u.shape[1], u.shape[2], self.S
        shp = self.out_shapes(Q, N)
        th = np.ascontiguousarray(thetas, dtype=np.float64)
        assert th.shape == (nmu, Q)
        eta = self.empty(3, S, nmu)
        names = ('G_nc', 'G_bb', 'G_rdd', 'G_ab', 'G_aa', 'r_fd', 'Rb', 'Yb', 'Dp', 'Xab', 'As', 'Cn')
        args = [self._ptr(u, (self.S_ext, N, nmu), 'u')] + [self._ptr(out[k], shp[k], k) for k in names]
        args += [self._ptr(ops['ebar'], (S, self.n_T, 100), 'ebar'), self._ptr(ops['Bbb'], (S, self.n_T, 16), 'Bbb'),
                 self._ptr(ops['bdiv'], (S, self.n_T), 'bdiv'), self._ptr(ops['f2'], (S,), 'f2'), self._ptr(ops['ceps'], (S,), 'ceps')]
        rc = self.lib.lrbms3_reduced_estimate_batch(self.handle, Q, N, nmu, th.ctypes.data_as(_P_DBL), *args, float(hdiam),
                                                    c_vp(eta.data_ptr()), self._stream())
        self._check(rc, 'lrbms3_reduced_estimate_batch')
        return eta

    def reduced_solve(self, Q, theta, B_sys, rhs_red, rtol=1e-13, max_iter=5000, work=None):
        N = rhs_red.shape[1]
        S = self.S
        th = np.ascontiguousarray(theta, dtype=np.float64)
        if work is None:
            work = self.empty(int(self.lib.lrbms3_reduced_solve_work_size(self.handle, N)))
        u = self.empty(S, N)
        info = (c_dbl * 2)()
        rc = self.lib.lrbms3_reduced_solve(self.handle, Q, N, th.ctypes.data_as(_P_DBL), self._ptr(B_sys, (Q, S, 7, N, N), 'B_sys'),
                                           self._ptr(rhs_red, (S, N), 'rhs_red'), c_vp(work.data_ptr()), c_vp(u.data_ptr()),
                                           float(rtol), int(max_iter), info, self._stream())
        self._check(rc, 'lrbms3_reduced_solve')
        return u, (int(info[0]), float(info[1]))

    def reduced_solve_batch(self, Q, thetas, B_sys, rhs_red, rtol=1e-13, max_iter=5000, work=None):
        """thetas [nmu, Q] (nmu <= 64: up to four groups of 16 on four streams) -> u [S, N, nmu] (parameter fastest),
        (iterations, worst relative residual)."""
        N, S = rhs_red.shape[1], self.S
        th = np.ascontiguousarray(thetas, dtype=np.float64)
        nmu = th.shape[0]
        assert th.shape == (nmu, Q)
        if work is None:
            work = self.empty(int(self.lib.lrbms3_reduced_solve_batch_work_size(self.handle, N, nmu)))
        u = self.empty(S, N, nmu)
        info = (c_dbl * 2)()
        rc = self.lib.lrbms3_reduced_solve_batch(self.handle, Q, N, nmu, th.ctypes.data_as(_P_DBL),
                                                 self._ptr(B_sys, (Q, S, 7, N, N), 'B_sys'), self._ptr(rhs_red, (S, N), 'rhs_red'),
                                                 c_vp(work.data_ptr()), c_vp(u.data_ptr()), float(rtol), int(max_iter), info,
                                                 self._stream())
        self._check(rc, 'lrbms3_reduced_solve_batch')
        return u, (int(info[0]), float(info[1]))

    def reduced_precond_build(self, Q, theta, B_sys):
        """Two-level preconditioner of the batched reduced solve at the reference parameter ``theta``: one device buffer,
        the coarse inverse [S, S] followed by the inverse diagonal blocks [S, N, N]."""
        N, S = B_sys.shape[-1], self.S
        th = np.ascontiguousarray(theta, dtype=np.float64)
        assert th.shape == (Q,)
        work = self.empty(int(self.lib.lrbms3_reduced_precond_work_size(self.handle, N)))
        pc = self.empty(int(self.lib.lrbms3_reduced_precond_size(self.handle, N)))        # [S, S] coarse inverse | [S, N, N] inverse blocks
        rc = self.lib.lrbms3_reduced_precond_build(self.handle, Q, N, th.ctypes.data_as(_P_DBL), self._ptr(B_sys, (Q, S, 7, N, N), 'B_sys'),
                                                   c_vp(work.data_ptr()), c_vp(pc.data_ptr()), self._stream())
        self._check(rc, 'lrbms3_reduced_precond_build')
        self.torch.cuda.current_stream().synchronize()          # `work` goes out of scope
        pc._lrbms_N = N
        return pc

    def reduced_precond_use(self, pc):
        """Subsequent ``reduced_solve_batch`` calls use ``pc`` (``None``: inverse diagonal blocks alone).  The context keeps a reference."""
        self._pc_keep = pc
        rc = self.lib.lrbms3_reduced_precond_use(self.handle, int(pc._lrbms_N) if pc is not None else 0,
                                                 c_vp(pc.data_ptr()) if pc is not None else None)
        self._check(rc, 'lrbms3_reduced_precond_use')

    def fom_solve(self, Q, theta, A_diag, A_cpl, b, rtol=1e-10, max_iter=50000, work=None):
        th = np.ascontiguousarray(theta, dtype=np.float64)
        if work is None:
            work = self.empty(int(self.lib.lrbms3_fom_solve_work_size(self.handle)))
        x = self.empty(self.S, self.n)
        info = (c_dbl * 2)()
        rc = self.lib.lrbms3_fom_solve(self.handle, Q, th.ctypes.data_as(_P_DBL),
                                       self._ptr(A_diag, (Q, self.S, self.n_T, 5, 100), 'A_diag'),
                                       self._ptr(A_cpl, (Q, self.S, 6, self.ncf, 100), 'A_cpl'), self._ptr(b, (self.S, self.n), 'b'),
                                       c_vp(work.data_ptr()), c_vp(x.data_ptr()), float(rtol), int(max_iter), info, self._stream())
        self._check(rc, 'lrbms3_fom_solve')
        return x, (int(info[0]), float(info[1]))

    def fom_apply(self, Q, theta, A_diag, A_cpl, x):
        M = x.shape[2]
        th = np.ascontiguousarray(theta, dtype=np.float64)
        y = self.empty(self.S, self.n, M)
        rc = self.lib.lrbms3_fom_apply(self.handle, Q, M, th.ctypes.data_as(_P_DBL),
                                       self._ptr(A_diag, (Q, self.S, self.n_T, 5, 100), 'A_diag'),
                                       self._ptr(A_cpl, (Q, self.S, 6, self.ncf, 100), 'A_cpl'),
                                       self._ptr(x, (self.S_ext, self.n, M), 'x'), c_vp(y.data_ptr()), self._stream())
        self._check(rc, 'lrbms3_fom_apply')
        return y
