"""ctypes binding of liblrbms_hip.so (C ABI: include/lrbms_hip.h).

There is NO CPU fallback: if the shared library is missing or a call fails this module raises.  PyTorch is only the
device-array container (allocation, stream handle); every number is produced by the HIP kernels.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'liblrbms_hip.so')

c_i32, c_i64, c_dbl, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_double, ctypes.c_void_p
_P_I32 = ctypes.POINTER(c_i32)
_P_DBL = ctypes.POINTER(c_dbl)


class MeshDesc(ctypes.Structure):
    _fields_ = ([(k, c_i32) for k in ('kx', 'ky', 'n_T', 'n_rt', 'n_vertices', 'ncf')] +
                [('hx', c_dbl), ('hy', c_dbl), ('kappa', c_dbl * 4)] +
                [(k, _P_I32) for k in ('nb_elem', 'nb_face', 'nb_elem_out', 'nb_face_out', 'elem_side_pos', 'elem_rt',
                                       'face_sign', 'dof_vertex', 'vdof_ptr', 'vdof_idx', 'rt_e0', 'rt_f0', 'rt_e1',
                                       'rt_f1', 'rt_side', 'side_elem', 'side_elem_out', 'side_count')] +
                [('ntouch', c_i32), ('touch_elem', _P_I32), ('touch_count', _P_I32)] +
                [(k, _P_DBL) for k in ('grad', 'area', 'normal', 'face_len', 'points')])


# name -> (restype, argtypes); exactly the symbols include/lrbms_hip.h declares
SIGNATURES = {
    'lrbms_version': (ctypes.c_char_p, []),
    'lrbms_ctx_create': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(c_vp)]),
    'lrbms_ctx_destroy': (ctypes.c_int, [c_vp]),
    'lrbms_last_error': (ctypes.c_char_p, [c_vp]),
    'lrbms_ctx_aux_stream': (c_vp, [c_vp, c_i32]),
    'lrbms_ctx_set_option': (ctypes.c_int, [c_vp, c_i32, c_i32]),
    'lrbms_set_quadrature': (ctypes.c_int, [c_vp, c_vp]),
    'lrbms_fused_set_subset': (ctypes.c_int, [c_vp, _P_I32, c_i32]),
    'lrbms_set_diagonal_neighbours': (ctypes.c_int, [c_vp, _P_I32]),
    'lrbms_kernel_timing': (ctypes.c_int, [c_vp, c_i32]),
    'lrbms_kernel_timing_read': (ctypes.c_int, [c_vp, ctypes.c_char_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_double), c_i32,
                                                ctypes.POINTER(c_i32)]),
    'lrbms_mesh_upload': (ctypes.c_int, [c_vp, ctypes.POINTER(MeshDesc), c_i32, c_i32, _P_I32]),
    'lrbms_assemble_swipdg': (ctypes.c_int, [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp]),
    'lrbms_assemble_rhs': (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'lrbms_assemble_products': (ctypes.c_int, [c_vp, c_i32, _P_DBL, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'lrbms_assemble_flux': (ctypes.c_int, [c_vp, c_i32, c_vp, c_vp, c_vp]),
    'lrbms_oswald_apply': (ctypes.c_int, [c_vp, c_i32, c_vp, c_vp, c_vp]),
    'lrbms_flux_reconstruct': (ctypes.c_int, [c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    'lrbms_project_system': (ctypes.c_int, [c_vp, c_i32, c_i32] + [c_vp] * 11),
    'lrbms_estimator_work_size': (c_i64, [c_vp, c_i32, c_i32]),
    'lrbms_estimator_grams': (ctypes.c_int, [c_vp, c_i32, c_i32] + [c_vp] * 16),
    'lrbms_fused_supported': (ctypes.c_int, [c_vp, c_i32, c_i32]),
    'lrbms_fused_mfma_per_subdomain': (c_i64, [c_vp, c_i32, c_i32]),
    'lrbms_fused_fnc_ld': (c_i32, [c_vp, c_i32]),
    'lrbms_fused_factored_supported': (ctypes.c_int, [c_vp, c_i32, c_i32]),
    'lrbms_fused_work_size': (c_i64, [c_vp, c_i32, c_i32]),
    'lrbms_project_estimate_fused': (ctypes.c_int, [c_vp, c_i32, c_i32] + [c_vp] * 22),
    'lrbms_project_estimate_fused_phase': (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32] + [c_vp] * 22),
    'lrbms_fside_size': (c_i64, [c_vp, c_i32, c_i32]),
    'lrbms_fnc_size': (c_i64, [c_vp, c_i32]),
    'lrbms_project_estimate_fused_factored': (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32] + [c_vp] * 24),
    'lrbms_reduced_estimate_factored': (ctypes.c_int, [c_vp, c_i32, c_i32, _P_DBL] + [c_vp] * 11 + [c_dbl, c_vp, c_vp]),
    'lrbms_reduced_estimate_batch_factored': (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, _P_DBL] + [c_vp] * 11 + [c_dbl, c_vp, c_vp]),
    'lrbms_reduced_estimate': (ctypes.c_int, [c_vp, c_i32, c_i32, _P_DBL] + [c_vp] * 9 + [c_dbl, c_vp, c_vp]),
    'lrbms_reduced_estimate_batch': (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, _P_DBL] + [c_vp] * 9 + [c_dbl, c_vp, c_vp]),
    'lrbms_reduced_solve_work_size': (c_i64, [c_vp, c_i32]),
    'lrbms_reduced_solve': (ctypes.c_int, [c_vp, c_i32, c_i32, _P_DBL, c_vp, c_vp, c_vp, c_vp, c_dbl, c_i32, _P_DBL, c_vp]),
    'lrbms_reduced_solve_batch_work_size': (c_i64, [c_vp, c_i32, c_i32]),
    'lrbms_reduced_solve_batch': (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, _P_DBL, c_vp, c_vp, c_vp, c_vp, c_dbl, c_i32, _P_DBL,
                                                 c_vp]),
    'lrbms_reduced_precond_size': (c_i64, [c_vp, c_i32]),
    'lrbms_reduced_precond_build': (ctypes.c_int, [c_vp, c_i32, c_i32, _P_DBL, c_vp, c_vp, c_vp, c_vp]),
    'lrbms_reduced_precond_use': (ctypes.c_int, [c_vp, c_i32, c_vp]),
    'lrbms_fom_solve_work_size': (c_i64, [c_vp]),
    'lrbms_fom_solve': (ctypes.c_int, [c_vp, c_i32, _P_DBL, c_vp, c_vp, c_vp, c_vp, c_vp, c_dbl, c_i32, _P_DBL, c_vp]),
    'lrbms_fom_implicit_euler': (ctypes.c_int, [c_vp, c_i32, _P_DBL, c_dbl, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_dbl, c_i32, _P_DBL,
                                                c_vp]),
    'lrbms_mass_inverse_norm2': (ctypes.c_int, [c_vp, c_i32, c_vp, c_vp, c_vp]),
    'lrbms_div_apply': (ctypes.c_int, [c_vp, c_i32, c_i32, c_vp, c_vp, c_vp]),
    'lrbms_div_pairing': (ctypes.c_int, [c_vp, c_i32, c_i32, _P_DBL, c_vp, c_vp, c_vp, c_vp]),
    'lrbms_reduced_reconstruction_terms': (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, _P_DBL] + [c_vp] * 8),
    'lrbms_reduced_implicit_euler': (ctypes.c_int, [c_vp, c_i32, c_i32, _P_DBL, c_dbl, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_dbl,
                                                    c_i32, _P_DBL, c_vp]),
    'lrbms_reduced_time_residual_work_size': (c_i64, [c_vp, c_i32]),
    'lrbms_reduced_time_residual': (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, _P_DBL, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'lrbms_assemble_dirichlet_correction': (ctypes.c_int, [c_vp, c_i32, c_vp, c_vp, c_vp]),
    'lrbms_local_correction_work_size': (c_i64, [c_vp, c_i32]),
    'lrbms_local_correction_solve': (ctypes.c_int, [c_vp, c_i32, _P_DBL, c_i32, _P_I32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_dbl,
                                                    c_i32, _P_DBL, c_vp]),
    'lrbms_blockell_apply': (ctypes.c_int, [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp]),
    'lrbms_fom_apply': (ctypes.c_int, [c_vp, c_i32, c_i32, _P_DBL, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'lrbms_gemm_tn': (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_vp,
                                     c_i64, c_i32, c_vp, c_dbl, c_vp]),
}

_lib = None


class NativeError(RuntimeError):
    pass


def load_library(path=None):
    """Load liblrbms_hip.so and bind every declared symbol; raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get('LRBMS_HIP_LIB') or LIB_PATH     # LRBMS_HIP_LIB: A/B builds of tools/build_variant.sh
    # torch bundles its own libamdhip64: import it FIRST so that our NEEDED libamdhip64.so.7 resolves to the copy
    # torch uses (two HIP runtimes in one process do not share devices, streams or allocations)
    import torch  # noqa: F401
    if not os.path.exists(path):
        raise NativeError('{} is missing: run `python __graft_entry__.py` (or pylrbms_amd/_build.py) to build the HIP '
                          'extension; there is no CPU fallback'.format(path))
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)         # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = restype, argtypes
    _lib = lib
    return lib


def _i32p(a):
    return a.ctypes.data_as(_P_I32)


def _dblp(a):
    return a.ctypes.data_as(_P_DBL)


class NativeContext:
    """One lrbms_ctx per (process, device).  Tensor arguments must be contiguous float64 CUDA tensors on that device;
    shapes are checked here, on the host, before any kernel may dereference them."""

    def __init__(self, device_index=0):
        import torch
        self.torch = torch
        if not torch.cuda.is_available():
            raise NativeError('no HIP device visible: the LRBMS hot path has no CPU fallback')
        self.lib = load_library()
        self.device = torch.device('cuda', device_index)
        handle = c_vp()
        rc = self.lib.lrbms_ctx_create(device_index, ctypes.byref(handle))
        if rc != 0:
            raise NativeError('lrbms_ctx_create failed with code {}'.format(rc))
        self.handle = handle
        self._pid = os.getpid()
        self._keep = None
        self.S = self.S_ext = None

    def close(self):
        if getattr(self, 'handle', None):
            # a fork()ed child (e.g. a multiprocessing manager started after the GPU was initialised) inherits this
            # object but not the device context behind it: freeing the parent's allocations from there aborts the process
            if getattr(self, '_pid', None) == os.getpid():
                self.lib.lrbms_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _check(self, rc, what):
        if rc != 0:
            msg = self.lib.lrbms_last_error(self.handle)
            raise NativeError('{} failed ({}): {}'.format(what, rc, msg.decode() if msg else ''))

    def _stream(self):
        return c_vp(self.torch.cuda.current_stream(self.device).cuda_stream)

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
        return self.torch.from_numpy(np.array(a, dtype=np.float64, order='C', copy=True)).to(self.device)

    # ------------------------------------------------------------------ mesh
    def mesh_upload(self, template, kappa, nbr, S, S_ext):
        t = template
        self.t, self.S, self.S_ext = t, int(S), int(S_ext)
        self.n_T, self.n, self.n_rt, self.ncf = t.n_T, t.n, t.n_rt, t.ncf
        self.nvs = max(t.nvx, t.nvy)
        arrs = {}
        d = MeshDesc()
        d.kx, d.ky, d.n_T, d.n_rt, d.n_vertices, d.ncf = t.kx, t.ky, t.n_T, t.n_rt, t.n_vertices, t.ncf
        d.hx, d.hy = t.hx, t.hy
        d.ntouch = t.ntouch
        kap = np.asarray(kappa, dtype=np.float64).reshape(4)
        for i in range(4):
            d.kappa[i] = kap[i]
        for k in ('nb_elem', 'nb_face', 'nb_elem_out', 'nb_face_out', 'elem_side_pos', 'elem_rt', 'face_sign',
                  'dof_vertex', 'vdof_ptr', 'vdof_idx', 'rt_e0', 'rt_f0', 'rt_e1', 'rt_f1', 'rt_side', 'side_elem',
                  'side_elem_out', 'side_count', 'touch_elem', 'touch_count'):
            arrs[k] = np.ascontiguousarray(getattr(t, k), dtype=np.int32)
            setattr(d, k, _i32p(arrs[k]))
        for k in ('grad', 'area', 'normal', 'face_len', 'points'):
            arrs[k] = np.ascontiguousarray(getattr(t, k), dtype=np.float64)
            setattr(d, k, _dblp(arrs[k]))
        nb = np.ascontiguousarray(nbr, dtype=np.int32)
        assert nb.shape == (S, 5)
        rc = self.lib.lrbms_mesh_upload(self.handle, ctypes.byref(d), S, S_ext, _i32p(nb))
        self._keep = (arrs, nb)
        self._check(rc, 'lrbms_mesh_upload')

    # ------------------------------------------------------------------ assembly
    def set_quadrature(self, spec):
        """Upload the rules of a ``pylrbms_amd.quadrature.QuadratureSpec`` (must precede the assembly calls; after a mesh
        upload).  Keeps the native struct: its o_* / *_stride fields are the sample record layout the host fills."""
        from pylrbms_amd.quadrature import native_quadrature
        self.quad = native_quadrature(spec)
        self.quad_spec = spec
        self._check(self.lib.lrbms_set_quadrature(self.handle, ctypes.byref(self.quad)), 'lrbms_set_quadrature')

    def _quad(self):
        if getattr(self, 'quad', None) is None:
            raise NativeError('set_quadrature() must run before the assembly calls')
        return self.quad

    def assemble_swipdg(self, lam):
        Q, qd = lam.shape[0], self._quad()
        A_diag = self.empty(Q, self.S, self.n_T, 4, 9)
        A_cpl = self.empty(Q, self.S, 4, self.ncf, 9)
        rc = self.lib.lrbms_assemble_swipdg(self.handle, Q, self._ptr(lam, (Q, self.S_ext, self.n_T, qd.lam_stride), 'lam'),
                                            c_vp(A_diag.data_ptr()), c_vp(A_cpl.data_ptr()), self._stream())
        self._check(rc, 'lrbms_assemble_swipdg')
        return A_diag, A_cpl

    def assemble_rhs(self, f_smp, lhat):
        qd = self._quad()
        b, f2, ceps = self.empty(self.S, self.n), self.empty(self.S), self.empty(self.S)
        rc = self.lib.lrbms_assemble_rhs(self.handle, self._ptr(f_smp, (self.S, self.n_T, qd.f_stride), 'f_smp'),
                                         self._ptr(lhat, (self.S, self.n_T, qd.lhat_stride), 'lhat'), c_vp(b.data_ptr()),
                                         c_vp(f2.data_ptr()), c_vp(ceps.data_ptr()), self._stream())
        self._check(rc, 'lrbms_assemble_rhs')
        return b, f2, ceps

    def assemble_products(self, theta_bar, lam, lam_df, lbar, lhat):
        Q, qd = lam.shape[0], self._quad()
        th = np.ascontiguousarray(theta_bar, dtype=np.float64)
        assert th.shape == (Q,)
        P_diag = self.empty(self.S, self.n_T, 4, 9)
        ebar = self.empty(self.S, self.n_T)
        caa = self.empty(Q, Q, self.S, self.n_T)
        Aab = self.empty(Q, self.S, self.n_T, 3, 3)
        Bbb = self.empty(self.S, self.n_T, 3, 3)
        rc = self.lib.lrbms_assemble_products(
            self.handle, Q, _dblp(th), self._ptr(lam, (Q, self.S_ext, self.n_T, qd.lam_stride), 'lam'),
            self._ptr(lam_df, (Q, self.S, self.n_T, qd.lamdf_stride), 'lam_df'),
            self._ptr(lbar, (self.S, self.n_T, qd.lbar_stride), 'lbar'), self._ptr(lhat, (self.S, self.n_T, qd.lhat_stride), 'lhat'),
            c_vp(P_diag.data_ptr()), c_vp(ebar.data_ptr()), c_vp(caa.data_ptr()), c_vp(Aab.data_ptr()),
            c_vp(Bbb.data_ptr()), self._stream())
        self._check(rc, 'lrbms_assemble_products')
        return P_diag, ebar, caa, Aab, Bbb

    def assemble_flux(self, lam):
        Q, qd = lam.shape[0], self._quad()
        F = self.empty(Q, self.S, self.n_rt, 6)
        rc = self.lib.lrbms_assemble_flux(self.handle, Q, self._ptr(lam, (Q, self.S_ext, self.n_T, qd.lam_stride), 'lam'),
                                          c_vp(F.data_ptr()), self._stream())
        self._check(rc, 'lrbms_assemble_flux')
        return F

    # ------------------------------------------------------------------ project + estimate-offline
    def oswald_apply(self, V, out=None):
        N = V.shape[2]
        Wt = out if out is not None else self.empty(self.S, self.n, 5 * N)
        rc = self.lib.lrbms_oswald_apply(self.handle, N, self._ptr(V, (self.S_ext, self.n, N), 'V'),
                                         self._ptr(Wt, (self.S, self.n, 5 * N), 'Wt'), self._stream())
        self._check(rc, 'lrbms_oswald_apply')
        return Wt

    def flux_reconstruct(self, F, V, out=None):
        Q, N = F.shape[0], V.shape[2]
        Rt = out if out is not None else self.empty(self.S, self.n_rt, 5 * Q * N)
        rc = self.lib.lrbms_flux_reconstruct(self.handle, Q, N, self._ptr(F, (Q, self.S, self.n_rt, 6), 'F'),
                                             self._ptr(V, (self.S_ext, self.n, N), 'V'),
                                             self._ptr(Rt, (self.S, self.n_rt, 5 * Q * N), 'Rt'), self._stream())
        self._check(rc, 'lrbms_flux_reconstruct')
        return Rt

    def project_system(self, V, A_diag, A_cpl, P_diag, b, work=None, out=None):
        Q, N, S = A_diag.shape[0], V.shape[2], self.S
        if work is None:
            work = self.empty(Q * S * self.n * N)
        if work.numel() < Q * S * self.n * N:
            raise NativeError('project_system: work too small')
        if out is None:
            out = (self.empty(Q, S, 5, N, N), self.empty(S, N), self.empty(S, N, N), self.empty(S, N, N))
        B_sys, rhs_red, E_red, M_red = out
        rc = self.lib.lrbms_project_system(
            self.handle, Q, N, self._ptr(V, (self.S_ext, self.n, N), 'V'),
            self._ptr(A_diag, (Q, S, self.n_T, 4, 9), 'A_diag'), self._ptr(A_cpl, (Q, S, 4, self.ncf, 9), 'A_cpl'),
            self._ptr(P_diag, (S, self.n_T, 4, 9), 'P_diag'), self._ptr(b, (S, self.n), 'b'), c_vp(work.data_ptr()),
            self._ptr(B_sys, (Q, S, 5, N, N), 'B_sys'), self._ptr(rhs_red, (S, N), 'rhs_red'),
            self._ptr(E_red, (S, N, N), 'E_red'), self._ptr(M_red, (S, N, N), 'M_red'), self._stream())
        self._check(rc, 'lrbms_project_system')
        return B_sys, rhs_red, E_red, M_red

    def estimator_work_size(self, Q, N):
        sz = self.lib.lrbms_estimator_work_size(self.handle, Q, N)
        if sz < 0:
            raise NativeError('lrbms_estimator_work_size failed')
        return int(sz)

    def estimator_grams(self, V, Wt, Rt, ebar, caa, Aab, Bbb, b, work=None, out=None):
        Q, N, S = caa.shape[0], V.shape[2], self.S
        W, C = 5 * N, 5 * Q * N
        need = self.estimator_work_size(Q, N)
        if work is None:
            work = self.empty(need)
        if work.numel() < need:
            raise NativeError('estimator_grams: work too small')
        if out is None:
            out = (self.empty(S, W, W), self.empty(S, C), self.empty(S, 9, Q * N, Q * N), self.empty(S, 9, Q * N, Q * N),
                   self.empty(Q, S, N, C), self.empty(Q, Q, S, N, N))
        G_nc, r_fd, G_rdd, G_bb, G_ab, G_aa = out
        rc = self.lib.lrbms_estimator_grams(
            self.handle, Q, N, self._ptr(V, (self.S_ext, self.n, N), 'V'), self._ptr(Wt, (S, self.n, W), 'Wt'),
            self._ptr(Rt, (S, self.n_rt, C), 'Rt'), self._ptr(ebar, (S, self.n_T), 'ebar'),
            self._ptr(caa, (Q, Q, S, self.n_T), 'caa'), self._ptr(Aab, (Q, S, self.n_T, 3, 3), 'Aab'),
            self._ptr(Bbb, (S, self.n_T, 3, 3), 'Bbb'), self._ptr(b, (S, self.n), 'b'), c_vp(work.data_ptr()),
            self._ptr(G_nc, (S, W, W), 'G_nc'), self._ptr(r_fd, (S, C), 'r_fd'), self._ptr(G_rdd, (S, 9, Q * N, Q * N), 'G_rdd'),
            self._ptr(G_bb, (S, 9, Q * N, Q * N), 'G_bb'), self._ptr(G_ab, (Q, S, N, C), 'G_ab'),
            self._ptr(G_aa, (Q, Q, S, N, N), 'G_aa'), self._stream())
        self._check(rc, 'lrbms_estimator_grams')
        return G_nc, r_fd, G_rdd, G_bb, G_ab, G_aa

    def fused_supported(self, Q, N, factored=False):
        """Whether the fused pass runs this (template, Q, N) with the dense (default) or the factored output layout."""
        fn = self.lib.lrbms_fused_factored_supported if factored else self.lib.lrbms_fused_supported
        return bool(fn(self.handle, Q, N))

    def fused_work_size(self, Q, N):
        sz = self.lib.lrbms_fused_work_size(self.handle, Q, N)
        if sz < 0:
            raise NativeError('lrbms_fused_work_size failed')
        return int(sz)

    def fside_ld(self, Q, N):
        return 4 * Q * N + 4

    def fnc_ld(self, N):
        """Row length of F_nc [S, 4, nvs, .]: A_a | C_a | M_a0 .. M_a3 (| A_diag with the vertex-patch option) (include/lrbms_hip.h)."""
        return int(self.lib.lrbms_fused_fnc_ld(self.handle, int(N)))

    def _gram_ptrs(self, grams, Q, N):
        """Pointers of the projected estimator operators in either layout: 6 tensors = dense (G_rdd / G_bb block-compact
        [S, 9, QN, QN], G_ab [Q, S, N, 5QN]); 8 tensors = factored (self parts [S, N, N] / [S, QN, QN] / [Q, S, N, QN] +
        F_side [S, 4, ncf, 4QN + 4] + F_nc [S, 4, nvs, 2N + 4nvs], include/lrbms_hip.h).  Returns (pointer list, factored flag)."""
        S, W, C, QN = self.S, 5 * N, 5 * Q * N, Q * N
        if len(grams) == 8:
            G_nc, r_fd, G_rdd, G_bb, G_ab, G_aa, Fs, Fn = grams
            return [self._ptr(G_nc, (S, N, N), 'G_nc_self'), self._ptr(r_fd, (S, C), 'r_fd'), self._ptr(G_rdd, (S, QN, QN), 'G_rdd_self'),
                    self._ptr(G_bb, (S, QN, QN), 'G_bb_self'), self._ptr(G_ab, (Q, S, N, QN), 'G_ab_self'),
                    self._ptr(G_aa, (Q, Q, S, N, N), 'G_aa'), self._ptr(Fs, (S, 4, self.ncf, self.fside_ld(Q, N)), 'F_side'),
                    self._ptr(Fn, (S, 4, self.nvs, self.fnc_ld(N)), 'F_nc')], True
        G_nc, r_fd, G_rdd, G_bb, G_ab, G_aa = grams
        return [self._ptr(G_nc, (S, W, W), 'G_nc'), self._ptr(r_fd, (S, C), 'r_fd'), self._ptr(G_rdd, (S, 9, QN, QN), 'G_rdd'),
                self._ptr(G_bb, (S, 9, QN, QN), 'G_bb'), self._ptr(G_ab, (Q, S, N, C), 'G_ab'),
                self._ptr(G_aa, (Q, Q, S, N, N), 'G_aa')], False

    def project_estimate_fused(self, V, F, A_diag, A_cpl, P_diag, b, ebar, caa, Aab, Bbb, work, sys_out, gram_out, phase=0):
        """phase 0: the whole pass; 1 / 2: its halo-independent / halo-dependent halves (lrbms_project_estimate_fused_phase)."""
        Q, N, S = A_diag.shape[0], V.shape[2], self.S
        W, C = 5 * N, 5 * Q * N
        if work.numel() < self.fused_work_size(Q, N):
            raise NativeError('project_estimate_fused: work too small')
        B_sys, rhs_red, E_red, M_red = sys_out
        gptrs, factored = self._gram_ptrs(gram_out, Q, N)
        fn = self.lib.lrbms_project_estimate_fused_factored if factored else self.lib.lrbms_project_estimate_fused_phase
        rc = fn(
            self.handle, int(phase), Q, N, self._ptr(V, (self.S_ext, self.n, N), 'V'), self._ptr(F, (Q, S, self.n_rt, 6), 'F'),
            self._ptr(A_diag, (Q, S, self.n_T, 4, 9), 'A_diag'), self._ptr(A_cpl, (Q, S, 4, self.ncf, 9), 'A_cpl'),
            self._ptr(P_diag, (S, self.n_T, 4, 9), 'P_diag'), self._ptr(b, (S, self.n), 'b'),
            self._ptr(ebar, (S, self.n_T), 'ebar'), self._ptr(caa, (Q, Q, S, self.n_T), 'caa'),
            self._ptr(Aab, (Q, S, self.n_T, 3, 3), 'Aab'), self._ptr(Bbb, (S, self.n_T, 3, 3), 'Bbb'),
            c_vp(work.data_ptr()), self._ptr(B_sys, (Q, S, 5, N, N), 'B_sys'), self._ptr(rhs_red, (S, N), 'rhs_red'),
            self._ptr(E_red, (S, N, N), 'E_red'), self._ptr(M_red, (S, N, N), 'M_red'), *gptrs, self._stream())
        self._check(rc, 'lrbms_project_estimate_fused_phase')

    def bind_project_estimate_fused(self, V, F, A_diag, A_cpl, P_diag, b, ebar, caa, Aab, Bbb, work, sys_out, gram_out):
        """The same call with the argument checks and the pointer marshalling done ONCE: returns ``run(phase=0)``.  A
        sharded step makes three library calls on ~0.2 ms of device work; checking 23 tensors per call made the host
        the slower side.  The closure keeps the tensors alive; it must not outlive a change of their storage."""
        Q, N, S = A_diag.shape[0], V.shape[2], self.S
        W, C = 5 * N, 5 * Q * N
        if work.numel() < self.fused_work_size(Q, N):
            raise NativeError('project_estimate_fused: work too small')
        B_sys, rhs_red, E_red, M_red = sys_out
        gptrs, factored = self._gram_ptrs(gram_out, Q, N)
        ptrs = (self._ptr(V, (self.S_ext, self.n, N), 'V'), self._ptr(F, (Q, S, self.n_rt, 6), 'F'),
                self._ptr(A_diag, (Q, S, self.n_T, 4, 9), 'A_diag'), self._ptr(A_cpl, (Q, S, 4, self.ncf, 9), 'A_cpl'),
                self._ptr(P_diag, (S, self.n_T, 4, 9), 'P_diag'), self._ptr(b, (S, self.n), 'b'),
                self._ptr(ebar, (S, self.n_T), 'ebar'), self._ptr(caa, (Q, Q, S, self.n_T), 'caa'),
                self._ptr(Aab, (Q, S, self.n_T, 3, 3), 'Aab'), self._ptr(Bbb, (S, self.n_T, 3, 3), 'Bbb'),
                c_vp(work.data_ptr()), self._ptr(B_sys, (Q, S, 5, N, N), 'B_sys'), self._ptr(rhs_red, (S, N), 'rhs_red'),
                self._ptr(E_red, (S, N, N), 'E_red'), self._ptr(M_red, (S, N, N), 'M_red')) + tuple(gptrs)
        keep = (V, F, A_diag, A_cpl, P_diag, b, ebar, caa, Aab, Bbb, work, sys_out, gram_out)
        fn = self.lib.lrbms_project_estimate_fused_factored if factored else self.lib.lrbms_project_estimate_fused_phase
        handle, cur, dev = self.handle, self.torch.cuda.current_stream, self.device

        def run(phase=0, stream=None, _keep=keep):      # stream: a raw HIP stream handle (default: torch's current stream)
            rc = fn(handle, phase, Q, N, *ptrs, c_vp(cur(dev).cuda_stream if stream is None else stream))
            if rc != 0:
                self._check(rc, 'lrbms_project_estimate_fused_phase')
        return run

    # ------------------------------------------------------------------ online
    def reduced_estimate(self, theta, u, grams, f2, ceps, hdiam):
        Q, S, N = grams[4].shape[0], self.S, grams[4].shape[2]
        th = np.ascontiguousarray(theta, dtype=np.float64)
        assert th.shape == (Q,)
        eta = self.empty(3, S)
        gptrs, factored = self._gram_ptrs(grams, Q, N)
        fn = self.lib.lrbms_reduced_estimate_factored if factored else self.lib.lrbms_reduced_estimate
        rc = fn(self.handle, Q, N, _dblp(th), self._ptr(u, (self.S_ext, N), 'u'), *gptrs, self._ptr(f2, (S,), 'f2'),
                self._ptr(ceps, (S,), 'ceps'), float(hdiam), c_vp(eta.data_ptr()), self._stream())
        self._check(rc, 'lrbms_reduced_estimate')
        return eta

    def reduced_estimate_batch(self, thetas, u, grams, f2, ceps, hdiam):
        """thetas [nmu, Q], u [S_ext, N, nmu] -> eta_loc [3, S, nmu]."""
        Q, S, N = grams[4].shape[0], self.S, grams[4].shape[2]
        th = np.ascontiguousarray(thetas, dtype=np.float64)
        nmu = th.shape[0]
        assert th.shape == (nmu, Q)
        eta = self.empty(3, S, nmu)
        gptrs, factored = self._gram_ptrs(grams, Q, N)
        fn = self.lib.lrbms_reduced_estimate_batch_factored if factored else self.lib.lrbms_reduced_estimate_batch
        rc = fn(self.handle, Q, N, nmu, _dblp(th), self._ptr(u, (self.S_ext, N, nmu), 'u'), *gptrs, self._ptr(f2, (S,), 'f2'),
                self._ptr(ceps, (S,), 'ceps'), float(hdiam), c_vp(eta.data_ptr()), self._stream())
        self._check(rc, 'lrbms_reduced_estimate_batch')
        return eta

    def reduced_solve(self, theta, B_sys, rhs_red, rtol=1e-13, max_iter=20000, work=None):
        Q, S, N = B_sys.shape[0], self.S, B_sys.shape[3]
        th = np.ascontiguousarray(theta, dtype=np.float64)
        assert th.shape == (Q,)
        need = int(self.lib.lrbms_reduced_solve_work_size(self.handle, N))
        if work is None:
            work = self.empty(need)
        if work.numel() < need:
            raise NativeError('reduced_solve: work too small')
        u = self.empty(S, N)
        info = np.zeros(2)
        rc = self.lib.lrbms_reduced_solve(self.handle, Q, N, _dblp(th), self._ptr(B_sys, (Q, S, 5, N, N), 'B_sys'),
                                          self._ptr(rhs_red, (S, N), 'rhs_red'), c_vp(work.data_ptr()),
                                          c_vp(u.data_ptr()), float(rtol), int(max_iter), _dblp(info), self._stream())
        self._check(rc, 'lrbms_reduced_solve')
        return u, {'iterations': int(info[0]), 'relative_residual': float(info[1])}

    def reduced_solve_batch(self, thetas, B_sys, rhs_red, rtol=1e-13, max_iter=20000, work=None):
        """thetas [nmu, Q] -> u [S, N, nmu] (mu fastest), info."""
        Q, S, N = B_sys.shape[0], self.S, B_sys.shape[3]
        th = np.ascontiguousarray(thetas, dtype=np.float64)
        nmu = th.shape[0]
        assert th.shape == (nmu, Q)
        need = int(self.lib.lrbms_reduced_solve_batch_work_size(self.handle, N, nmu))
        if work is None:
            work = self.empty(need)
        if work.numel() < need:
            raise NativeError('reduced_solve_batch: work too small')
        u = self.empty(S, N, nmu)
        info = np.zeros(2)
        rc = self.lib.lrbms_reduced_solve_batch(self.handle, Q, N, nmu, _dblp(th), self._ptr(B_sys, (Q, S, 5, N, N), 'B_sys'),
                                                self._ptr(rhs_red, (S, N), 'rhs_red'), c_vp(work.data_ptr()),
                                                c_vp(u.data_ptr()), float(rtol), int(max_iter), _dblp(info), self._stream())
        self._check(rc, 'lrbms_reduced_solve_batch')
        return u, {'iterations': int(info[0]), 'relative_residual': float(info[1])}

    def reduced_solve_batches(self, thetas, B_sys, rhs_red, per_call=64, rtol=1e-13, max_iter=20000, concat=True):
        """Parameter sweep: thetas [nmu, Q] for any nmu -> u [S, N, nmu] (``concat=False``: the list of per-call arrays
        [S, N, <= per_call], no copy), info.  ``lrbms_reduced_solve_batch`` takes up to 64 parameters per call and runs them as
        (<= 4) groups of 16 on the caller's stream and the library's side streams itself -- no host threads, one caller per
        context (include/lrbms_hip.h: a ctx is not re-entrant)."""
        torch = self.torch
        th = np.ascontiguousarray(thetas, dtype=np.float64)
        per_call = max(1, min(int(per_call), 64))
        work = self.empty(int(self.lib.lrbms_reduced_solve_batch_work_size(self.handle, int(B_sys.shape[3]), min(per_call, th.shape[0]))))
        res = [self.reduced_solve_batch(th[b0:b0 + per_call], B_sys, rhs_red, rtol=rtol, max_iter=max_iter, work=work)
               for b0 in range(0, th.shape[0], per_call)]
        if concat:
            u = torch.cat([r[0] for r in res], dim=2) if len(res) > 1 else res[0][0]
        else:
            u = [r[0] for r in res]
        return u, {'iterations': max(r[1]['iterations'] for r in res), 'relative_residual': max(r[1]['relative_residual'] for r in res)}

    def reduced_precond_build(self, theta, B_sys):
        """Two-level preconditioner of the reduced solves at the reference parameter ``theta`` -> device buffer."""
        Q, S, N = B_sys.shape[0], self.S, B_sys.shape[3]
        th = np.ascontiguousarray(theta, dtype=np.float64)
        assert th.shape == (Q,)
        work = self.empty(int(self.lib.lrbms_reduced_solve_work_size(self.handle, N)))
        pc = self.empty(int(self.lib.lrbms_reduced_precond_size(self.handle, N)))
        rc = self.lib.lrbms_reduced_precond_build(self.handle, Q, N, _dblp(th), self._ptr(B_sys, (Q, S, 5, N, N), 'B_sys'),
                                                  c_vp(work.data_ptr()), c_vp(pc.data_ptr()), self._stream())
        self._check(rc, 'lrbms_reduced_precond_build')
        pc._lrbms_N = N
        return pc

    def reduced_precond_use(self, pc):
        """Subsequent reduced solves use ``pc`` (``None``: per-call preconditioners).  The context keeps a reference."""
        self._pc_in_use = pc
        rc = self.lib.lrbms_reduced_precond_use(self.handle, int(pc._lrbms_N) if pc is not None else 0,
                                                c_vp(pc.data_ptr()) if pc is not None else None)
        self._check(rc, 'lrbms_reduced_precond_use')

    OPTIONS = {'oswald_zero_on_subdomain_boundary': 1, 'accumulate_coupling_across_q': 2, 'oswald_vertex_patch': 9, 'prep_lds': 10,
               # launch policy (no numerical convention): the library reads no environment variable
               'streams': 3, 'f1_ksplit': 4, 'f1_form': 5, 'coarse': 6, 'solve_valu': 7, 'estimate_valu': 8}

    def set_option(self, name, value):
        """Switch one of the conventions the reference tree leaves open, or the launch policy of the library
        (include/lrbms_hip.h, LRBMS_OPT_*)."""
        if name not in self.OPTIONS:
            raise NativeError('unknown option {!r}; known: {}'.format(name, sorted(self.OPTIONS)))
        self._check(self.lib.lrbms_ctx_set_option(self.handle, self.OPTIONS[name], int(value)), 'lrbms_ctx_set_option')

    def set_diagonal_neighbours(self, nbr_diag):
        """[S, 4] int32: index into the S_ext slabs of the diagonal neighbour at corner SW, SE, NW, NE of every local subdomain
        (or -1) -- read by the Oswald vertex patch (include/lrbms_hip.h: lrbms_set_diagonal_neighbours)."""
        arr = np.ascontiguousarray(np.asarray(nbr_diag, dtype=np.int32))
        self._check(self.lib.lrbms_set_diagonal_neighbours(self.handle, arr.ctypes.data_as(_P_I32)), 'lrbms_set_diagonal_neighbours')

    def fused_set_subset(self, subset):
        """Restrict the following fused passes to the local subdomains ``subset`` (strictly ascending local indices; ``None`` or
        empty: all) -- incremental re-projection after online enrichment (include/lrbms_hip.h: lrbms_fused_set_subset)."""
        if subset is None or len(subset) == 0:
            self._check(self.lib.lrbms_fused_set_subset(self.handle, None, 0), 'lrbms_fused_set_subset')
            return
        arr = np.ascontiguousarray(np.asarray(subset, dtype=np.int32))
        self._check(self.lib.lrbms_fused_set_subset(self.handle, arr.ctypes.data_as(_P_I32), int(arr.size)), 'lrbms_fused_set_subset')

    def fused_mfma_per_subdomain(self, Q, N):
        """fp64 MFMA instructions the dense projection kernel executes per subdomain (bench.py's roofline)."""
        return int(self.lib.lrbms_fused_mfma_per_subdomain(self.handle, int(Q), int(N)))

    def kernel_timing(self, enable):
        """Bracket every kernel of the fused pass by HIP events on its own stream (measurement only)."""
        self._check(self.lib.lrbms_kernel_timing(self.handle, 1 if enable else 0), 'lrbms_kernel_timing')

    def kernel_timing_read(self):
        """[(kernel name, milliseconds)] of the fused passes since the last read (synchronises the device)."""
        cap = 256
        names = ctypes.create_string_buffer(8192)
        ms = (ctypes.c_double * cap)()
        count = c_i32(0)
        self._check(self.lib.lrbms_kernel_timing_read(self.handle, names, 8192, ms, cap, ctypes.byref(count)), 'lrbms_kernel_timing_read')
        nm = names.value.decode().split('\n') if count.value else []
        return [(nm[i], ms[i]) for i in range(count.value)]

    def aux_stream(self, i=0):
        """The i-th library-owned stream as a ``torch.cuda.ExternalStream`` (cached)."""
        cache = self.__dict__.setdefault('_aux_streams', {})
        if i not in cache:
            ptr = self.lib.lrbms_ctx_aux_stream(self.handle, int(i))
            if not ptr:
                raise NativeError('lrbms_ctx_aux_stream({}) returned NULL'.format(i))
            cache[i] = self.torch.cuda.ExternalStream(int(ptr), device=self.device)
        return cache[i]

    # ------------------------------------------------------------------ snapshot generation
    def fom_solve(self, theta, A_diag, A_cpl, b, rtol=1e-12, max_iter=100000):
        """A(mu) x = b for the full-order block operator -> (x [S, n], info)."""
        Q, S = A_diag.shape[0], self.S
        th = np.ascontiguousarray(theta, dtype=np.float64)
        assert th.shape == (Q,)
        work = self.empty(int(self.lib.lrbms_fom_solve_work_size(self.handle)))
        x = self.empty(S, self.n)
        info = np.zeros(2)
        rc = self.lib.lrbms_fom_solve(self.handle, Q, _dblp(th), self._ptr(A_diag, (Q, S, self.n_T, 4, 9), 'A_diag'),
                                      self._ptr(A_cpl, (Q, S, 4, self.ncf, 9), 'A_cpl'), self._ptr(b, (S, self.n), 'b'),
                                      c_vp(work.data_ptr()), c_vp(x.data_ptr()), float(rtol), int(max_iter), _dblp(info),
                                      self._stream())
        self._check(rc, 'lrbms_fom_solve')
        return x, {'iterations': int(info[0]), 'relative_residual': float(info[1])}

    # ------------------------------------------------------------------ parabolic path
    def fom_implicit_euler(self, theta, dt, nt, A_diag, A_cpl, b, U0=None, rtol=1e-12, max_iter=100000):
        """(M + dt A(mu)) u_{k+1} = M u_k + dt b, nt steps -> (U [nt + 1, S, n], info); U0 [S, n] (default 0)."""
        Q, S = A_diag.shape[0], self.S
        th = np.ascontiguousarray(theta, dtype=np.float64)
        assert th.shape == (Q,)
        work = self.empty(int(self.lib.lrbms_fom_solve_work_size(self.handle)))
        U = self.zeros(int(nt) + 1, S, self.n)
        if U0 is not None:
            U[0] = U0.reshape(S, self.n)
        info = np.zeros(2)
        rc = self.lib.lrbms_fom_implicit_euler(self.handle, Q, _dblp(th), float(dt), int(nt),
                                               self._ptr(A_diag, (Q, S, self.n_T, 4, 9), 'A_diag'),
                                               self._ptr(A_cpl, (Q, S, 4, self.ncf, 9), 'A_cpl'), self._ptr(b, (S, self.n), 'b'),
                                               c_vp(work.data_ptr()), c_vp(U.data_ptr()), float(rtol), int(max_iter), _dblp(info),
                                               self._stream())
        self._check(rc, 'lrbms_fom_implicit_euler')
        return U, {'iterations': int(info[0]), 'relative_residual': float(info[1])}

    def mass_inverse_norm2(self, Y):
        """Y [S, n, L] -> [S, L]: y^T M^-1 y per subdomain and column."""
        S, L = self.S, Y.shape[2]
        out = self.empty(S, L)
        rc = self.lib.lrbms_mass_inverse_norm2(self.handle, L, self._ptr(Y, (S, self.n, L), 'Y'), c_vp(out.data_ptr()),
                                               self._stream())
        self._check(rc, 'lrbms_mass_inverse_norm2')
        return out

    def div_apply(self, Rt, mode=0):
        """Rt [S, n_rt, C] -> mode 0: div per element [S, n_T, C]; mode 1: M Div Rt [S, n, C]."""
        S, C = self.S, Rt.shape[2]
        out = self.empty(S, self.n_T if mode == 0 else self.n, C)
        rc = self.lib.lrbms_div_apply(self.handle, C, int(mode), self._ptr(Rt, (S, self.n_rt, C), 'Rt'), c_vp(out.data_ptr()),
                                      self._stream())
        self._check(rc, 'lrbms_div_apply')
        return out

    def div_pairing(self, theta, D, G):
        """D [S, n_T, 5 Q L] (div_apply mode 0), G [S, n, L] -> [S, L]: g^T Div U_r."""
        th = np.ascontiguousarray(theta, dtype=np.float64)
        Q, S, L = len(th), self.S, G.shape[2]
        out = self.empty(S, L)
        rc = self.lib.lrbms_div_pairing(self.handle, Q, L, _dblp(th), self._ptr(D, (S, self.n_T, 5 * Q * L), 'D'),
                                        self._ptr(G, (S, self.n, L), 'G'), c_vp(out.data_ptr()), self._stream())
        self._check(rc, 'lrbms_div_pairing')
        return out

    def reduced_reconstruction_terms(self, theta, B_sys, M_red, rhs_red, G_ud, U):
        """U [L, S, N] -> [L, S]: the elliptic-reconstruction terms of the reduced estimate."""
        Q, S, N, L = B_sys.shape[0], self.S, B_sys.shape[3], U.shape[0]
        th = np.ascontiguousarray(theta, dtype=np.float64)
        work = self.empty(int(self.lib.lrbms_reduced_time_residual_work_size(self.handle, N)))
        out = self.empty(L, S)
        rc = self.lib.lrbms_reduced_reconstruction_terms(
            self.handle, Q, N, L, _dblp(th), self._ptr(B_sys, (Q, S, 5, N, N), 'B_sys'), self._ptr(M_red, (S, N, N), 'M_red'),
            self._ptr(rhs_red, (S, N), 'rhs_red'), self._ptr(G_ud, (S, N, 5 * Q * N), 'G_ud'), self._ptr(U, (L, S, N), 'U'),
            c_vp(work.data_ptr()), c_vp(out.data_ptr()), self._stream())
        self._check(rc, 'lrbms_reduced_reconstruction_terms')
        return out

    def reduced_implicit_euler(self, theta, dt, nt, B_sys, M_red, rhs_red, U0=None, rtol=1e-13, max_iter=20000):
        """(M_red + dt A_red(mu)) u_{k+1} = M_red u_k + dt rhs_red -> (U [nt + 1, S, N], info)."""
        Q, S, N = B_sys.shape[0], self.S, B_sys.shape[3]
        th = np.ascontiguousarray(theta, dtype=np.float64)
        assert th.shape == (Q,)
        work = self.empty(int(self.lib.lrbms_reduced_solve_work_size(self.handle, N)))
        U = self.zeros(int(nt) + 1, S, N)
        if U0 is not None:
            U[0] = U0.reshape(S, N)
        info = np.zeros(2)
        rc = self.lib.lrbms_reduced_implicit_euler(self.handle, Q, N, _dblp(th), float(dt), int(nt),
                                                   self._ptr(B_sys, (Q, S, 5, N, N), 'B_sys'), self._ptr(M_red, (S, N, N), 'M_red'),
                                                   self._ptr(rhs_red, (S, N), 'rhs_red'), c_vp(work.data_ptr()),
                                                   c_vp(U.data_ptr()), float(rtol), int(max_iter), _dblp(info), self._stream())
        self._check(rc, 'lrbms_reduced_implicit_euler')
        return U, {'iterations': int(info[0]), 'relative_residual': float(info[1])}

    def reduced_time_residual(self, theta, B_sys, M_red, dU):
        """dU [L, S, N] -> [L, S]: y^T M_red^-1 y with y = A_red(mu) dU_l."""
        Q, S, N, L = B_sys.shape[0], self.S, B_sys.shape[3], dU.shape[0]
        th = np.ascontiguousarray(theta, dtype=np.float64)
        assert th.shape == (Q,)
        work = self.empty(int(self.lib.lrbms_reduced_time_residual_work_size(self.handle, N)))
        out = self.empty(L, S)
        rc = self.lib.lrbms_reduced_time_residual(self.handle, Q, N, L, _dblp(th), self._ptr(B_sys, (Q, S, 5, N, N), 'B_sys'),
                                                  self._ptr(M_red, (S, N, N), 'M_red'), self._ptr(dU, (L, S, N), 'dU'),
                                                  c_vp(work.data_ptr()), c_vp(out.data_ptr()), self._stream())
        self._check(rc, 'lrbms_reduced_time_residual')
        return out

    # ------------------------------------------------------------------ online enrichment
    def assemble_dirichlet_correction(self, lam):
        Q = lam.shape[0]
        D = self.empty(Q, self.S, 4, self.ncf, 9)
        rc = self.lib.lrbms_assemble_dirichlet_correction(self.handle, Q, self._ptr(lam, (Q, self.S_ext, self.n_T, self._quad().lam_stride), 'lam'),
                                                          c_vp(D.data_ptr()), self._stream())
        self._check(rc, 'lrbms_assemble_dirichlet_correction')
        return D

    def local_correction_solve(self, theta, marked, A_diag, A_cpl, D_corr, b, rtol=1e-12, max_iter=20000):
        """marked: subdomain indices -> (corr [nmark, n], info [nmark, 2] = iterations, relative residual)."""
        Q, S = A_diag.shape[0], self.S
        th = np.ascontiguousarray(theta, dtype=np.float64)
        assert th.shape == (Q,)
        mk = np.ascontiguousarray(marked, dtype=np.int32)
        nmark = int(mk.shape[0])
        work = self.empty(int(self.lib.lrbms_local_correction_work_size(self.handle, nmark)))
        corr = self.empty(nmark, self.n)
        info = np.zeros((nmark, 2))
        rc = self.lib.lrbms_local_correction_solve(
            self.handle, Q, _dblp(th), nmark, mk.ctypes.data_as(_P_I32), self._ptr(A_diag, (Q, S, self.n_T, 4, 9), 'A_diag'),
            self._ptr(A_cpl, (Q, S, 4, self.ncf, 9), 'A_cpl'), self._ptr(D_corr, (Q, S, 4, self.ncf, 9), 'D_corr'),
            self._ptr(b, (S, self.n), 'b'), c_vp(work.data_ptr()), c_vp(corr.data_ptr()), float(rtol), int(max_iter),
            _dblp(info), self._stream())
        self._check(rc, 'lrbms_local_correction_solve')
        return corr, info

    # ------------------------------------------------------------------ helpers
    def blockell_apply(self, A, x):
        M = x.shape[2]
        y = self.empty(self.S, self.n, M)
        rc = self.lib.lrbms_blockell_apply(self.handle, M, self._ptr(A, (self.S, self.n_T, 4, 9), 'A'),
                                           self._ptr(x, (self.S, self.n, M), 'x'), c_vp(y.data_ptr()), self._stream())
        self._check(rc, 'lrbms_blockell_apply')
        return y

    def fom_apply(self, theta, A_diag, A_cpl, x, out=None):
        Q, M = A_diag.shape[0], x.shape[2]
        th = np.ascontiguousarray(theta, dtype=np.float64)
        assert th.shape == (Q,)
        y = out if out is not None else self.empty(self.S, self.n, M)
        rc = self.lib.lrbms_fom_apply(self.handle, Q, M, _dblp(th), self._ptr(A_diag, (Q, self.S, self.n_T, 4, 9), 'A_diag'),
                                      self._ptr(A_cpl, (Q, self.S, 4, self.ncf, 9), 'A_cpl'),
                                      self._ptr(x, (self.S_ext, self.n, M), 'x'), self._ptr(y, (self.S, self.n, M), 'y'),
                                      self._stream())
        self._check(rc, 'lrbms_fom_apply')
        return y

    def gemm_tn(self, X, Y, rowscale=None, alpha=1.0):
        """G[b] = alpha X[b]^T diag(rowscale) Y[b] for X [B, K, Mx], Y [B, K, My]."""
        B, K, Mx = X.shape
        My = Y.shape[2]
        assert Y.shape[:2] == (B, K)
        G = self.empty(B, Mx, My)
        rs = c_vp(rowscale.data_ptr()) if rowscale is not None else c_vp(None)
        rc = self.lib.lrbms_gemm_tn(self.handle, B, K, Mx, My, self._ptr(X, (B, K, Mx), 'X'), K * Mx, Mx,
                                    self._ptr(Y, (B, K, My), 'Y'), K * My, My, c_vp(G.data_ptr()), Mx * My, My, rs,
                                    float(alpha), self._stream())
        self._check(rc, 'lrbms_gemm_tn')
        return G
