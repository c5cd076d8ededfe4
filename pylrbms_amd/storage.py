"""On-disk format for local bases and reduced models (SURVEY.md section 8f "next" #4: "on-disk format for bases /
reduced blocks"; the reference keeps both only in memory and pickles nothing on this path).

One ``.safetensors`` file per object: device tensors exactly as the kernels lay them out (no transposition, no
pickling -- safetensors executes nothing on load), plus a string header that pins what the tensors belong to (grid
shape, subdomain template, affine components, basis width) and is checked on load.

* bases:   ``V`` [S, n, N_max] (zero-padded columns), ``nloc`` [S]
* reduced: ``B_sys`` [Q, S, 5, N, N], ``rhs_red`` [S, N], ``E_red``, ``M_red`` [S, N, N], ``r_fd`` [S, 5QN],
           ``G_aa`` [Q, Q, S, N, N] and either the factored layout ``G_nc`` [S, N, N], ``G_rdd`` / ``G_bb`` [S, QN, QN],
           ``G_ab`` [Q, S, N, QN], ``F_side`` [S, 4, ncf, 4QN + 4], ``F_nc`` [S, 4, nvs, 2N + 4nvs] (default of the fused
           pass) or the dense one ``G_nc`` [S, 5N, 5N], ``G_rdd`` / ``G_bb`` [S, 9, QN, QN] (block-compact), ``G_ab``
           [Q, S, N, 5QN] (include/lrbms_hip.h)
"""
import json

import numpy as np

FORMAT_VERSION = '2'      # 2: the header also pins the quadrature orders and the open conventions the stored arrays were computed with
CONVENTION_NAMES = ('oswald_zero_on_subdomain_boundary', 'accumulate_coupling_across_q', 'oswald_vertex_patch')
_GRAMS = ('G_nc', 'r_fd', 'G_rdd', 'G_bb', 'G_ab', 'G_aa')
_SYS = ('B_sys', 'rhs_red', 'E_red', 'M_red')


def _signature(d, N):
    eng = d.engine
    g, t = d.grid, eng.t
    return {'format': FORMAT_VERSION, 'K': json.dumps([int(k) for k in g.K]), 'P': json.dumps([int(p) for p in g.P]),
            'lower_left': json.dumps([float(v) for v in g.lower_left]), 'upper_right': json.dumps([float(v) for v in g.upper_right]),
            'local_subdomains': json.dumps([int(i) for i in eng.local]), 'n': str(int(t.n)), 'Q': str(int(eng.Q)), 'N': str(int(N)),
            # what the numbers mean: a reduced model stored under other quadrature orders or conventions would be combined with
            # this discretization's f2 / c_eps / operators into inconsistent estimates -- rejected on load like a wrong grid
            'quadrature': json.dumps({k: int(v) for k, v in sorted(eng.quadrature.as_dict().items())}),
            'conventions': json.dumps({k: bool(eng.conventions.get(k, False)) for k in CONVENTION_NAMES})}


def _check(meta, want, what):
    if (meta or {}).get('format') != FORMAT_VERSION:
        raise ValueError('{}: format version {!r}, this build reads {!r} (older files do not record the quadrature orders and '
                         'conventions their arrays were computed with)'.format(what, (meta or {}).get('format'), FORMAT_VERSION))
    for k, v in want.items():
        if meta.get(k) != v:
            raise ValueError('{}: stored {} = {} does not match this discretization ({})'.format(what, k, meta.get(k), v))


def save_bases(reductor, path):
    """Write the local bases of ``reductor`` (``reductor.bases['domain_i']``) to ``path``."""
    import torch
    from safetensors.torch import save_file
    meta = _signature(reductor.d, reductor.basis_size())
    meta['kind'] = 'bases'
    save_file({'V': reductor._V.contiguous(), 'nloc': torch.as_tensor(np.asarray(reductor._nloc, dtype=np.int64))}, path,
              metadata=meta)
    return path


def load_bases(d, path):
    """``bases`` dict (space id -> device array) for ``LRBMSReductor(d, bases=...)``."""
    from safetensors import safe_open
    from pylrbms_amd.vectorarrays import BlockVectorArray, BlockVectorSpace
    with safe_open(path, framework='pt', device=str(d.engine.ctx.device)) as f:
        meta = f.metadata()
        V, nloc = f.get_tensor('V'), f.get_tensor('nloc').cpu().numpy()
    want = _signature(d, V.shape[2])
    want['kind'] = 'bases'
    _check(meta, want, path)
    out = {}
    for i, ii in enumerate(d.engine.local):
        space = BlockVectorSpace([d.solution_space.subspaces[i]])
        out['domain_{}'.format(ii)] = BlockVectorArray(V[i:i + 1, :, :int(nloc[i])], space)
    return out


def save_reduced(rd, path):
    """Write the reduced model ``rd`` (projected system + projected estimator operators) to ``path``."""
    from safetensors.torch import save_file
    meta = _signature(rd.d, rd.N)
    meta['kind'] = 'reduced'
    meta['local_sizes'] = json.dumps(rd.reductor.local_sizes())
    tensors = dict(zip(_SYS, (rd.B_sys, rd.rhs_red, rd.E_red, rd.M_red)))
    tensors.update(dict(zip(_GRAMS + ('F_side', 'F_nc'), rd.grams)))       # zip stops after 6 tensors for the dense layout
    save_file({k: v.contiguous() for k, v in tensors.items()}, path, metadata=meta)
    return path


def load_reduced(reductor, path, cls=None):
    """Reduced model for ``reductor`` (which provides the bases for ``reconstruct``) from ``path`` -- no projection pass."""
    from safetensors import safe_open
    from pylrbms_amd.reductor import ReducedDiscretization
    d = reductor.d
    with safe_open(path, framework='pt', device=str(d.engine.ctx.device)) as f:
        meta = f.metadata()
        tensors = {k: f.get_tensor(k) for k in _SYS + _GRAMS}
        for k in ('F_side', 'F_nc'):
            if k in f.keys():
                tensors[k] = f.get_tensor(k)
    N = int(meta['N'])
    want = _signature(d, N)
    want['kind'] = 'reduced'
    _check(meta, want, path)
    if json.loads(meta['local_sizes']) != reductor.local_sizes():
        raise ValueError('{}: stored local basis sizes do not match the reductor'.format(path))
    buffers = {'sys': tuple(tensors[k] for k in _SYS),
               'grams': [tensors[k] for k in _GRAMS] + ([tensors['F_side'], tensors['F_nc']] if 'F_side' in tensors else [])}
    return (cls or ReducedDiscretization)(reductor, buffers, N)
