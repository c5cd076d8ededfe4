"""Build liblrbms_hip.so in-tree with hipcc for gfx950 (no JIT cache: the .so travels with the repo snapshot)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'liblrbms_hip.so')
SOURCES = ['capi.hip', 'assemble.hip', 'apply.hip', 'gemm.hip', 'fused.hip', 'online.hip', 'enrich.hip', 'fom.hip', 'lrbms3d.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError('hipcc not found')


def source_sha():
    """sha256 over the kernel sources and the two headers, in a fixed order: the identity of the build a profile was taken from
    (tools/pmc_traffic.py records it, bench.py reports measured HBM traffic only for the same sources)."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(('.hip', '.h'))]
    files += [os.path.join(HERE, '..', 'include', 'lrbms_hip.h'), os.path.join(HERE, '..', 'include', 'lrbms3d_hip.h')]
    for f in files:
        with open(f, 'rb') as fh:
            h.update(os.path.basename(f).encode() + b'\0' + fh.read())
    return h.hexdigest()[:16]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, '..', 'include', 'lrbms_hip.h'), os.path.join(HERE, '..', 'include', 'lrbms3d_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def fused_assembly(force=False):
    """gfx950 assembly of csrc/fused.hip as the product build compiles it (same flags, device side only), cached in
    csrc/_obj/fused.s; input of pylrbms_amd._isa_check."""
    objdir = os.path.join(CSRC, '_obj')
    os.makedirs(objdir, exist_ok=True)
    out = os.path.join(objdir, 'fused.s')
    deps = [os.path.join(CSRC, 'fused.hip'), os.path.join(CSRC, 'lrbms_dev.h'), os.path.join(HERE, '..', 'include', 'lrbms_hip.h')]
    if force or not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        cmd = [_hipcc()] + [f for f in FLAGS if f != '-fPIC'] + ['-S', '--cuda-device-only', os.path.join(CSRC, 'fused.hip'), '-o', out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc -S failed for fused.hip:\n{}'.format(r.stderr))
    return out


def check_isa(force=False):
    """Guard of the compiler-invisible prefetch in k_f1 / k_f2 (see _isa_check.py); raises on a violation."""
    from pylrbms_amd._isa_check import check_fused_isa
    return check_fused_isa(fused_assembly(force=force))


def build_native(force=False, verbose=False):
    """Compile every HIP translation unit and link the shared library.  Returns the library path."""
    if not force and not _stale():
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(HERE, 'csrc', '_obj')
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, src.replace('.hip', '.o'))
        cmd = [hipcc] + FLAGS + ['-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for {}:\n{}'.format(src, r.stderr))
        if verbose and r.stderr:
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES) + 1, os.cpu_count() or 1)) as pool:
        isa = pool.submit(check_isa, True)          # beside the compiles: the emitted ISA keeps the prefetch's assumptions
        objs = list(pool.map(compile_one, SOURCES))
        isa.result()
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + ['-L/opt/rocm/lib', '-lrocsolver', '-lrocblas']
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n{}'.format(r.stderr))
    return LIB


if __name__ == '__main__':
    print(build_native(force='--force' in sys.argv, verbose=True))
