"""The C-ABI shared library loads on a box without a GPU and exports every symbol include/lrbms_hip.h declares
(no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'lrbms_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(lrbms_[a-z_0-9]+)\s*\(', text)))


@pytest.fixture(scope='module')
def lib():
    from pylrbms_amd._build import build_native
    return build_native()


def test_every_declared_symbol_is_exported_and_bound(lib):
    from pylrbms_amd import _native
    names = declared_symbols()
    assert len(names) >= 18
    handle = ctypes.CDLL(lib)
    for name in names:
        assert hasattr(handle, name), name
    assert sorted(_native.SIGNATURES) == names       # the ctypes table binds exactly the declared surface
    bound = _native.load_library(lib)
    assert bound.lrbms_version().decode().startswith('lrbms_hip')


def test_every_declared_3d_symbol_is_exported_and_bound(lib):
    """include/lrbms3d_hip.h (config 5: 3D, P2): the same contract as the 2D header."""
    from pylrbms_amd import _native3d
    text = open(os.path.join(ROOT, 'include', 'lrbms3d_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    names = sorted(set(re.findall(r'\b(lrbms3_[a-z_0-9]+)\s*\(', text)))
    assert len(names) >= 16
    handle = ctypes.CDLL(lib)
    for name in names:
        assert hasattr(handle, name), name
    assert sorted(_native3d.SIGNATURES3) == names
    _native3d.load_library(lib)


def test_3d_product_fails_loudly_without_a_gpu(lib):
    import torch
    from pylrbms_amd._native import NativeError
    from pylrbms_amd._native3d import Native3DContext
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(NativeError):
        Native3DContext(0)


def test_product_fails_loudly_without_a_gpu(lib):
    import torch
    from pylrbms_amd._native import NativeContext, NativeError
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(NativeError):
        NativeContext(0)


def test_missing_library_is_an_error_not_a_fallback(tmp_path):
    from pylrbms_amd import _native
    with pytest.raises(_native.NativeError):
        _native.load_library(str(tmp_path / 'liblrbms_hip.so'))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'pylrbms_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f


def test_emitted_isa_keeps_the_assumptions_of_the_asm_managed_prefetch():
    """k_f1 / k_f2 prefetch with inline-asm loads the compiler does not track (csrc/fused.hip): the gfx950 assembly of the
    product build must show, in every producer loop, the hand-counted ``s_waitcnt vmcnt(n)`` (n > 0), no vmcnt wait and no
    vector-memory instruction of the compiler's own, and no spill in the instantiations the named configurations use
    (pylrbms_amd/_isa_check.py; cross-compiles on the CPU box)."""
    from pylrbms_amd._build import check_isa
    report = check_isa()
    assert any(r.startswith('k_f1u<3,2>') and "vmcnt(16)" in r for r in report), report       # config 3 (N = 40, Q = 2): role A (3 + 9 + 3 + 1 loads)
    assert any(r.startswith("k_f1u<3,2>") and "vmcnt(7)" in r for r in report), report        # ... and role B (6 + 1 loads)
    assert any(r.startswith('k_f1<3,7,2>') and "vmcnt(19)" in r for r in report), report      # producer / consumer form
    assert any(r.startswith('k_f2<5>') and "vmcnt(7)" in r for r in report), report


def test_experiment_switches_are_refused_in_a_product_build(tmp_path):
    import subprocess
    from pylrbms_amd._build import CSRC, FLAGS, _hipcc
    cmd = [_hipcc()] + FLAGS + ['-DF1_NO_MFMA', '-fsyntax-only', '--cuda-host-only', os.path.join(CSRC, 'fused.hip')]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode != 0 and 'experiment switch defined in a product build' in r.stderr
