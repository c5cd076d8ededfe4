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
