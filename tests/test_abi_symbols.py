"""CPU: libmi355dr.so loads and exports every function include/mi355dr.h declares (no compute calls)."""

import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _declared():
    text = (ROOT / "include" / "mi355dr.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355dr_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(native_built):
    import ctypes

    lib = ctypes.CDLL(str(native_built))
    names = _declared()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_python_binding_lists_the_same_symbols(native_built):
    from autorag_research_amd import _native

    assert sorted(_native.ABI_SYMBOLS) == _declared()
    _native.load()


def test_no_cpu_fallback_without_gpu(native_built):
    """without a HIP device the library must fail loudly, not fall back"""
    import pytest
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from autorag_research_amd import Mi355Index, NativeError

    with pytest.raises(NativeError):
        Mi355Index(8)


def test_product_never_imports_the_oracle():
    pkg = ROOT / "autorag_research_amd"
    for p in list(pkg.rglob("*.py")) + list(pkg.rglob("*.h")) + list(pkg.rglob("*.hip")):
        src = p.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), p
        assert "liboracle" not in src and "cpu_ref." not in src and "oracle.c\"" not in src, p
