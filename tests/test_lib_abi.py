"""CPU: libjasper_hip.so loads and exports exactly what include/jasper_hip.h declares (no compute calls here)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    h = open(os.path.join(ROOT, "include", "jasper_hip.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(jasper_[a-z0-9_]+)\s*\(", h)))


def test_header_matches_binding_table():
    from jasper_amd import _lib
    assert declared_symbols() == sorted(_lib.SYMBOLS)


def test_library_loads_and_exports_every_symbol():
    from jasper_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "jasper_amd", "csrc")])
    L = _lib.lib()
    for name in declared_symbols():
        assert hasattr(L, name), name


def test_product_does_not_touch_the_oracle():
    """the product path must not import, link or call anything under oracle/"""
    pkg = os.path.join(ROOT, "jasper_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "jasper_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(pkg, "libjasper_hip.so")], capture_output=True, text=True).stdout
    assert " jo_" not in out


def test_missing_library_fails_loudly(monkeypatch):
    from jasper_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libjasper_hip.so")
    import pytest
    with pytest.raises(ImportError):
        _lib.lib()


def test_kmer_hash_is_a_bijection_with_working_inverse():
    """the table stores only part of the hash, so the hash must be a bijection on 2k-bit keys; the .jf writer needs its
    inverse (host arithmetic through the C-ABI test hook: no GPU involved)"""
    import ctypes as C
    import random
    from jasper_amd import _lib
    L = _lib.lib()
    rng = random.Random(5)
    out = (C.c_uint64 * 2)()
    for k in list(range(1, 65)):
        B = 2 * k
        seen = set()
        keys = [0, (1 << B) - 1, 1, 1 << (B - 1)] + [rng.getrandbits(B) for _ in range(40)]
        for key in keys:
            assert L.jasper_debug_mix(k, 0, key >> 64, key & ((1 << 64) - 1), out) == 0
            h = (out[0] << 64) | out[1]
            assert h < (1 << B)
            assert L.jasper_debug_mix(k, 1, out[0], out[1], out) == 0
            assert ((out[0] << 64) | out[1]) == key, (k, hex(key))
            seen.add(h)
        assert len(seen) == len(set(keys))
    # small k exhaustively: every 2k-bit value is hit exactly once
    for k in (1, 2, 3, 5, 7):
        B = 2 * k
        img = set()
        for key in range(1 << B):
            L.jasper_debug_mix(k, 0, 0, key, out)
            img.add(out[1])
        assert len(img) == 1 << B and max(img) < (1 << B)
