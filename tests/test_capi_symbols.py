"""CPU: the C-ABI library builds/loads and exports every symbol include/pygps_amd.h declares, and the
ctypes prototypes cover exactly that set (no compute calls: there is no GPU here)."""
import os
import re

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "pygps_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(pgp_[a-z0-9_]+)\s*\(", src))


def test_library_exports_every_declared_symbol():
    from pygps_amd import _lib
    dll = _lib.load()
    names = _declared()
    assert len(names) >= 24
    for n in names:
        assert hasattr(dll, n), n


def test_ctypes_prototypes_match_header():
    from pygps_amd import _lib
    assert set(_lib.SIGNATURES) == _declared()
    # the public header carries the drop-in boundary + measurement only: every self-test hook lives in the private
    # csrc/testhooks.h (and is bound through a separate table)
    assert not [n for n in _declared() if n.startswith("pgp_test_")]
    priv = open(os.path.join(ROOT, "pygps_amd", "csrc", "testhooks.h")).read()
    priv = re.sub(r"/\*.*?\*/", "", priv, flags=re.S)
    assert set(re.findall(r"\b(pgp_test_[a-z0-9_]+)\s*\(", priv)) == set(_lib.TEST_SIGNATURES)
    dll = _lib.load()
    for n in _lib.TEST_SIGNATURES:
        assert hasattr(dll, n), n


def test_strerror_and_version_without_gpu():
    from pygps_amd import _lib
    dll = _lib.load()
    assert b"gfx950" in dll.pgp_version()
    assert _lib.strerror(0) == "ok" and "positive definite" in _lib.strerror(3)
    assert dll.pgp_profile_classes() >= 8 and b"gemm" in dll.pgp_profile_class_name(1)
