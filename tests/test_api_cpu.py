"""CPU: host-side logic and the C-ABI surface (no compute calls without a GPU)."""
import os
import re

import pytest

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from pyskani_amd import _capi
    lib = _capi.load()
    header = open(os.path.join(ROOT, "include", "pyskani_amd.h")).read()
    declared = set(re.findall(r"\b(psk_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_capi.SYMBOLS), declared ^ set(_capi.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.psk_version()


def test_rust_binding_and_integration_doc_list_every_symbol():
    """the FFI block must mirror the header, not a subset of it."""
    header = open(os.path.join(ROOT, "include", "pyskani_amd.h")).read()
    declared = set(re.findall(r"\b(psk_[a-z0-9_]+)\s*\(", header))
    for rel in ("rust/ffi.rs", "INTEGRATION.md"):
        text = open(os.path.join(ROOT, rel)).read()
        bound = set(re.findall(r"pub fn (psk_[a-z0-9_]+)\s*\(", text))
        assert bound == declared, (rel, bound ^ declared)


def test_no_gpu_fails_loudly():
    import pyskani_amd
    try:
        pyskani_amd.Database()
    except RuntimeError as e:
        assert "no CPU fallback" in str(e)
    else:
        pytest.skip("GPU present")


def test_hit_validation_and_repr():
    """Hit.__init__ argument order and range checks, hit.rs:27-48; repr, hit.rs:61-74."""
    from pyskani_amd import Hit
    h = Hit(0.5, "q", 0.25, "r", 0.75)
    assert (h.identity, h.query_name, h.query_fraction, h.reference_name, h.reference_fraction) == (0.5, "q", 0.25, "r", 0.75)
    assert repr(h) == "Hit(identity=0.5, query_name='q', query_fraction=0.25, reference_name='r', reference_fraction=0.75)"
    for bad in ((1.5, "q", 0.1, "r", 0.1), (0.5, "q", -0.1, "r", 0.1), (0.5, "q", 0.1, "r", 1.1)):
        with pytest.raises(ValueError):
            Hit(*bad)


def test_product_path_never_imports_oracle():
    pkg = os.path.join(ROOT, "pyskani_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "skani_oracle.h" not in src, f


def test_text_marshalling():
    from pyskani_amd.database import _as_bytes
    import array
    assert _as_bytes("ACGT") == b"ACGT" and _as_bytes(bytearray(b"AC")) == b"AC"
    assert _as_bytes(memoryview(b"GG")) == b"GG" and _as_bytes(array.array("B", b"TT")) == b"TT"
    with pytest.raises(TypeError):
        _as_bytes(12)
