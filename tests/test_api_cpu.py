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


def test_rust_binding_lists_every_symbol_and_handover_files_use_them():
    """the FFI block mirrors the header, not a subset of it; the lib.rs patch and the model hand-over exist as source."""
    header = open(os.path.join(ROOT, "include", "pyskani_amd.h")).read()
    declared = set(re.findall(r"\b(psk_[a-z0-9_]+)\s*\(", header))
    text = open(os.path.join(ROOT, "rust", "ffi.rs")).read()
    bound = set(re.findall(r"pub fn (psk_[a-z0-9_]+)\s*\(", text))
    assert bound == declared, bound ^ declared
    patch = open(os.path.join(ROOT, "rust", "lib_patch.rs")).read()
    for sym in ("psk_ctx_create", "psk_db_create", "psk_sketch_host", "psk_db_add", "psk_query_host", "psk_db_name", "psk_free", "psk_sketch_free"):
        assert f"ffi::{sym}(" in patch, sym
    model = open(os.path.join(ROOT, "rust", "model.rs")).read()
    assert "fn to_psk_model(g: &gbdt::gradient_boost::GBDT)" in model and "ffi::psk_model_create(" in model and "ffi::psk_model_load_json(" in model
    used = set(re.findall(r"ffi::(psk_[a-z0-9_]+)\(", patch + model))
    assert used <= declared, used - declared
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for rel in ("rust/ffi.rs", "rust/lib_patch.rs", "rust/model.rs", "tools/dump_pyskani_goldens.py"):
        assert rel in doc and os.path.exists(os.path.join(ROOT, rel)), rel


def test_comm_entry_points_fail_cleanly_without_a_device():
    """The multi-GPU entry points validate their arguments before touching RCCL or a GPU."""
    import ctypes as C
    from pyskani_amd import _capi
    lib = _capi.load()
    out = C.c_void_p()
    assert lib.psk_comm_create(None, 0, 1, None, C.byref(out)) == _capi.PSK_EINVAL
    assert lib.psk_comm_unique_id(None) == _capi.PSK_EINVAL
    assert lib.psk_gather_hits(None, None, 0, None, None, None) == _capi.PSK_EINVAL
    lib.psk_comm_destroy(None)


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


HIT_SEMANTICS = """
import pickle, sys
if FALLBACK:
    sys.modules["pyskani_amd._hitlist"] = None      # the import of the C module fails: database.py takes its pure-Python twin
from pyskani_amd import database
from pyskani_amd.database import Hit
assert (database._hitlist is None) == FALLBACK
h = Hit(0.9, "q", 0.5, "r", 0.4)
for op in (lambda: len(h), lambda: list(h), lambda: h[0], lambda: h + (), lambda: h * 2, lambda: sorted([h, h]), lambda: h < h, lambda: 0.5 in h):
    try:
        op()
    except TypeError:
        continue
    raise AssertionError("a Hit behaved like a sequence")
a, b = h, pickle.loads(pickle.dumps(h))
assert a == a and a != b and hash(a) != hash(b) and (b.identity, b.query_name, b.reference_name, b.learned, b._raw) == (a.identity, "q", "r", False, None)
assert h.identity == float(__import__("numpy").float32(0.9)) and h.query_fraction == 0.5 and h.reference_fraction == float(__import__("numpy").float32(0.4))
assert repr(h).startswith("Hit(identity=0.8999999") and repr(h).endswith("reference_name='r', reference_fraction=0.4000000059604645)")
for bad in ((1.5, "q", 0.5, "r", 0.4), (0.5, "q", -0.1, "r", 0.4), (0.5, "q", 0.1, "r", 1.4)):
    try:
        Hit(*bad)
    except ValueError:
        continue
    raise AssertionError("range check missing (hit.rs:34-48)")
for name in ("identity", "query_name", "query_fraction", "reference_name", "reference_fraction"):      # getters only: hit.rs:77-104
    try:
        setattr(h, name, 0.5)
    except AttributeError:
        continue
    raise AssertionError(name + " is writable")
import numpy as np
from pyskani_amd import _capi
recs = np.zeros(3, np.dtype(_capi.Hit)); recs["ani"] = [0.5, 0.25, 0.125]; recs["ref_index"] = [2, 0, 1]; recs["learned"] = [0, 1, 0]; recs["n_anchors"] = [7, 8, 9]
hs = Hit._from_records(recs, "query", ["a", "b", "c"])
assert [(x.identity, x.query_name, x.reference_name, x.learned, int(x._raw["n_anchors"])) for x in hs] == [(0.5, "query", "c", False, 7), (0.25, "query", "a", True, 8), (0.125, "query", "b", False, 9)]
assert all(type(x) is Hit for x in hs)
print("ok")
"""


@pytest.mark.parametrize("fallback", [False, True])
def test_hit_is_not_a_sequence(fallback):
    """like the reference's pyclass (hit.rs) a Hit has no length, cannot be iterated, indexed, concatenated or ordered; equality and
    hashing are by identity; the five fields are read-only getters (hit.rs:77-104) and the constructor checks their ranges (hit.rs:34-48).
    Held for the C-level storage (csrc/hitlist.c: HitBase) and for the pure-Python twin database.py takes when that module is not built."""
    import subprocess, sys
    from pyskani_amd import database
    if not fallback and database._hitlist is None:
        pytest.skip("pyskani_amd/_hitlist is not built (make -C pyskani_amd/csrc)")
    r = subprocess.run([sys.executable, "-c", f"FALLBACK = {fallback}\n" + HIT_SEMANTICS], capture_output=True, text=True, cwd=ROOT, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr[-2000:]
