"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/simd_minimizers_amd.h declares, validates arguments exactly like the reference's asserts,
and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "simd_minimizers_amd.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mm_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(sm):
    L = sm.lib()
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in the header but not exported"
    assert sorted(sm.EXPORTED_SYMBOLS) == names


def test_plan_validation_matches_reference_asserts(sm):
    E = sm.ERR
    cases = [
        ((0, 5, False, 0), E["K_ZERO"]),
        ((5, 0, False, 0), E["W_ZERO"]),                     # src/sliding_min.rs:91
        ((5, 1 << 15, False, 0), E["W_TOO_LARGE"]),           # src/sliding_min.rs:92-95
        ((5, 6, True, 0), E["EVEN_L"]),                       # src/canonical.rs:13-16
        ((5, 6, False, 2), E["OPEN_EVEN_W"]),                 # src/syncmers.rs:24-29
        ((5, 7, False, 3), E["BAD_MODE"]),                    # src/lib.rs:437
    ]
    for (k, w, canon, mode), code in cases:
        with pytest.raises(sm.MinimizerError) as e:
            sm.Plan(k, w, canon, mode, None)
        assert e.value.code == code, (k, w, canon, mode)
    with pytest.raises(sm.MinimizerError) as e:               # src/minimizers.rs:81,139
        sm.Plan(5, 7, True, 0, sm.NtHasher(5, canonical=False))
    assert e.value.code == E["HASHER_NOT_CANONICAL"]
    p = sm.Plan(5, 7, True, 0, None)
    assert p.value_len() == 5
    assert sm.Plan(5, 7, True, 1, None).value_len() == 11    # src/lib.rs:439-447
    with pytest.raises(sm.MinimizerError):
        sm.closed_syncmers(5, 7).super_kmers([])              # src/lib.rs:339


def test_default_hasher_tables(sm, oracle):
    for canon in (False, True):
        a, b = sm.NtHasher(21, canon), oracle.default_hasher(canon)
        assert list(a.fw) == list(b.fw) and list(a.rc) == list(b.rc)
        assert a.rot == b.rot == 7 and a.canonical == b.canonical == int(canon)


def test_no_cpu_fallback(sm):
    """Without a GPU every compute entry point must fail loudly (MM_ERR_NO_DEVICE)."""
    L = sm.lib()
    if L.mm_device_count() > 0:
        pytest.skip("GPU present")
    h = C.c_void_p()
    assert L.mm_workspace_create(C.byref(h), 0, None) == sm.ERR["NO_DEVICE"]
    with pytest.raises(sm.MinimizerError) as e:
        sm.minimizer_positions(sm.AsciiSeq(b"ACGTGCTCAGAGACTCAG"), 5, 7)
    assert e.value.code == sm.ERR["NO_DEVICE"]


def test_product_does_not_import_oracle():
    """oracle/ is test infrastructure: nothing in the product package may reference it."""
    pkg = os.path.join(ROOT, "simd-minimizers_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "mm_oracle" not in text and "oracle/" not in text, os.path.join(dirpath, f)


def test_packed_seq_helpers(sm, oracle):
    import numpy as np
    seq = b"ACGTGCTCAGAGACTCAGAGGA"
    ps = sm.PackedSeqVec.from_ascii(seq)
    want = oracle.pack_ascii(seq)
    assert np.array_equal(ps.data[: (len(seq) + 3) // 4], want[: (len(seq) + 3) // 4])
    rc = ps.to_revcomp()
    want_rc = oracle.revcomp_packed(want, len(seq))
    assert np.array_equal(rc.data[: (len(seq) + 3) // 4], want_rc[: (len(seq) + 3) // 4])
    sl = ps.slice(3, 17)
    assert list(sl.codes()) == list(ps.codes()[3:17])
