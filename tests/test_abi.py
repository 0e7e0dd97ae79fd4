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


def _plan(sm, w, canonical, mode, n_windows_list, batch):
    L = sm.lib()
    arr = (C.c_uint64 * len(n_windows_list))(*n_windows_list)
    out7 = (C.c_uint64 * 7)()
    if not batch:
        assert L.mm_debug_launch_plan(w, int(canonical), mode, 0, arr, out7, None, None, None, 0, None) == 0
        return [int(x) for x in out7], None
    cap = 1 << 22
    seq, win0, nblk = (C.c_uint32 * cap)(), (C.c_uint32 * cap)(), (C.c_uint32 * cap)()
    nt = C.c_uint64()
    assert L.mm_debug_launch_plan(w, int(canonical), mode, len(n_windows_list), arr, out7, seq, win0, nblk, cap, C.byref(nt)) == 0
    n = int(nt.value)
    return [int(x) for x in out7], (list(seq[:n]), list(win0[:n]), list(nblk[:n]))


def test_launch_plan_tiles_every_window(sm, monkeypatch):
    """Host logic of the fused kernel's launch (DESIGN.md 4.1 "Launch geometry"), no GPU: with MM_TAPER_SLOTS naming the
    chip's workgroup slots the planner needs no device.  Single sequences: the kernel's closed form for a tapered
    tile's lane length and first window (mm_fused_impl.h, restated here) must tile the window range exactly - no gap, no
    overlap, the last tile reaching the end.  Batches: the tile table must do the same for every sequence."""
    import random
    rng = random.Random(5)
    checked_tapered = 0
    for it in range(400):
        w = rng.choice([1, 3, 5, 11, 17, 25, 33, 51])
        canonical = rng.random() < 0.5
        mode = rng.choice([0, 0, 1])
        slots = rng.choice([1, 2, 3, 7, 64, 768, 1024, 1792])
        monkeypatch.setenv("MM_TAPER_SLOTS", str(slots))
        n = rng.choice([rng.randrange(1, 10**5), rng.randrange(10**5, 10**8), rng.randrange(10**8, 4 * 10**9)])
        o, _ = _plan(sm, w, canonical, mode, [n], False)
        nblk, tiles, first, per_level, min_nblk, start, blk_w = o
        assert blk_w == 256 * w and nblk >= 1
        NB = nblk * blk_w
        if first == 0xFFFFFFFF:
            assert tiles == -(-n // NB)
            continue
        checked_tapered += 1
        lmax = nblk - min_nblk
        assert first * NB == start and tiles > first and lmax >= 1 and per_level >= 1

        def tile(bid):  # the kernel's arithmetic (fused_kernel, "Lane length of THIS tile")
            if bid < first:
                return bid * NB, nblk
            j = bid - first
            lv = min(1 + j // per_level, lmax)
            nb = nblk - lv
            before = per_level * ((lv - 1) * nblk - (lv - 1) * lv // 2) + (j - (lv - 1) * per_level) * nb
            return start + before * blk_w, nb
        end = start
        for bid in range(first, tiles):
            off, nb = tile(bid)
            assert off == end and min_nblk <= nb < nblk, (it, bid)
            end = off + nb * blk_w
        assert end >= n and end - tile(tiles - 1)[1] * blk_w < n, (it, end, n)
    assert checked_tapered > 100
    # batches: every sequence tiled exactly, lanes never longer than the table says, the tail of the launch tapered
    tapered_batches = 0
    for it in range(150):
        w = rng.choice([3, 11, 17, 51])
        slots = rng.choice([1, 2, 5, 64, 768])
        monkeypatch.setenv("MM_TAPER_SLOTS", str(slots))
        n_seqs = rng.randrange(1, 30)
        scale = rng.choice([10**4, 10**6, 10**8])
        nws = [rng.choice([0, rng.randrange(1, scale)]) if rng.random() < 0.9 else 0 for _ in range(n_seqs)]
        if sum(nws) == 0:
            continue
        o, (seq, win0, nb) = _plan(sm, w, True, 0, nws, True)
        longest, blk_w = o[0], 256 * w
        pos = {}
        for s, w0, b in zip(seq, win0, nb):
            assert 1 <= b <= longest and w0 == pos.get(s, 0) and w0 < nws[s], (it, s)
            pos[s] = min(nws[s], w0 + b * blk_w)
        for s, nw in enumerate(nws):
            assert pos.get(s, 0) == nw, (it, s)
        assert seq == sorted(seq)  # input order
        if any(b < longest for b in nb):
            tapered_batches += 1
            k = next(i for i, b in enumerate(nb) if b < longest)
            assert all(x >= y for x, y in zip(nb[k:], nb[k + 1:])), it  # lanes only shrink from there on
    assert tapered_batches > 20
    monkeypatch.delenv("MM_TAPER_SLOTS")


def test_overread_bound_covers_every_launch_plan(sm):
    """the bytes a launch may touch behind its last window are the launcher's own bound
    (mm_fused_overread_bytes: longest lane + two load groups), not a literal in the residency check.  Every launch plan
    the planner produces - small and large windows, super-k-mer indices, long and short runs - keeps its lanes within it."""
    import ctypes as C
    L = sm.lib()
    bound = int(L.mm_fused_overread_bytes())
    assert 15000 < bound < 32768
    out7 = (C.c_uint64 * 7)()
    n_tiles = C.c_uint64()
    os.environ["MM_TAPER_SLOTS"] = "1024"
    try:
        for w in (1, 5, 11, 17, 33, 51, 100, 128):
            for canonical in (0, 1):
                for mode in (0, 1):
                    for nw in (10_000, 3_000_000, 400_000_000, 3_099_999_970):
                        nws = (C.c_uint64 * 1)(nw)
                        sm._check(L.mm_debug_launch_plan(w, canonical, mode, 0, nws, out7, None, None, None, 0, C.byref(n_tiles)))
                        lane_windows = int(out7[0]) * w
                        # a lane's bases + the largest window + two load groups of 8 blocks, 4 bases per byte, + a wide load
                        assert (lane_windows + 2 * 8 * w + 3) // 4 + 20 <= bound, (w, canonical, mode, nw, lane_windows)
    finally:
        del os.environ["MM_TAPER_SLOTS"]


def test_one_round_rule_keeps_every_lane_bound(sm):
    """ADVICE r4 (high), on the CPU: the one-round launch rule (0.6 .. 1 round of the chip's workgroup slots -> one tile
    per slot) used to apply its floor of 6 blocks per lane AFTER the bounds geometry() had applied - with super-k-mer
    indices and w >= 86 the 16-bit list entry (window << shift) + offset allows at most 5 blocks, and the lanes came out
    longer than their entries can address.  Every plan - with and without the rule applying - keeps S << shift <= 65536
    with super-k-mer indices, S + w <= 255 for the 8-bit lists of small forward windows, S <= 60 000 always."""
    import ctypes as C
    L = sm.lib()
    out7 = (C.c_uint64 * 7)()
    n_tiles = C.c_uint64()
    checked = applied = 0
    try:
        for slots in (256, 512, 768, 1024, 1792):
            os.environ["MM_TAPER_SLOTS"] = str(slots)
            for w in (3, 11, 13, 31, 51, 64, 86, 100, 127, 128):
                shift = max(1, w.bit_length())
                for mode in (0, 3):
                    for canonical in (0, 1):
                        nws0 = (C.c_uint64 * 1)(10**10)  # (a long run: the default lanes)
                        sm._check(L.mm_debug_launch_plan(w, canonical, mode, 0, nws0, out7, None, None, None, 0, C.byref(n_tiles)))
                        default_nblk = int(out7[0])
                        for frac in (0.3, 0.55, 0.61, 0.8, 0.99, 1.0, 1.01, 1.7):
                            nw = int(frac * slots * default_nblk * w * 256)
                            if nw < 1:
                                continue
                            nws = (C.c_uint64 * 1)(nw)
                            sm._check(L.mm_debug_launch_plan(w, canonical, mode, 0, nws, out7, None, None, None, 0, C.byref(n_tiles)))
                            nblk, tiles = int(out7[0]), int(out7[1])
                            S = nblk * w
                            assert nblk >= 1 and S <= 60000, (w, mode, slots, frac, nblk)
                            if mode == 3:
                                assert (S << shift) <= 65536, (w, slots, frac, nblk, shift)
                            elif not canonical and w <= 13:
                                assert S + w <= 255, (w, slots, frac, nblk)
                            # uniform tiles cover the range (tapered plans are checked tile by tile in the test above)
                            if int(out7[2]) == 0xffffffff:
                                assert tiles * nblk * w * 256 >= nw > (tiles - 1) * nblk * w * 256, (w, mode, slots, frac)
                            if 0.6 <= frac <= 1.0 and tiles == slots:
                                applied += 1
                            checked += 1
    finally:
        del os.environ["MM_TAPER_SLOTS"]
    assert checked > 1000 and applied > 100, (checked, applied)


def test_kernel_source_compiles_with_hiprtc(tmp_path):
    """Window sizes without a prebuilt kernel are compiled at first use by hiprtc from the kernel's source (mm_jit.hip).
    hiprtc cross-compiles without a device, so the CPU suite can catch what only it rejects - round 5: an inline-assembly
    "n" constraint whose operand is a constant only after unrolling compiled under hipcc and failed under hiprtc, and the
    run quietly took the generic family.  Two instances that exercise every walk: w = 100 (no prebuilt kernel, beyond the
    LDS landing's range) and w = 40 (inside it)."""
    import shutil
    import subprocess
    if not shutil.which("g++") or not os.path.exists("/opt/rocm/lib/libhiprtc.so"):
        pytest.skip("no g++ / libhiprtc")
    probe_dir = os.path.join(ROOT, "tools", "jit_probe")
    exe = str(tmp_path / "probe.bin")
    subprocess.run(["g++", "-O1", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-o", exe, os.path.join(probe_dir, "probe.cpp"),
                    "-L/opt/rocm/lib", "-lhiprtc", "-Wl,-rpath,/opt/rocm/lib"], check=True, capture_output=True)
    for name in ("mm::fused_kernel<100, true, true, 0, false, false>", "mm::fused_kernel<40, true, true, 0, false, true>"):
        r = subprocess.run([exe, name], cwd=probe_dir, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "HIPRTC_SUCCESS" in r.stdout, (name, (r.stdout + r.stderr)[-1500:])


def test_skip_ambiguous_launches_fit_the_cu(sm):
    """Round 5, no GPU: a skip-ambiguous launch carries the landing area of its look-ahead loads in front of the lane lists
    (12 KB for 32 <= w <= 35, 10 KB for w = 36, 37, 13 KB for 38 <= w <= 54, 18 KB above, 0 elsewhere).  Lists + landing + the static tables have to
    fit the CU's 160 KB as many times as the kernel's register bound lets workgroups share it (4 up to w = 37, 3 up to
    w = 54, 2 up to w = 64) - one workgroup fewer is a quarter to a third of the walk's speed - and the window sizes with
    chunked window bits keep the default lanes of the plain walk (a genome's clean waves walk those lanes)."""
    import ctypes as C
    L = sm.lib()
    out2, out7, out7p = (C.c_uint64 * 2)(), (C.c_uint64 * 7)(), (C.c_uint64 * 7)()
    nw = (C.c_uint64 * 1)(3 * 10**9)
    checked = 0
    for w in list(range(13, 66)) + [81, 96, 100, 128]:
        sm._check(L.mm_debug_launch_lds(w, 1, 4, nw[0], out2))
        lists, landing = int(out2[0]), int(out2[1])
        sm._check(L.mm_debug_launch_lds(w, 1, 0, nw[0], out2))
        assert int(out2[1]) == 0                                     # (no ambiguity bits: no landing area)
        sm._check(L.mm_debug_launch_plan(w, 1, 4, 0, nw, out7, None, None, None, 0, None))
        sm._check(L.mm_debug_launch_plan(w, 1, 0, 0, nw, out7p, None, None, None, 0, None))
        nblk, nblk_plain = int(out7[0]), int(out7p[0])
        want_landing = 0 if (w < 32 or w > 96) else (4 * 3072 if w < 36 else (4 * 2560 if w < 38 else (4 * 3328 if w <= 54 else 4 * 4608)))
        if 32 <= w <= 96 and landing == 0:
            continue                                                 # (a window size whose loads are not grouped: no landing)
        assert landing == want_landing, (w, landing)
        per_cu = 4 if w <= 37 else (3 if w <= 54 else (2 if w <= 64 else 1))
        granule = 1280                                               # (the CU's 160 KB are handed out in units of 1280 bytes)
        assert -(-(lists + landing + 512) // granule) * granule * per_cu <= 160 * 1024, (w, lists, landing, per_cu)
        if 21 <= w <= 35:
            assert nblk * w <= 400 or nblk == 6, (w, nblk)          # (the short lanes of the middle window sizes)
        elif w in (36, 37):
            assert 12 <= nblk <= nblk_plain, (w, nblk, nblk_plain)   # (chunked window bits beside lists that fit four times per CU)
        else:
            assert nblk == nblk_plain, (w, nblk, nblk_plain)
        checked += 1
    assert checked >= 50


def test_lane_plan_keeps_every_bound(sm):
    """The lane-table launch's plan (round 6; mm_debug_lane_plan, no device): for every window size up to 128 and every
    flavour - minimizers, closed / open syncmers, super-k-mer indices, PackedNSeq - the lists (and the skip-ambiguous walk's
    landing area) fit a CU's LDS, a lane's element positions fit 16 bits (and the packed (window, offset) entry of the
    super-k-mer flavour), the list holds the expected entries with head-room, and the lane bound covers every read owning a
    lane plus ceil(windows / S) lanes per read."""
    import ctypes as C
    L = sm.lib()
    out = (C.c_uint64 * 6)()
    lds2 = (C.c_uint64 * 2)()
    checked = 0
    for w in range(1, 129):
        for canonical in (0, 1):
            for mode in (0, 1, 2, 3, 4):
                if mode == 2 and w % 2 == 0:
                    continue
                if mode == 4 and not canonical:
                    continue
                k = 21 if (21 + w - 1) % 2 == 1 else 22
                for (n_reads, total, nb) in ((200_000, 2_500_000_000, 0), (1, 50_000, 0), (8_000_000, 1_200_000_000, 0), (1000, 10_000_000, 3)):
                    assert L.mm_debug_lane_plan(k, w, canonical, mode, n_reads, total, nb, out) == 0, (w, canonical, mode)
                    nblk, S, cap, lds, lanes_cap, tiles = (int(x) for x in out)
                    assert S == w * nblk and nblk >= 1 and S + w <= 60_000
                    land = 0
                    if mode == 4:
                        assert L.mm_debug_launch_lds(w, canonical, 4, 10**9, lds2) == 0
                        land = int(lds2[1])
                    assert lds == cap * 516 and lds + land <= 159 * 1024, (w, mode, lds, land)
                    if mode == 3:
                        shift = max(1, w.bit_length())
                        assert (S << shift) <= 65536, (w, S, shift)
                    dens = 1.0 / w if mode == 2 else (2.0 / w if mode == 1 else 2.0 / (w + 1))
                    assert cap >= min(S + w, int(1.3 * dens * S) + 8), (w, mode, cap, S)
                    assert lanes_cap % 256 == 0 and tiles * 256 == lanes_cap
                    assert lanes_cap >= n_reads + total // S, (w, mode, lanes_cap)
                    if nb:
                        assert nblk <= nb
                    checked += 1
    assert checked > 4000
