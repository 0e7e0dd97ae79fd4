"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, bit-exact.

Mirrors the reference's test strategy (src/test.rs): known-answer vectors, a (k, w, len,
offset) sweep with naive == product on packed and ASCII input, super-k-mer indices, syncmers,
the reverse-complement metamorphic test, k-mer values — plus large-size runs and the
window-range sharding property the multi-GPU path relies on.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _packed(sm, oracle, n, seed):
    data = oracle.gen_packed(seed, n)
    return sm.PackedSeq(data, 0, n)


def _builder(sm, k, w, canonical, mode):
    return sm.Builder(k, w, canonical, mode)


# ------------------------------------------------------------- known answers
def test_reference_known_answers(sm, oracle, gpu):
    ref = json.load(open(os.path.join(GOLD, "reference_vectors.json")))
    for case in ref["minimizers"]:
        seq = case["seq"].encode()
        ps = sm.PackedSeqVec.from_ascii(seq)
        if case.get("revcomp_input"):
            ps = ps.to_revcomp()
        b = _builder(sm, case["k"], case["w"], case["canonical"], case["mode"])
        for force_generic in (False, True):
            gpu.force_generic(force_generic)
            pos: list = []
            out = b.run(ps, pos)
            assert pos == case["positions"], (case["source"], force_generic)
            if "values_u64" in case:
                assert list(map(int, out.values_u64())) == case["values_u64"]
            if not case.get("revcomp_input"):
                assert b.run_once(sm.AsciiSeq(seq)) == case["positions"]
        gpu.force_generic(False)


def test_closed_syncmer_values_all_g(sm, gpu):
    # src/test.rs:576-597
    n = 100
    ps = sm.PackedSeqVec.from_ascii(b"G" * n)
    for k in range(1, 10):
        for w in range(1, 10):
            pos: list = []
            out = sm.closed_syncmers(k, w).run(ps, pos)
            vals = out.values_u64()
            l = k + w - 1
            assert len(vals) == n - l + 1
            assert all(int(v) == (1 << (2 * l)) - 1 for v in vals)


def test_model_anchors_1mbp(sm, oracle, gpu):
    anchors = json.load(open(os.path.join(GOLD, "model_anchors.json")))
    n, seed = anchors["generator"]["n"], anchors["generator"]["seed"]
    ps = _packed(sm, oracle, n, seed)
    for c in anchors["cases"]:
        b = _builder(sm, c["k"], c["w"], c["canonical"], c["mode"])
        pos, _ = b._run_arrays(ps)
        assert len(pos) == c["count"], c["name"]
        assert [int(x) for x in pos[:8]] == c["first8"], c["name"]
        assert oracle.checksum(pos) == (c["checksum_weighted"], c["checksum_plain"]), c["name"]


# ------------------------------------------------------------------- sweeps
# Window sizes with a prebuilt fused-kernel instance come from the LIBRARY (mm_prebuilt_window_sizes), so that an
# instance cannot ship without being compared with the oracle.  The sizes below get the full
# list of k; the others a shorter one (every size still runs every length and slice offset of the sweep).
DEEP_W = {True: [1, 2, 3, 4, 5, 7, 8, 11, 12, 16, 17, 19, 31, 33, 41, 51],
          False: [1, 2, 3, 4, 5, 7, 8, 11, 13, 16, 17, 19, 31, 33, 41, 51]}


def _sweep_inputs(rng, oracle, sm):
    data = oracle.gen_packed(int(rng.integers(1 << 30)), 8192 + 8)
    lens = list(range(0, 100, 7)) + [int(x) for x in rng.integers(100, 8192, size=4)]
    for ln in lens:
        off = int(rng.integers(0, min(3, ln) + 1))
        yield sm.PackedSeq(data, off, ln - off), data, off, ln - off


@pytest.mark.parametrize("canonical", [False, True])
@pytest.mark.parametrize("force_generic", [False, True])
def test_sweep_minimizers(sm, oracle, gpu, canonical, force_generic):
    """src/test.rs:53-110: naive definition == product for every (k, w, len, slice offset)."""
    rng = np.random.default_rng(1234 + canonical)
    ks = [1, 2, 3, 4, 5, 21, 31, 32, 33, 63, 64, 65] + [int(x) for x in rng.integers(6, 100, size=3)]
    ws = [1, 2, 3, 4, 5, 31, 32, 33, 63, 64, 65] + [int(x) for x in rng.integers(6, 100, size=3)]
    if not force_generic:
        ws = sm.prebuilt_window_sizes(canonical)
        assert set(DEEP_W[canonical]) <= set(ws) and len(ws) >= 35
    gpu.force_generic(force_generic)
    try:
        for k in ks:
            for w in ws:
                if canonical and (k + w - 1) % 2 == 0:
                    continue
                if not force_generic and w not in DEEP_W[canonical] and k not in (1, 2, 5, 21, 32, 64, 65):
                    continue
                b = _builder(sm, k, w, canonical, 0)
                for ps, data, off, n in _sweep_inputs(rng, oracle, sm):
                    want = oracle.run(data, n, k, w, canonical=canonical, flavour=oracle.NAIVE,
                                      base_offset=off)
                    got, _ = b._run_arrays(ps)
                    assert np.array_equal(got, want), f"k={k} w={w} n={n} off={off} generic={force_generic}"
                    if n >= k + w - 1:
                        assert gpu.last_path() == (sm.PATH_GENERIC if force_generic else sm.PATH_FUSED)
    finally:
        gpu.force_generic(False)


@pytest.mark.parametrize("canonical", [False, True])
def test_sweep_superkmers(sm, oracle, gpu, canonical):
    """src/test.rs:154-277: positions and super-k-mer start indices."""
    rng = np.random.default_rng(99 + canonical)
    for force_generic in (False, True):
        gpu.force_generic(force_generic)
        try:
            for k, w in [(5, 7), (21, 11), (31, 19), (4, 5), (15, 17)]:
                if canonical and (k + w - 1) % 2 == 0:
                    continue
                for ps, data, off, n in _sweep_inputs(rng, oracle, sm):
                    wp, wsk = oracle.run(data, n, k, w, canonical=canonical, base_offset=off,
                                         super_kmers=True)
                    sk: list = []
                    b = _builder(sm, k, w, canonical, 0).super_kmers(sk)
                    pos = b.run_once(ps)
                    assert pos == list(map(int, wp)) and sk == list(map(int, wsk)), (k, w, n, off)
        finally:
            gpu.force_generic(False)


@pytest.mark.parametrize("canonical", [False, True])
@pytest.mark.parametrize("mode", [1, 2])
def test_sweep_syncmers(sm, oracle, gpu, canonical, mode):
    """src/test.rs:517-574, 599-640: syncmer window indices, open and closed."""
    rng = np.random.default_rng(7 + canonical + 2 * mode)
    for force_generic in (False, True):
        gpu.force_generic(force_generic)
        try:
            for k, w in [(5, 7), (15, 17), (3, 5), (21, 11), (8, 19), (2, 33)]:
                if canonical and (k + w - 1) % 2 == 0:
                    continue
                if mode == 2 and w % 2 == 0:
                    continue
                b = _builder(sm, k, w, canonical, mode)
                for ps, data, off, n in _sweep_inputs(rng, oracle, sm):
                    want = oracle.run(data, n, k, w, canonical=canonical, mode=mode,
                                      flavour=oracle.NAIVE, base_offset=off)
                    got, _ = b._run_arrays(ps)
                    assert np.array_equal(got, want), (k, w, n, off, force_generic)
        finally:
            gpu.force_generic(False)


def test_revcomp_metamorphic(sm, oracle, gpu):
    """src/test.rs:112-152: pos_fwd[i] + pos_rc[-1-i] == len - k and equal values."""
    rng = np.random.default_rng(5)
    for k, w in [(5, 7), (21, 11), (31, 51), (15, 17)]:
        for n in (0, 50, 200, 4097):
            ps = sm.PackedSeqVec.random(n, seed=int(rng.integers(1 << 30)))
            rc = ps.to_revcomp()
            b = sm.canonical_minimizers(k, w)
            fp: list = []
            rp: list = []
            fv = b.run(ps, fp).values_u64() if k <= 32 else None
            rv = b.run(rc, rp).values_u64() if k <= 32 else None
            assert len(fp) == len(rp)
            for x, y in zip(fp, reversed(rp)):
                assert x + y == n - k
            if fv is not None:
                assert np.array_equal(fv, rv[::-1])


@pytest.mark.parametrize("mode", [1, 2])
def test_revcomp_metamorphic_syncmers(sm, oracle, gpu, mode):
    """src/test.rs:642-711 (canonical_syncmers_positions_and_values): canonical closed / open syncmers
    of a sequence and of its reverse complement mirror each other: pos_fwd[i] + pos_rc[-1-i] ==
    len - (k+w-1), and the l-mer values are equal in reverse order (l <= 32)."""
    rng = np.random.default_rng(6)
    for k, w in [(5, 7), (3, 5), (15, 17), (11, 11), (9, 23), (1, 1), (25, 7)]:
        l = k + w - 1
        assert l % 2 == 1 and l <= 32 and w % 2 == 1
        for n in (0, 31, 50, 200, 4097):
            ps = sm.PackedSeqVec.random(n, seed=int(rng.integers(1 << 30)))
            rc = ps.to_revcomp()
            b = sm.Builder(k, w, True, mode)
            fp: list = []
            rp: list = []
            fv = b.run(ps, fp).values_u64()
            rv = b.run(rc, rp).values_u64()
            assert len(fp) == len(rp), (k, w, n)
            for x, y in zip(fp, reversed(rp)):
                assert x + y == n - l, (k, w, n, x, y)
            assert np.array_equal(fv, rv[::-1])


def test_append_semantics(sm, gpu):
    """`run` appends; a leading duplicate of out_vec.last() is dropped (src/collect.rs:265-271)."""
    ps = sm.PackedSeqVec.from_ascii(b"ACGTGCTCAGAGACTCAGAGGA")
    out = [123, 0]
    sm.canonical_minimizers(5, 7).run(ps, out)
    assert out == [123, 0, 7, 9, 15]


# ---------------------------------------------------------------- large sizes
@pytest.mark.parametrize("k,w,canonical,mode", [(21, 11, False, 0), (21, 11, True, 0), (31, 51, True, 0),
                                                (15, 17, True, 1), (5, 7, False, 0)])
def test_large_device_vs_oracle(sm, oracle, gpu, k, w, canonical, mode):
    """16 Mbp device-resident run, every position compared with the streaming oracle."""
    import torch
    n = 16_000_003
    data = oracle.gen_packed(11, n)
    want = oracle.run(data, n, k, w, canonical=canonical, mode=mode)
    d = torch.from_numpy(data).cuda()
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    b = _builder(sm, k, w, canonical, mode)
    cnt = b.run_device(d, n, out)
    assert gpu.last_path() == sm.PATH_FUSED
    got = out[:cnt].cpu().numpy().view(np.uint32)
    assert cnt == len(want)
    assert np.array_equal(got, want)
    # the generic family must agree on the device too
    gpu.force_generic(True)
    try:
        out.zero_()
        cnt2 = b.run_device(d, n, out)
        assert cnt2 == cnt and np.array_equal(out[:cnt2].cpu().numpy().view(np.uint32), want)
    finally:
        gpu.force_generic(False)


def test_window_range_sharding(sm, oracle, gpu):
    """Concatenating window-range shards reproduces the whole-sequence output exactly
    (this is what the multi-GPU path does; seam rule of src/collect.rs:265-271)."""
    import torch
    n, k, w = 3_000_017, 21, 11
    data = oracle.gen_packed(21, n)
    d = torch.from_numpy(data).cuda()
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    for canonical, mode in [(True, 0), (False, 0), (True, 1)]:
        b = _builder(sm, k, w, canonical, mode)
        whole = out[:b.run_device(d, n, out)].cpu().numpy().copy()
        nw = n - (k + w - 1) + 1
        cuts = [0, 1, 1000, nw // 3 + 5, nw // 2, nw - 1, nw]
        parts = []
        for a, e in zip(cuts[:-1], cuts[1:]):
            c = b.run_device(d, n, out, win_begin=a, win_end=e)
            parts.append(out[:c].cpu().numpy().copy())
        assert np.array_equal(np.concatenate(parts), whole), (canonical, mode)


def test_unaligned_device_pointer(sm, oracle, gpu):
    """Device buffers at any byte alignment and base offset (PackedSeq slices, src/test.rs:42-45)."""
    import torch
    n, k, w = 100_003, 21, 11
    data = oracle.gen_packed(3, n + 64)
    d = torch.from_numpy(data).cuda()
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    for byte_off in (0, 1, 2, 3, 5):
        for base_off in (0, 1, 3, 7):
            nb = n - 4 * byte_off - base_off
            want = oracle.run(data, nb, k, w, canonical=True, base_offset=4 * byte_off + base_off)
            c = sm.canonical_minimizers(k, w).run_device(d[byte_off:], nb, out, base_offset=base_off)
            assert np.array_equal(out[:c].cpu().numpy().view(np.uint32), want), (byte_off, base_off)


def test_capacity_error(sm, oracle, gpu):
    import torch
    n = 100_000
    d = torch.from_numpy(oracle.gen_packed(4, n)).cuda()
    out = torch.zeros(10, dtype=torch.int32, device="cuda")
    with pytest.raises(sm.MinimizerError) as e:
        sm.minimizers(21, 11).run_device(d, n, out)
    assert e.value.code == sm.ERR["CAPACITY"]


def test_low_complexity_dense_output(sm, oracle, gpu):
    """Homopolymers / short repeats: every hash ties, the output is dense (one position per
    window for forward minimizers) — exercises the direct-store path of the compaction."""
    import torch
    n = 300_000
    for unit in (b"A", b"AC", b"ACGTT", b"G"):
        seq = (unit * (n // len(unit) + 1))[:n]
        data = oracle.pack_ascii(seq)
        d = torch.from_numpy(data).cuda()
        out = torch.zeros(n, dtype=torch.int32, device="cuda")
        for k, w, canonical, mode in [(21, 11, False, 0), (21, 11, True, 0), (15, 17, True, 1), (5, 7, False, 2)]:
            want = oracle.run(data, n, k, w, canonical=canonical, mode=mode)
            c = _builder(sm, k, w, canonical, mode).run_device(d, n, out)
            assert c == len(want) and np.array_equal(out[:c].cpu().numpy().view(np.uint32), want), (unit, k, w)


def test_mixed_density_lists_with_superkmers(sm, oracle, gpu):
    """Sequences whose lanes emit between a few and all of their windows: lane lists shorter than a
    wave, longer than a wave (second copy-out loop) and overflowing (direct redo) in the same run —
    positions and super-k-mer indices (packed list entries) against the oracle."""
    import torch
    rng = np.random.default_rng(5)
    n = 1_200_000
    codes = rng.integers(0, 4, size=n).astype(np.uint8)
    pos = 0
    while pos < n:  # homopolymer stretches of growing length between random stretches
        run = int(rng.integers(20, 400))
        codes[pos:pos + run] = codes[pos]
        pos += run + int(rng.integers(100, 900))
    seq = np.frombuffer(b"ACTG", dtype=np.uint8)[codes].tobytes()
    data = oracle.pack_ascii(seq)
    d = torch.from_numpy(data).cuda()
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    sk = torch.zeros(n, dtype=torch.int32, device="cuda")
    for k, w, canonical in [(21, 11, True), (21, 11, False), (9, 5, True), (31, 51, True), (12, 18, True)]:
        want, wsk = oracle.run(data, n, k, w, canonical=canonical, super_kmers=True)
        b = sm.Builder(k, w, canonical, 0)
        c = b.run_device(d, n, out)
        assert c == len(want) and np.array_equal(out[:c].cpu().numpy().view(np.uint32), want), (k, w, "pos")
        c = b.run_device(d, n, out, out_sk=sk)
        assert gpu.last_path() == sm.PATH_FUSED
        assert c == len(want) and np.array_equal(out[:c].cpu().numpy().view(np.uint32), want), (k, w, "pos+sk")
        assert np.array_equal(sk[:c].cpu().numpy().view(np.uint32), wsk), (k, w, "sk")
    # the same through reads mode (reads of 400 bases cut from the sequence)
    n_reads, stride, read_len = 2500, 400, 400
    offs = torch.zeros(n_reads + 1, dtype=torch.int64, device="cuda")
    total = sm.run_reads_device(sm.Builder(21, 11, True, 0), d, n_reads, stride, read_len, out, offs, out_sk=sk)
    ho = offs.cpu().numpy()
    hp, hs = out[:total].cpu().numpy().view(np.uint32), sk[:total].cpu().numpy().view(np.uint32)
    for r in range(0, n_reads, 7):
        want, wsk = oracle.run(data, read_len, 21, 11, canonical=True, base_offset=r * stride, super_kmers=True)
        assert np.array_equal(hp[ho[r]:ho[r + 1]], want) and np.array_equal(hs[ho[r]:ho[r + 1]], wsk), r


def test_register_bounded_canonical_kernels(sm, oracle, gpu):
    """Canonical kernels for w = 19..47 are compiled under a register bound (a few spilled registers,
    one more wave per SIMD): positions, syncmers and super-k-mer indices against the oracle on a
    multi-tile input, prebuilt (25, 31, 33, 41) and run-time specialised (45) window sizes."""
    import torch
    n = 1_500_003
    data = oracle.gen_packed(78, n)
    d = torch.from_numpy(data).cuda()
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    sk = torch.zeros_like(out)
    for w, modes in ((25, (0, 1, 2)), (31, (0,)), (33, (0, 2)), (41, (0, 1)), (45, (0,))):
        k = 21
        for mode in modes:
            want = oracle.run(data, n, k, w, canonical=True, mode=mode)
            c = sm.Builder(k, w, True, mode).run_device(d, n, out)
            assert gpu.last_path() == sm.PATH_FUSED
            assert c == len(want) and np.array_equal(out[:c].cpu().numpy().view(np.uint32), want), (w, mode)
        want, wsk = oracle.run(data, n, k, w, canonical=True, super_kmers=True)
        c = sm.Builder(k, w, True, 0).run_device(d, n, out, out_sk=sk)
        assert c == len(want) and np.array_equal(out[:c].cpu().numpy().view(np.uint32), want), w
        assert np.array_equal(sk[:c].cpu().numpy().view(np.uint32), wsk), w


def test_custom_hasher_tables(sm, oracle, gpu):
    """.hasher(&h) (src/lib.rs:327): tables cross the ABI as data (seeded hashers)."""
    rng = np.random.default_rng(77)
    n = 50_000
    data = oracle.gen_packed(8, n)
    ps = sm.PackedSeq(data, 0, n)
    for canonical in (False, True):
        fw = [int(x) for x in rng.integers(0, 1 << 32, size=4)]
        rc = [fw[c ^ 2] for c in range(4)]
        for rot in (7, 1, 13):
            h = sm.Hasher.from_tables(fw, rc, rot, canonical)
            oh = oracle.Hasher()
            for i in range(4):
                oh.fw[i], oh.rc[i] = fw[i], rc[i]
            oh.rot, oh.canonical = rot, int(canonical)
            for k, w in [(21, 11), (5, 7), (31, 19)]:
                if canonical and (k + w - 1) % 2 == 0:
                    continue
                want = oracle.run(data, n, k, w, hasher=oh, canonical=canonical)
                got, _ = sm.Builder(k, w, canonical, 0, hasher=h)._run_arrays(ps)
                assert np.array_equal(got, want), (canonical, rot, k, w)
    # forward windows with a canonical hasher (allowed by the reference, generic kernel family)
    h = sm.NtHasher(21, canonical=True)
    want = oracle.run(data, n, 21, 11, hasher=oracle.default_hasher(True), canonical=False)
    got, _ = sm.minimizers(21, 11).hasher(h)._run_arrays(ps)
    assert np.array_equal(got, want)


def test_ticket_mode_matches(sm, oracle, gpu, monkeypatch):
    """Safe mode (tile ids from an atomic ticket) produces the same output."""
    import torch
    n, k, w = 5_000_011, 21, 11
    data = oracle.gen_packed(31, n)
    want = oracle.run(data, n, k, w, canonical=True)
    d = torch.from_numpy(data).cuda()
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    monkeypatch.setenv("MM_FORCE_TICKET", "1")
    c = sm.canonical_minimizers(k, w).run_device(d, n, out)
    assert c == len(want) and np.array_equal(out[:c].cpu().numpy().view(np.uint32), want)


def test_blocks_per_lane_knob(sm, oracle, gpu):
    """Any legal tile geometry gives the same result."""
    import torch
    n, k, w = 2_000_003, 21, 11
    data = oracle.gen_packed(41, n)
    want = oracle.run(data, n, k, w, canonical=True)
    d = torch.from_numpy(data).cuda()
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    try:
        for nblk in (1, 16, 32, 48):
            gpu.set_blocks_per_lane(nblk)
            c = sm.canonical_minimizers(k, w).run_device(d, n, out)
            assert c == len(want) and np.array_equal(out[:c].cpu().numpy().view(np.uint32), want), nblk
        for nblk, (k2, w2) in ((1, (31, 51)), (3, (31, 51)), (5, (15, 19))):
            gpu.set_blocks_per_lane(nblk)
            want2 = oracle.run(data, n, k2, w2, canonical=True)
            c = sm.canonical_minimizers(k2, w2).run_device(d, n, out)
            assert c == len(want2) and np.array_equal(out[:c].cpu().numpy().view(np.uint32), want2), (nblk, k2, w2)
        # super-k-mer runs pack (window, offset) into one 16-bit list entry, which bounds the lane
        # length: over-long requests are shortened by the launcher, results stay the same
        sk = torch.zeros(n, dtype=torch.int32, device="cuda")
        for nblk, (k2, w2) in ((500, (21, 11)), (372, (21, 11)), (30, (31, 51)), (1, (21, 11)), (700, (9, 7))):
            gpu.set_blocks_per_lane(nblk)
            want2, wsk2 = oracle.run(data, n, k2, w2, canonical=True, super_kmers=True)
            c = sm.canonical_minimizers(k2, w2).run_device(d, n, out, out_sk=sk)
            assert gpu.last_path() == sm.PATH_FUSED
            assert c == len(want2) and np.array_equal(out[:c].cpu().numpy().view(np.uint32), want2), (nblk, k2, w2)
            assert np.array_equal(sk[:c].cpu().numpy().view(np.uint32), wsk2), (nblk, k2, w2)
    finally:
        gpu.set_blocks_per_lane(0)


def test_cxx_builder_example(gpu):
    """The header-only C++ mirror of the builder reproduces the reference doctests."""
    import subprocess
    exe = os.path.join(os.path.dirname(__file__), "cxx", "builder_example")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.dirname(exe)], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)


def test_device_values_and_generator(sm, oracle, gpu):
    """Output::values_u64 on device-resident positions; device generator == host generator."""
    import ctypes as C
    import torch
    n = 1_000_000
    d = sm.generate_device(n, 1)
    host = oracle.gen_packed(1, n)
    assert np.array_equal(d[: (n + 3) // 4].cpu().numpy(), host[: (n + 3) // 4])
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    for k, w, canonical, mode in [(21, 11, True, 0), (15, 17, True, 1), (31, 5, False, 0)]:
        b = _builder(sm, k, w, canonical, mode)
        c = b.run_device(d, n, out)
        vals = torch.zeros(c, dtype=torch.int64, device="cuda")
        ln = k if mode == 0 else k + w - 1
        sm._check(sm.lib().mm_values_u64_device_async(gpu.h, C.c_void_p(d.data_ptr()), d.numel(), 0, n, ln,
                                                      int(canonical), C.c_void_p(out.data_ptr()), c,
                                                      C.c_void_p(vals.data_ptr())))
        gpu.sync()
        want = oracle.values_u64(host, ln, out[:c].cpu().numpy().view(np.uint32), canonical)
        assert np.array_equal(vals.cpu().numpy().view(np.uint64), want), (k, w, mode)


def _device_checksum(t, c):
    """(count, sum, order-sensitive weighted sum mod 2^64) of the first c int32 entries."""
    import torch
    v = t[:c].to(torch.int64) & 0xFFFFFFFF
    plain = int(v.sum().item())
    weighted = 0
    step = 1 << 27
    for a in range(0, c, step):
        e = min(c, a + step)
        idx = torch.arange(a + 1, e + 1, dtype=torch.int64, device=t.device)
        weighted = (weighted + int((v[a:e] * idx).sum().item())) & ((1 << 64) - 1)
    return c, plain, weighted


@pytest.mark.parametrize("k,w,mode", [(21, 11, 0), (31, 51, 0), (15, 17, 1)])
def test_full_size_properties(sm, oracle, gpu, k, w, mode):
    """BASELINE's full size (3.1 Gbp; configs 3, 4 and 5 of SURVEY.md §8d: canonical k=21 w=11,
    canonical k=31 w=51, canonical closed syncmers k=15 w=17) through size-independent properties:
    fused kernel == generic kernel family (independent implementation) == concatenation of
    window-range shards, by count and an order-sensitive checksum; densities as expected; and
    the first / last megabase against the oracle."""
    import torch
    n = 3_100_000_000
    d = sm.generate_device(n, 3)
    density = 2.0 / (w + 1) if mode == 0 else 2.0 / w
    cap = int(n * density * 1.15)
    out = torch.empty(cap, dtype=torch.int32, device="cuda")
    b = sm.Builder(k, w, True, mode)
    c = b.run_device(d, n, out)
    assert gpu.last_path() == sm.PATH_FUSED
    assert abs(c / n - density) < 2e-3
    whole = _device_checksum(out, c)
    # head and tail against the oracle
    m = 1_000_000
    head = oracle.run(oracle.gen_packed(3, m + 256), m + 256, k, w, canonical=True, mode=mode)
    head = head[head < m - 256]
    got_head = out[: len(head)].cpu().numpy().view(np.uint32)
    assert np.array_equal(got_head, head)
    tail_start = n - m
    tail = oracle.run(oracle.gen_packed(3, m, first_base=tail_start), m, k, w, canonical=True, mode=mode)
    got_tail = out[c - len(tail) + 50: c].cpu().numpy().view(np.uint32).astype(np.int64) - tail_start
    assert np.array_equal(got_tail, tail[50:].astype(np.int64))
    # monotone-ish: canonical positions never jump back by w or more; syncmer indices increase
    diffs = (out[1:c].to(torch.int64) & 0xFFFFFFFF) - (out[: c - 1].to(torch.int64) & 0xFFFFFFFF)
    assert int(diffs.min().item()) > (-w if mode == 0 else 0)
    del diffs
    # window-range shards
    nw = n - (k + w - 1) + 1
    cuts = [0, nw // 3 + 11, 2 * nw // 3 - 5, nw]
    tot_c, tot_plain, tot_weighted = 0, 0, 0
    for a, e in zip(cuts[:-1], cuts[1:]):
        # a shard dedups against the window before its range, so plain concatenation is exact
        cc = b.run_device(d, n, out, win_begin=a, win_end=e)
        _, plain, weighted = _device_checksum(out, cc)
        # re-base the weighted sum: indices of this shard start at tot_c
        tot_weighted = (tot_weighted + weighted + tot_c * plain) & ((1 << 64) - 1)
        tot_plain += plain
        tot_c += cc
    assert (tot_c, tot_plain, tot_weighted) == whole
    # generic family
    gpu.force_generic(True)
    try:
        cg = b.run_device(d, n, out)
        assert gpu.last_path() == sm.PATH_GENERIC
        assert _device_checksum(out, cg) == whole
    finally:
        gpu.force_generic(False)


def test_values_u128(sm, oracle, gpu):
    """Output::values_u128 (src/lib.rs:587-629) for 32 < len <= 64 and small lens."""
    n = 200_000
    data = oracle.gen_packed(14, n)
    ps = sm.PackedSeq(data, 0, n)
    for k, w, canonical, mode in [(41, 11, True, 0), (64, 5, True, 0), (33, 3, False, 0), (31, 17, True, 1),
                                  (5, 7, True, 0), (32, 1, True, 0)]:
        if canonical and (k + w - 1) % 2 == 0:
            continue
        pos: list = []
        out = sm.Builder(k, w, canonical, mode).run(ps, pos)
        if out.len > 64:
            continue
        got = out.values_u128()
        want = oracle.values_u128(data, out.len, np.array(pos, dtype=np.uint32), canonical)
        assert got == [int(a) | (int(b) << 64) for a, b in want], (k, w, canonical, mode)
        if out.len <= 32:
            assert got == [int(v) for v in out.values_u64()]


def test_batch_of_contigs(sm, oracle, gpu):
    """Independent sequences with one plan (the reference calls run() per contig,
    bench/src/bin/paper.rs:410-431): sequence-local positions, back to back, with offsets."""
    import torch
    lens = [250_001, 17, 30, 31, 99_999, 0, 1_000_003, 64]
    datas = [oracle.gen_packed(100 + i, max(n, 1)) for i, n in enumerate(lens)]
    d = [torch.from_numpy(x).cuda() for x in datas]
    out = torch.zeros(sum(lens) + 16, dtype=torch.int32, device="cuda")
    sk = torch.zeros_like(out)
    # (the forward plans with w <= 13 and no super-k-mer indices run the 8-bit-list flavour of the kernel)
    for k, w, canonical, mode, use_sk in [(21, 11, True, 0, False), (21, 11, False, 0, True), (15, 17, True, 1, False),
                                          (9, 23, False, 0, False), (15, 10, False, 0, False), (12, 9, False, 1, False),
                                          (7, 13, False, 2, False)]:
        b = sm.Builder(k, w, canonical, mode)
        offs = sm.run_batch_device(b, d, lens, out, sk if use_sk else None)
        assert offs[0] == 0 and len(offs) == len(lens) + 1
        host = out[: offs[-1]].cpu().numpy().view(np.uint32)
        hsk = sk[: offs[-1]].cpu().numpy().view(np.uint32)
        for i, n in enumerate(lens):
            if use_sk:
                want, wsk = oracle.run(datas[i], n, k, w, canonical=canonical, mode=mode, super_kmers=True)
                assert np.array_equal(hsk[offs[i]:offs[i + 1]], wsk), (i, k, w)
            else:
                want = oracle.run(datas[i], n, k, w, canonical=canonical, mode=mode)
            assert np.array_equal(host[offs[i]:offs[i + 1]], want), (i, n, k, w)


def test_batch_many_small_contigs(sm, oracle, gpu):
    """A few thousand contigs of 0 .. 6000 bases carved out of one buffer at odd byte and base
    offsets: one launch (tile -> sequence table); every contig equals an independent run; the
    generic family (one launch per sequence) gives the same."""
    import torch
    rng = np.random.default_rng(77)
    n_seq = 1500
    lens = rng.integers(0, 6001, size=n_seq)
    lens[:8] = [0, 1, 30, 31, 32, 6000, 0, 5999]
    total = int(lens.sum()) + 8 * n_seq + 64
    data = oracle.gen_packed(9, total)
    big = torch.from_numpy(data).cuda()
    starts = np.concatenate([[0], np.cumsum(lens + rng.integers(0, 8, size=n_seq))])[:n_seq]  # base positions
    d, offs_b = [], []
    for s0 in starts:
        byte0 = int(s0) // 4
        d.append(big[byte0:])
        offs_b.append(int(s0) % 4)
    out = torch.zeros(int(lens.sum()) + 64, dtype=torch.int32, device="cuda")
    sk = torch.zeros_like(out)
    for k, w, canonical, mode, use_sk in [(21, 11, True, 0, False), (21, 11, False, 0, True), (15, 17, True, 1, False)]:
        b = sm.Builder(k, w, canonical, mode)
        results = []
        for force_generic in (False, True):
            gpu.force_generic(force_generic)
            offs = sm.run_batch_device(b, d, [int(x) for x in lens], out, sk if use_sk else None, base_offsets=offs_b)
            assert gpu.last_path() == (sm.PATH_GENERIC if force_generic else sm.PATH_FUSED)
            results.append((offs, out[: offs[-1]].cpu().numpy().view(np.uint32).copy(),
                            sk[: offs[-1]].cpu().numpy().view(np.uint32).copy()))
        gpu.force_generic(False)
        assert results[0][0] == results[1][0]
        assert np.array_equal(results[0][1], results[1][1])
        if use_sk:
            assert np.array_equal(results[0][2], results[1][2])
        offs, host, hsk = results[0]
        assert offs[0] == 0 and len(offs) == n_seq + 1
        for i in list(range(12)) + [int(x) for x in rng.integers(0, n_seq, size=150)]:
            n = int(lens[i])
            if use_sk:
                want, wsk = oracle.run(data, n, k, w, canonical=canonical, mode=mode, super_kmers=True,
                                       base_offset=int(starts[i]))
                assert np.array_equal(hsk[offs[i]:offs[i + 1]], wsk), (i, n)
            else:
                want = oracle.run(data, n, k, w, canonical=canonical, mode=mode, base_offset=int(starts[i]))
            assert np.array_equal(host[offs[i]:offs[i + 1]], want), (i, n, k, w)


def test_pack_ascii_device(sm, oracle, gpu):
    """PackedSeqVec::from_ascii on the device (aligned 16-base fast path, tails, odd alignments)."""
    import ctypes as C
    import torch
    rng = np.random.default_rng(2)
    letters = np.frombuffer(b"ACGTacgt", dtype=np.uint8)
    for n in (0, 1, 15, 16, 17, 63, 64, 1000, 100_003):
        for shift in (0, 1, 5):
            a = letters[rng.integers(0, 8, size=n + shift)]
            d_a = torch.from_numpy(a.copy()).cuda()
            d_p = torch.zeros((n + 3) // 4 + 8, dtype=torch.uint8, device="cuda")
            sm._check(sm.lib().mm_pack_ascii_device_async(gpu.h, C.c_void_p(d_a.data_ptr() + shift), n,
                                                          C.c_void_p(d_p.data_ptr())))
            gpu.sync()
            want = oracle.pack_ascii(a[shift:].tobytes())
            assert np.array_equal(d_p[: (n + 3) // 4].cpu().numpy(), want[: (n + 3) // 4]), (n, shift)


def test_maximum_length(sm, oracle, gpu):
    """Sequences up to 2^32 - 1 bases (src/sliding_min.rs:96-99): positions above 2^31 are emitted
    correctly; 2^32 bases is rejected like the reference's assert."""
    import ctypes as C
    import torch
    n, k, w = (1 << 32) - 1, 21, 11
    d = sm.generate_device(n, 9)
    cap = int(n * 2.3 / (w + 1))
    out = torch.empty(cap, dtype=torch.int32, device="cuda")
    b = sm.canonical_minimizers(k, w)
    c = b.run_device(d, n, out)
    assert abs(c / n - 2.0 / (w + 1)) < 1e-3
    whole = _device_checksum(out, c)
    # the last megabase against the oracle (positions > 2^32 - 2^20)
    m = 1_000_000
    tail_start = n - m
    tail = oracle.run(oracle.gen_packed(9, m, first_base=tail_start), m, k, w, canonical=True)
    got_tail = (out[c - len(tail) + 50: c].to(torch.int64) & 0xFFFFFFFF).cpu().numpy() - tail_start
    assert np.array_equal(got_tail, tail[50:].astype(np.int64))
    # three window-range shards reproduce it
    nw = n - (k + w - 1) + 1
    cuts = [0, nw // 2 - 7, nw - 3, nw]
    tc, tp, tw = 0, 0, 0
    for a, e in zip(cuts[:-1], cuts[1:]):
        cc = b.run_device(d, n, out, win_begin=a, win_end=e)
        _, plain, weighted = _device_checksum(out, cc)
        tw = (tw + weighted + tc * plain) & ((1 << 64) - 1)
        tp += plain
        tc += cc
    assert (tc, tp, tw) == whole
    plan = b.plan()
    cnt = C.c_uint64()
    code = sm.lib().mm_run_device(plan.h, gpu.h, C.c_void_p(d.data_ptr()), d.numel(), 0, 1 << 32, 0, sm.U64_MAX,
                                  C.c_void_p(out.data_ptr()), None, cap, C.byref(cnt))
    assert code == sm.ERR["LEN_TOO_LARGE"]




def _check_reads(sm, oracle, k, w, canonical, mode, n_reads, stride, read_len, lens, base_offset, seed):
    import torch
    span = (n_reads - 1) * stride + read_len if n_reads else 0
    data = oracle.gen_packed(seed, base_offset + span + 64)
    d = torch.from_numpy(data).cuda()
    d_lens = torch.from_numpy(lens.astype(np.int32)).cuda() if lens is not None else None
    cap = max(1, n_reads * max(1, read_len))
    out = torch.zeros(cap, dtype=torch.int32, device="cuda")
    offs = torch.full((n_reads + 1,), -1, dtype=torch.int64, device="cuda")
    b = sm.Builder(k, w, canonical, mode)
    total = sm.run_reads_device(b, d, n_reads, stride, read_len, out, offs, read_lens=d_lens,
                                base_offset=base_offset)
    h_offs = offs.cpu().numpy()
    host = out[:total].cpu().numpy().view(np.uint32)
    assert h_offs[0] == 0 and h_offs[-1] == total
    for r in range(n_reads):
        n = int(lens[r]) if lens is not None else read_len
        want = oracle.run(data, n, k, w, canonical=canonical, mode=mode, base_offset=base_offset + r * stride)
        got = host[h_offs[r]:h_offs[r + 1]]
        assert np.array_equal(got, want), (k, w, canonical, r, n, got[:8], want[:8])
    return total


def test_reads_mode(sm, oracle, gpu):
    """Batched short reads (one lane per read, one launch): every read equals an independent run
    of the oracle on that read (src/lib.rs:378 called once per read)."""
    rng = np.random.default_rng(11)
    # fixed-length reads, strides that are not multiples of 4 / 16, non-zero buffer offsets
    for k, w, canonical in [(21, 11, True), (15, 5, True), (31, 19, True), (21, 11, False), (9, 10, False),
                            (5, 7, False), (13, 15, True)]:
        for n_reads, stride, read_len, off in [(1, 150, 150, 0), (700, 151, 150, 3), (257, 160, 101, 17),
                                               (300, 250, 250, 0)]:
            _check_reads(sm, oracle, k, w, canonical, 0, n_reads, stride, read_len, None, off, 5)
            assert gpu.last_path() == 1
    # variable lengths, including reads shorter than l (no window) and of length exactly l
    for k, w, canonical in [(21, 11, True), (7, 5, False)]:
        l = k + w - 1
        n_reads = 1000
        lens = rng.integers(0, 301, size=n_reads)
        lens[:6] = [0, l - 1, l, l + 1, 300, 1]
        _check_reads(sm, oracle, k, w, canonical, 0, n_reads, 304, 300, lens, 5, 6)
    # syncmer plans have no prebuilt reads-mode kernel: specialised at first use, still ONE launch;
    # with MM_JIT=0 one launch per read - same results
    for (k, w, canonical, mode) in [(15, 17, True, 1), (15, 17, True, 2), (9, 11, False, 1), (8, 5, False, 2),
                                    (10, 12, True, 1)]:
        lens = rng.integers(0, 301, size=700)
        _check_reads(sm, oracle, k, w, canonical, mode, 700, 303, 300, lens, 3, 70 + w + mode)
        assert gpu.last_path() == 1, sm.lib().mm_last_error()
    _check_reads(sm, oracle, 15, 17, True, 1, 600, 151, 150, None, 2, 7)
    assert gpu.last_path() == 1
    os.environ["MM_JIT"] = "0"
    try:
        _check_reads(sm, oracle, 15, 17, True, 1, 40, 150, 150, None, 2, 7)
    finally:
        del os.environ["MM_JIT"]
    # a w without a prebuilt instance: specialised at run time (one launch) or, with MM_JIT=0, one
    # launch per read through the generic family
    _check_reads(sm, oracle, 12, 34, True, 0, 40, 150, 149, None, 0, 8)
    assert gpu.last_path() == 1
    os.environ["MM_JIT"] = "0"
    try:
        _check_reads(sm, oracle, 12, 34, True, 0, 40, 150, 149, None, 0, 8)
        assert gpu.last_path() == 2
        lens = rng.integers(0, 200, size=30)
        _check_reads(sm, oracle, 12, 34, True, 0, 30, 200, 199, lens, 1, 9)
    finally:
        del os.environ["MM_JIT"]
    # every window size with an instance
    for canonical in (True, False):
        for w in sm.prebuilt_window_sizes(canonical, reads=True):
            k = 20 if (canonical and w % 2 == 0) else 21
            _check_reads(sm, oracle, k, w, canonical, 0, 300, 153, 151, None, 1, 10 + w)
            assert gpu.last_path() == 1
    # zero reads
    _check_reads(sm, oracle, 21, 11, True, 0, 0, 150, 150, None, 0, 1)


def test_reads_mode_superkmers(sm, oracle, gpu):
    """Super-k-mer indices per read in one launch (Builder::super_kmers + run per read,
    src/lib.rs:341,545-576): positions and read-local window indices equal the oracle's per read;
    reads too long for the packed list entry fall back to one launch per read."""
    import torch
    rng = np.random.default_rng(23)
    for k, w, canonical, n_reads, stride, read_len, off in [(21, 11, True, 900, 151, 150, 0), (21, 11, False, 300, 303, 300, 3),
                                                            (15, 5, True, 300, 160, 101, 1), (31, 19, True, 300, 250, 250, 2),
                                                            (12, 18, True, 40, 400, 397, 0), (21, 11, True, 3, 9000, 8999, 1)]:
        span = (n_reads - 1) * stride + read_len
        data = oracle.gen_packed(90 + w, off + span + 64)
        d = torch.from_numpy(data).cuda()
        lens = rng.integers(0, read_len + 1, size=n_reads)
        lens[:2] = [read_len, k + w - 1]
        d_lens = torch.from_numpy(lens.astype(np.int32)).cuda()
        out = torch.zeros(n_reads * read_len, dtype=torch.int32, device="cuda")
        sk = torch.zeros_like(out)
        offs = torch.zeros(n_reads + 1, dtype=torch.int64, device="cuda")
        total = sm.run_reads_device(sm.Builder(k, w, canonical, 0), d, n_reads, stride, read_len, out, offs,
                                    read_lens=d_lens, base_offset=off, out_sk=sk)
        if read_len < 4000:
            assert gpu.last_path() == sm.PATH_FUSED, sm.lib().mm_last_error()
        ho = offs.cpu().numpy()
        hp, hs = out[:total].cpu().numpy().view(np.uint32), sk[:total].cpu().numpy().view(np.uint32)
        assert ho[0] == 0 and ho[-1] == total
        for r in range(n_reads):
            want, wsk = oracle.run(data, int(lens[r]), k, w, canonical=canonical, base_offset=off + r * stride,
                                   super_kmers=True)
            assert np.array_equal(hp[ho[r]:ho[r + 1]], want), (k, w, r)
            assert np.array_equal(hs[ho[r]:ho[r + 1]], wsk), (k, w, r)
    # syncmer plans have no super-k-mers (src/lib.rs:339)
    with pytest.raises(sm.MinimizerError):
        sm.run_reads_device(sm.Builder(15, 17, True, 1), d, 2, 100, 100, out, offs, out_sk=sk)


def test_reads_mode_dense_and_long(sm, oracle, gpu):
    """Low-complexity reads (every window emits -> list overflow -> direct redo) and reads too
    long for the LDS lists (fallback to one launch per read)."""
    import torch
    n_reads, stride, read_len = 520, 152, 150
    nbytes = (n_reads * stride + 64) // 4
    for fill in (0x00, 0xFF, 0x1B):
        data = np.full(nbytes, fill, dtype=np.uint8)
        d = torch.from_numpy(data).cuda()
        for k, w, canonical in [(21, 11, True), (21, 11, False)]:
            out = torch.zeros(n_reads * read_len, dtype=torch.int32, device="cuda")
            offs = torch.zeros(n_reads + 1, dtype=torch.int64, device="cuda")
            total = sm.run_reads_device(sm.Builder(k, w, canonical, 0), d, n_reads, stride, read_len, out, offs)
            h_offs = offs.cpu().numpy()
            host = out[:total].cpu().numpy().view(np.uint32)
            for r in (0, 1, 255, 256, 519):
                want = oracle.run(data, read_len, k, w, canonical=canonical, base_offset=r * stride)
                assert np.array_equal(host[h_offs[r]:h_offs[r + 1]], want), (fill, k, w, r)
    _check_reads(sm, oracle, 21, 11, True, 0, 6, 70_000, 69_999, None, 0, 3)


# ------------------------------------------------- skip-ambiguous windows (PackedNSeq)
def _ascii_with_n(rng, n, frac, runs):
    a = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].copy()
    if n and frac > 0:
        if runs:
            for _ in range(max(1, int(n * frac / 30))):
                s = int(rng.integers(0, n))
                a[s:s + int(rng.integers(1, 60))] = ord("N")
        else:
            a[rng.integers(0, n, size=max(1, int(n * frac)))] = ord("N")
    return a


def test_skip_ambiguous_small_sweep(sm, oracle, gpu):
    """The reference's skip_ambiguous test shape (src/test.rs:428-482: 100 bases, 1% N, all odd
    l <= 64) plus lengths around l, against the oracle; fused and generic families; PackedNSeq
    and ASCII entry points; slices with non-zero base / ambiguity offsets."""
    rng = np.random.default_rng(21)
    for n, frac, runs in [(100, 0.01, False), (100, 0.06, False), (333, 0.05, True)]:
        a = _ascii_with_n(rng, n, frac, runs)
        nseq = sm.PackedNSeqVec.from_ascii(a.tobytes())
        packed, amb = oracle.pack_ascii_n(a.tobytes())
        assert np.array_equal(nseq.amb[: (n + 7) // 8], amb[: (n + 7) // 8])
        for k in range(1, 65, 2):
            for w in range(1, 64, 3):
                l = k + w - 1
                if l % 2 == 0 or l > 64:
                    continue
                for mode in (0, 1, 2):
                    if mode == 2 and w % 2 == 0:
                        continue
                    b = sm.Builder(k, w, True, mode)
                    want = list(map(int, oracle.run_skip_ambiguous(packed, amb, n, k, w, mode=mode)))
                    for force_generic in (False, True):
                        gpu.force_generic(force_generic)
                        assert b.run_skip_ambiguous_windows_once(nseq) == want, (n, k, w, mode, force_generic)
                    gpu.force_generic(False)
                    if mode == 0:
                        assert b.run_skip_ambiguous_windows_once(sm.AsciiSeq(a.tobytes())) == want
                        s0, s1 = 3, n - 2
                        sl = nseq.slice(s0, s1)
                        want_sl = list(map(int, oracle.run_skip_ambiguous(packed, amb, s1 - s0, k, w,
                                                                          base_offset=s0, amb_offset=s0)))
                        assert b.run_skip_ambiguous_windows_once(sl) == want_sl, (n, k, w, "slice")
    # forward plans are rejected like the reference's assert (src/minimizers.rs:176)
    with pytest.raises(sm.MinimizerError) as e:
        sm.minimizers(5, 7).run_skip_ambiguous_windows_once(nseq)
    assert e.value.code == sm.ERR["HASHER_NOT_CANONICAL"]


@pytest.mark.parametrize("k,w,mode", [(21, 11, 0), (31, 51, 0), (15, 17, 1), (15, 17, 2), (12, 18, 0), (7, 5, 0)])
def test_skip_ambiguous_large_device(sm, oracle, gpu, k, w, mode):
    """4 Mbp with isolated Ns, N runs (assembly gaps), Ns at both ends; device-resident input packed
    by mm_pack_ascii_n_device_async; whole run, window-range shards, fused vs generic."""
    import ctypes as C
    import torch
    rng = np.random.default_rng(k * 100 + w)
    n = 4_000_003
    a = _ascii_with_n(rng, n, 0.002, False)
    for s in rng.integers(0, n - 70_000, size=12):
        a[s:s + int(rng.integers(1, 70_000))] = ord("n")
    a[:3] = ord("N")
    a[-2:] = ord("N")
    d_a = torch.from_numpy(a).cuda()
    d_p = torch.zeros((n + 3) // 4 + 64, dtype=torch.uint8, device="cuda")
    d_m = torch.zeros((n + 7) // 8 + 64, dtype=torch.uint8, device="cuda")
    sm._check(sm.lib().mm_pack_ascii_n_device_async(gpu.h, C.c_void_p(d_a.data_ptr()), n,
                                                    C.c_void_p(d_p.data_ptr()), C.c_void_p(d_m.data_ptr())))
    gpu.sync()
    packed, amb = oracle.pack_ascii_n(a.tobytes())
    assert np.array_equal(d_p[: (n + 3) // 4].cpu().numpy(), packed[: (n + 3) // 4])
    assert np.array_equal(d_m[: (n + 7) // 8].cpu().numpy(), amb[: (n + 7) // 8])
    want = oracle.run_skip_ambiguous(packed, amb, n, k, w, mode=mode)
    b = sm.Builder(k, w, True, mode)
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    c = b.run_skip_ambiguous_device(d_p, d_m, n, out)
    assert gpu.last_path() == sm.PATH_FUSED  # (also for the w = 18 case: prebuilt since round 3)
    assert np.array_equal(out[:c].cpu().numpy().view(np.uint32), want)
    # window-range shards concatenate to the whole (a shard dedups against the window before it)
    nw = n - (k + w - 1) + 1
    cuts = [0, nw // 3 + 5, 2 * nw // 3 - 7, nw]
    parts = []
    for s, e in zip(cuts[:-1], cuts[1:]):
        cc = b.run_skip_ambiguous_device(d_p, d_m, n, out, win_begin=s, win_end=e)
        parts.append(out[:cc].cpu().numpy().view(np.uint32).copy())
    assert np.array_equal(np.concatenate(parts), want)
    gpu.force_generic(True)
    try:
        c = b.run_skip_ambiguous_device(d_p, d_m, n, out)
        assert gpu.last_path() == sm.PATH_GENERIC
        assert np.array_equal(out[:c].cpu().numpy().view(np.uint32), want)
    finally:
        gpu.force_generic(False)


def test_skip_ambiguous_reads(sm, oracle, gpu):
    """Reads with Ns in one launch: each read equals run_skip_ambiguous_windows on that read."""
    import torch
    rng = np.random.default_rng(31)
    for k, w, n_reads, stride, read_len, off, mode in [(21, 11, 900, 151, 150, 0, 0), (15, 5, 300, 160, 101, 3, 0),
                                                       (31, 19, 300, 250, 250, 1, 0), (12, 18, 40, 150, 150, 2, 0),
                                                       (15, 17, 500, 151, 150, 1, 1), (15, 9, 300, 151, 150, 0, 2)]:
        span = n_reads * stride + 64 + off
        a = _ascii_with_n(rng, span, 0.004, False)
        for s in rng.integers(0, span - 100, size=20):
            a[s:s + int(rng.integers(1, 90))] = ord("N")
        packed, amb = oracle.pack_ascii_n(a.tobytes())
        d_p, d_m = torch.from_numpy(packed).cuda(), torch.from_numpy(amb).cuda()
        lens = rng.integers(0, read_len + 1, size=n_reads)
        for use_lens in (False, True):
            d_lens = torch.from_numpy(lens.astype(np.int32)).cuda() if use_lens else None
            out = torch.zeros(n_reads * read_len, dtype=torch.int32, device="cuda")
            offs = torch.zeros(n_reads + 1, dtype=torch.int64, device="cuda")
            total = sm.run_reads_device(sm.Builder(k, w, True, mode), d_p, n_reads, stride, read_len, out, offs,
                                        read_lens=d_lens, base_offset=off, d_amb=d_m, amb_offset=off)
            assert gpu.last_path() == sm.PATH_FUSED
            ho = offs.cpu().numpy()
            hp = out[:total].cpu().numpy().view(np.uint32)
            assert ho[-1] == total
            for r in range(n_reads):
                m = int(lens[r]) if use_lens else read_len
                want = oracle.run_skip_ambiguous(packed, amb, m, k, w, mode=mode, base_offset=off + r * stride,
                                                 amb_offset=off + r * stride)
                assert np.array_equal(hp[ho[r]:ho[r + 1]], want), (k, w, r, m)


# ------------------------------------------------- run-time specialisation (any w <= 128)
@pytest.mark.parametrize("k,w,canonical", [(12, 34, True), (21, 36, False), (31, 35, True), (14, 64, True),
                                            (16, 100, True), (9, 128, False)])
def test_runtime_specialised_window_sizes(sm, oracle, gpu, k, w, canonical):
    """Window sizes without a prebuilt instance run the same fused kernel, compiled with hiprtc at
    first use (mm_jit.hip): minimizers, super-k-mers, syncmers, window ranges, reads mode — all
    against the oracle; with MM_JIT=0 the generic family gives the same answers."""
    import torch
    n = 300_011
    data = oracle.gen_packed(40 + w, n)
    ps = sm.PackedSeq(data, 0, n)
    d = torch.from_numpy(data).cuda()
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    sk = torch.zeros(n, dtype=torch.int32, device="cuda")
    for mode in (0, 1, 2):
        if mode == 2 and w % 2 == 0:
            continue
        want = oracle.run(data, n, k, w, canonical=canonical, mode=mode)
        b = sm.Builder(k, w, canonical, mode)
        c = b.run_device(d, n, out)
        assert gpu.last_path() == sm.PATH_FUSED, sm.lib().mm_last_error()
        assert np.array_equal(out[:c].cpu().numpy().view(np.uint32), want), (k, w, mode)
        os.environ["MM_JIT"] = "0"
        try:
            c = b.run_device(d, n, out)
            assert gpu.last_path() == sm.PATH_GENERIC
            assert np.array_equal(out[:c].cpu().numpy().view(np.uint32), want), (k, w, mode, "generic")
        finally:
            del os.environ["MM_JIT"]
    # super-k-mers and a window range
    want, wsk = oracle.run(data, n, k, w, canonical=canonical, super_kmers=True)
    b = sm.Builder(k, w, canonical, 0)
    c = b.run_device(d, n, out, out_sk=sk)
    assert gpu.last_path() == sm.PATH_FUSED
    assert np.array_equal(out[:c].cpu().numpy().view(np.uint32), want)
    assert np.array_equal(sk[:c].cpu().numpy().view(np.uint32), wsk)
    nw = n - (k + w - 1) + 1
    parts = []
    for s0, s1 in [(0, nw // 2 + 3), (nw // 2 + 3, nw)]:
        cc = b.run_device(d, n, out, win_begin=s0, win_end=s1)
        parts.append(out[:cc].cpu().numpy().view(np.uint32).copy())
    assert np.array_equal(np.concatenate(parts), want)
    # reads mode
    _check_reads(sm, oracle, k, w, canonical, 0, 300, 400, 397, None, 1, 60 + w)
    assert gpu.last_path() == sm.PATH_FUSED


def test_host_entry_point_pipelined(sm, oracle, gpu):
    """Long sequences take the pipelined host path (chunks: H2D, kernel and D2H overlapped on
    three streams): identical to the device-resident run, with super-k-mer indices, with a non-zero
    base offset, and with a capacity that is too small."""
    import ctypes as C
    import torch
    n = 70_000_000
    data = oracle.gen_packed(21, n + 8)
    d = torch.from_numpy(data).cuda()
    L = sm.lib()
    u8p, u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)
    for k, w, canonical, mode, use_sk, off in [(21, 11, True, 0, False, 0), (21, 11, False, 0, True, 3),
                                                (15, 17, True, 1, False, 1)]:
        b = sm.Builder(k, w, canonical, mode)
        cap = int(n * 0.2)
        dev_out = torch.zeros(cap, dtype=torch.int32, device="cuda")
        dev_sk = torch.zeros(cap, dtype=torch.int32, device="cuda") if use_sk else None
        c_dev = b.run_device(d, n, dev_out, out_sk=dev_sk, base_offset=off)
        want = dev_out[:c_dev].cpu().numpy().view(np.uint32)
        pos = np.zeros(cap, dtype=np.uint32)
        sk = np.zeros(cap, dtype=np.uint32) if use_sk else None
        cnt = C.c_uint64()
        sm._check(L.mm_run_host(b.plan().h, gpu.h, data.ctypes.data_as(u8p), off, n, pos.ctypes.data_as(u32p),
                                sk.ctypes.data_as(u32p) if use_sk else None, cap, C.byref(cnt)))
        assert cnt.value == c_dev
        assert np.array_equal(pos[:c_dev], want), (k, w, mode)
        if use_sk:
            assert np.array_equal(sk[:c_dev], dev_sk[:c_dev].cpu().numpy().view(np.uint32))
        # the one-shot path gives the same
        os.environ["MM_NO_PIPELINE"] = "1"
        try:
            pos2 = np.zeros(cap, dtype=np.uint32)
            sm._check(L.mm_run_host(b.plan().h, gpu.h, data.ctypes.data_as(u8p), off, n, pos2.ctypes.data_as(u32p),
                                    None, cap, C.byref(cnt)))
            assert cnt.value == c_dev and np.array_equal(pos2[:c_dev], want)
        finally:
            del os.environ["MM_NO_PIPELINE"]
    # capacity too small: the needed count comes back with the error
    small = np.zeros(1000, dtype=np.uint32)
    code = L.mm_run_host(b.plan().h, gpu.h, data.ctypes.data_as(u8p), 0, n, small.ctypes.data_as(u32p), None, 1000,
                         C.byref(cnt))
    assert code == sm.ERR["CAPACITY"] and cnt.value > 1000
