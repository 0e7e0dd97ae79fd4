"""Round-5 GPU tests (all through the C ABI, bit-exact against the oracle):

* the one-round launch rule with lanes that geometry() bounds below 6 blocks - super-k-mer runs
  with w >= 86, whose 16-bit list entry (window << shift) + offset caps the lane length;
* FASTQ text with blank bytes in front of the first '@';
* the host entry point's mechanisms: copy engines / copy kernels / the fused kernel's own stores
  into the caller's page-locked buffer, page-locked and pageable caller buffers, super-k-mer indices, base offsets,
  a capacity that is too small - every combination == the device-resident run;
* BASELINE config 5's "super-k-mer boundary emission" at full size: canonical minimizers
  k=21 w=11 with super-k-mer indices on 3.1 Gbp (src/lib.rs:341-351,545-576, src/collect.rs:39-76) through
  size-independent properties.
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(out, c):
    return out[:c].cpu().numpy().view(np.uint32)


def test_superkmers_large_w_one_round(sm, oracle, gpu):
    """super-k-mer indices with w = 100 / 128: the list entry (window << 7 or 8) + offset bounds a lane at 5 / 2 W-blocks;
    runs of 0.6 .. 1 round of the chip's workgroup slots used to be forced to 6 blocks (entries overflowed 16 bits,
    positions and indices silently wrong).  Lengths on both sides of one round for 1 and 2 resident workgroups per CU."""
    import torch
    sizes = [12_000_013, 22_000_013, 30_000_013, 45_000_013, 60_000_013]
    n_max = max(sizes)
    data = oracle.gen_packed(91, n_max + 64)
    d = torch.from_numpy(data).cuda()
    out = torch.zeros(n_max // 20, dtype=torch.int32, device="cuda")
    sk = torch.zeros_like(out)
    for k, w in ((21, 100), (20, 128)):
        want_all, wsk_all = oracle.run(data, n_max, k, w, canonical=False, super_kmers=True)
        for n in sizes:
            c = sm.Builder(k, w, False, 0).run_device(d, n, out, out_sk=sk)
            assert gpu.last_path() == sm.PATH_FUSED
            # the run on the first n bases = the prefix of the run on n_max bases whose window index is < n_w
            n_w = n - (k + w - 1) + 1
            keep = int(np.searchsorted(wsk_all, n_w, side="left"))
            assert c == keep, (w, n, c, keep)
            assert np.array_equal(_dev(out, c), want_all[:keep]), (w, n)
            assert np.array_equal(_dev(sk, c), wsk_all[:keep]), (w, n)


def test_fastq_leading_blank_bytes(sm, oracle, gpu):
    """mm_fasta_pack_device finds FASTQ by the first non-blank byte; the four-line packer counts lines from the start of
    ITS text, so the call hands it the text from the '@' on and keeps the records' text positions absolute."""
    body = b"@r1 x\nACGTTGCA\n+\nIIIIIIII\n@r2\nTTGACC\n+r2\nIIIIII\n@r3\nGATTACA\n+\n!!!!!!!\n"
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    many = b"".join(b"@q%d\n" % i + acgt[rng.integers(0, 4, 150)].tobytes() + b"\n+\n" + b"I" * 150 + b"\n" for i in range(600))
    for lead in (b"\n", b"\n\n\n", b"  ", b"\r\n\t \n", b"\n" * 7 + b" "):
        for text in (body, many):
            full = lead + text
            rec = sm.fasta_pack_device(full, max_records=1 << 12)
            want = oracle.fastq_records(text)
            assert len(rec) == len(want), (lead, len(rec), len(want))
            packed = rec.packed.cpu().numpy()
            for i, (pos, _name, seq) in enumerate(want):
                b, e = int(rec.base[i]), int(rec.base[i + 1])
                assert e - b == len(seq), (lead, i)
                assert int(rec.text_pos[i]) == pos + len(lead), (lead, i)
                if i < 8 or i == len(want) - 1:
                    codes = [(packed[(b + j) // 4] >> (2 * ((b + j) % 4))) & 3 for j in range(len(seq))]
                    assert codes == [(c >> 1) & 3 for c in seq], (lead, i)


def test_host_entry_point_mechanisms(sm, oracle, gpu):
    """mm_run_host (src/lib.rs:378: host memory in, Vec<u32> out) on a sequence long enough for the pipelined path: every
    mechanism of its two legs gives the device-resident run's output - with page-locked caller buffers (where the copy
    kernels and the direct stores apply) and with pageable ones (where they fall back to the engines)."""
    import torch
    n = 110_000_000
    data = oracle.gen_packed(23, n + 64)
    d = torch.from_numpy(data).cuda()
    L = sm.lib()
    u8p, u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)
    cap = int(n * 0.19)
    hp, _o1 = sm.pinned_array((len(data),), np.uint8)
    hp[:] = data
    ppos, _o2 = sm.pinned_array((cap + 3,), np.uint32)
    psk, _o3 = sm.pinned_array((cap + 3,), np.uint32)
    saved = {k: os.environ.get(k) for k in ("MM_HOST_OUT", "MM_HOST_IN", "MM_PIPE_CHUNKS")}
    ran = 0
    try:
        for k, w, canonical, mode, use_sk, off in [(21, 11, True, 0, False, 0), (21, 11, False, 0, True, 3), (15, 17, True, 1, False, 1)]:
            b = sm.Builder(k, w, canonical, mode)
            dev_out = torch.zeros(cap, dtype=torch.int32, device="cuda")
            dev_sk = torch.zeros(cap, dtype=torch.int32, device="cuda") if use_sk else None
            c_dev = b.run_device(d, n, dev_out, out_sk=dev_sk, base_offset=off)
            want = _dev(dev_out, c_dev)
            want_sk = _dev(dev_sk, c_dev) if use_sk else None
            for out_m, in_m, chunks, pinned in [("engine", "engine", None, True), ("blit", "engine", None, True),
                                                ("direct", "engine", None, True), ("engine", "blit", None, True),
                                                ("blit", "blit", "3", True), ("direct", "blit", "7", True),
                                                ("engine", "engine", "64", True), ("blit", "blit", None, False),
                                                ("direct", "engine", "5", False)]:
                os.environ["MM_HOST_OUT"], os.environ["MM_HOST_IN"] = out_m, in_m
                if chunks:
                    os.environ["MM_PIPE_CHUNKS"] = chunks
                else:
                    os.environ.pop("MM_PIPE_CHUNKS", None)
                # (page-locked outputs at an odd dword offset: the copy kernel's 16-byte stores meet an unaligned start)
                src = hp if pinned else data
                pos = ppos[1: cap + 1] if pinned else np.zeros(cap, dtype=np.uint32)
                sk = (psk[3: cap + 3] if pinned else np.zeros(cap, dtype=np.uint32)) if use_sk else None
                pos[:] = 0xDEADBEEF
                cnt = C.c_uint64()
                sm._check(L.mm_run_host(b.plan().h, gpu.h, src.ctypes.data_as(u8p), off, n, pos.ctypes.data_as(u32p),
                                        sk.ctypes.data_as(u32p) if use_sk else None, cap, C.byref(cnt)))
                assert cnt.value == c_dev, (k, w, out_m, in_m, chunks, pinned)
                assert np.array_equal(pos[:c_dev], want), (k, w, out_m, in_m, chunks, pinned)
                assert pos[c_dev] == 0xDEADBEEF  # nothing past the count
                if use_sk:
                    assert np.array_equal(sk[:c_dev], want_sk), (k, w, out_m, in_m, chunks, pinned)
                ran += 1
            # a capacity that is too small: the needed count comes back with the error, nothing is written past it
            for out_m in ("engine", "blit", "direct"):
                os.environ["MM_HOST_OUT"], os.environ["MM_HOST_IN"] = out_m, "engine"
                os.environ.pop("MM_PIPE_CHUNKS", None)
                small = ppos[: 1000 + 1]
                small[:] = 0xDEADBEEF
                cnt = C.c_uint64()
                code = L.mm_run_host(b.plan().h, gpu.h, hp.ctypes.data_as(u8p), off, n, small.ctypes.data_as(u32p), None, 1000,
                                     C.byref(cnt))
                assert code == sm.ERR["CAPACITY"] and cnt.value == c_dev, (out_m, code, cnt.value)
                assert small[1000] == 0xDEADBEEF and np.array_equal(small[:1000], want[:1000]), out_m
    finally:
        for k_, v in saved.items():
            if v is None:
                os.environ.pop(k_, None)
            else:
                os.environ[k_] = v
    assert ran == 27


def _checksum(t, c):
    """(count, sum, order-sensitive weighted sum mod 2^64) of the first c u32 elements of a device tensor"""
    import torch
    plain, weighted = 0, 0
    step = 1 << 26
    for a in range(0, c, step):
        e = min(c, a + step)
        v = t[a:e].to(torch.int64) & 0xFFFFFFFF
        plain += int(v.sum().item())
        idx = torch.arange(a + 1, e + 1, dtype=torch.int64, device=t.device)
        weighted = (weighted + int((v * idx).sum().item())) & ((1 << 64) - 1)
    return c, plain, weighted


def test_full_size_superkmers(sm, oracle, gpu):
    """BASELINE config 5 says "with super-k-mer boundary emission"; the reference refuses `.super_kmers()` on syncmer
    builders (src/lib.rs:339,496-500), so the conformant operation at that size is canonical_minimizers(21, 11)
    .super_kmers(&mut sk) on 3.1 Gbp (src/lib.rs:341-351,545-576; collector src/collect.rs:39-76, known answer
    src/test.rs:344-356).  Positions and indices by order-sensitive checksum: fused == concatenated window-range shards
    == generic family; positions == the run without indices; indices strictly increasing, each the first window whose
    minimizer is its position (pos - w < sk <= pos); head and tail against the oracle."""
    import torch
    n, k, w = 3_100_000_000, 21, 11
    d = sm.generate_device(n, 3)
    cap = int(n * 2.0 / (w + 1) * 1.1)
    out = torch.empty(cap, dtype=torch.int32, device="cuda")
    sk = torch.empty(cap, dtype=torch.int32, device="cuda")
    b = sm.canonical_minimizers(k, w)
    c = b.run_device(d, n, out, out_sk=sk)
    assert gpu.last_path() == sm.PATH_FUSED
    whole_pos, whole_sk = _checksum(out, c), _checksum(sk, c)
    # the same positions as the run without indices
    out2 = torch.empty(cap, dtype=torch.int32, device="cuda")
    c2 = b.run_device(d, n, out2)
    assert c2 == c and _checksum(out2, c2) == whole_pos
    del out2
    # the index is the first window that selects the position: strictly increasing, and the position lies inside it
    s64 = sk[:c].to(torch.int64) & 0xFFFFFFFF
    p64 = out[:c].to(torch.int64) & 0xFFFFFFFF
    assert int(s64[0].item()) == 0 and bool((s64[1:] > s64[:-1]).all())
    rel = p64 - s64
    assert int(rel.min().item()) >= 0 and int(rel.max().item()) <= w - 1
    del s64, p64, rel
    # head and tail against the oracle
    m = 1_000_000
    hp, hs = oracle.run(oracle.gen_packed(3, m + 256), m + 256, k, w, canonical=True, super_kmers=True)
    keep = hp < m - 256
    assert np.array_equal(_dev(out, int(keep.sum())), hp[keep]) and np.array_equal(_dev(sk, int(keep.sum())), hs[keep])
    tail_start = n - m
    tp, ts = oracle.run(oracle.gen_packed(3, m, first_base=tail_start), m, k, w, canonical=True, super_kmers=True)
    got_p = out[c - len(tp) + 50: c].cpu().numpy().view(np.uint32).astype(np.int64) - tail_start
    got_s = sk[c - len(tp) + 50: c].cpu().numpy().view(np.uint32).astype(np.int64) - tail_start
    assert np.array_equal(got_p, tp[50:].astype(np.int64)) and np.array_equal(got_s, ts[50:].astype(np.int64))
    # window-range shards laid end to end (a shard's first entry is compared with the window before its range)
    nw = n - (k + w - 1) + 1
    cuts = [0, nw // 3 + 11, 2 * nw // 3 - 5, nw]
    tot = [0, 0, 0]
    tot_s = [0, 0, 0]
    for a, e in zip(cuts[:-1], cuts[1:]):
        cc = b.run_device(d, n, out, out_sk=sk, win_begin=a, win_end=e)
        for acc, t in ((tot, out), (tot_s, sk)):
            _, plain, weighted = _checksum(t, cc)
            acc[2] = (acc[2] + weighted + acc[0] * plain) & ((1 << 64) - 1)
            acc[1] += plain
        tot[0] += cc
        tot_s[0] += cc
    assert tuple(tot) == whole_pos and tuple(tot_s) == whole_sk
    # generic family
    gpu.force_generic(True)
    try:
        cg = b.run_device(d, n, out, out_sk=sk)
        assert gpu.last_path() == sm.PATH_GENERIC
        assert _checksum(out, cg) == whole_pos and _checksum(sk, cg) == whole_sk
    finally:
        gpu.force_generic(False)


def test_cross_checks_in_the_experiments_build(sm):
    """the FASTA packers of rounds 2-4 (mm_fasta.hip: one pass over lines, three passes) and the split
    path (mm_split.hip, walk kernels) left the shipped library; they are cross-checks in libsimd_minimizers_amd_exp.so.
    Their tests - every FASTA test under all three packers, the packers against one another byte by byte, the split path
    against the oracle - run here in a child pytest that loads that build through MM_LIB_PATH."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if os.path.basename(sm.LIB_PATH).endswith("_exp.so"):
        pytest.skip("already running under the experiments build")
    lib = os.path.join(root, "simd-minimizers_amd", "libsimd_minimizers_amd_exp.so")
    assert os.path.exists(lib), "experiments library not built (make -C simd-minimizers_amd/csrc exp)"
    env = dict(os.environ, MM_LIB_PATH=lib, MM_ENV_DYNAMIC="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(root, "tests", "test_gpu_fasta.py"),
                        os.path.join(root, "tests", "test_gpu_round3.py") + "::test_split_path_matches_oracle",
                        os.path.join(root, "tests", "test_gpu_round3.py") + "::test_split_path_redo_and_flavours",
                        os.path.join(root, "tests", "test_gpu_round3.py") + "::test_split_path_host_pipeline",
                        os.path.join(root, "tests", "test_gpu_round4.py") + "::test_fasta_packers_agree_on_random_texts"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=3000)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], tail


def test_skip_ambiguous_large_windows_landing(sm, oracle, gpu):
    """Round 5: the skip-ambiguous walk over windows of 32 .. 96 keeps its look-ahead loads' landing data in LDS (in front
    of the lane lists).  Random sequences with isolated Ns and N runs over several tiles, windows with and without a fifth
    load dword, and DENSE tiles whose lists overflow (k = 1 .. 3: every base nearly its own k-mer - the redo pass walks
    with the same landing area, and list entries past a list's capacity must not reach it) against the oracle."""
    import torch
    rng = np.random.default_rng(55)
    checked = 0
    for (k, w) in ((1, 55), (1, 33), (3, 51), (3, 41), (19, 33), (21, 35), (31, 51), (21, 63), (19, 65), (21, 81), (31, 33)):
        assert (k + w - 1) % 2 == 1
        for n in (333, 5_003, 120_007, 1_500_013):
            a = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n + 8)].copy()
            if k <= 3 and n > 1000:
                a[n // 3: n // 3 + 900] = ord("A")          # a homopolymer stretch: every window ties
            a[rng.integers(0, n, size=max(1, n // 250))] = ord("N")
            s0 = int(rng.integers(0, n))
            a[s0:s0 + int(rng.integers(1, 200))] = ord("N")
            packed, amb = oracle.pack_ascii_n(a.tobytes())
            d_p, d_m = torch.from_numpy(packed).cuda(), torch.from_numpy(amb).cuda()
            out = torch.zeros(n + 8, dtype=torch.int32, device="cuda")
            c = sm.canonical_minimizers(k, w).run_skip_ambiguous_device(d_p, d_m, n, out)
            want = oracle.run_skip_ambiguous(packed, amb, n, k, w)
            assert c == len(want) and np.array_equal(_dev(out, c), want), (k, w, n, c, len(want))
            checked += 1
    assert checked == 44


def test_strand_vote_at_every_lane_alignment(sm, oracle, gpu):
    """The strand window's leaving base of a block's last step is base W of the block's hash-out view (one base beyond the
    block).  For w = 49 and w = 65 that base sits right at the edge of the bits a wide load keeps valid for every lane
    alignment; round 5 found lanes at the worst alignment reading a zero there, a strand count off by one, and a tie
    resolved to the wrong side (two emits missing in 1.5 Mbp).  Tie-heavy sequences (long two-letter stretches: most
    windows tie at their minimum and the vote decides), every base offset, several lane lengths so that lane starts take
    every residue mod 16, window sizes around the wide loads' group boundaries - plain canonical runs against the oracle."""
    import torch
    rng = np.random.default_rng(65)
    n = 400_009
    codes = rng.integers(0, 4, size=n + 8).astype(np.uint8)
    for s in range(0, n, 40_000):
        m = min(15_000, n - s)
        codes[s:s + m] = rng.integers(0, 2, size=m) * 3            # A / G only: ties everywhere, the vote near its threshold
        m2 = max(0, min(8_000, n - s - 20_000))
        codes[s + 20_000:s + 20_000 + m2] = rng.integers(0, 2, size=m2) + 1   # C / T only
    ws = gpu
    checked = 0
    try:
        for off in (0, 1, 2, 3):
            packed = np.zeros((n + off + 3) // 4 + 64, dtype=np.uint8)
            shifted = np.concatenate([np.zeros(off, dtype=np.uint8), codes[:n]])
            for j in range(4):
                c = shifted[j::4]
                packed[: len(c)] |= (c << (2 * j)).astype(np.uint8)
            d = torch.from_numpy(packed).cuda()
            out = torch.zeros(n, dtype=torch.int32, device="cuda")
            for (k, w) in ((19, 49), (19, 65), (21, 47), (19, 51), (21, 63), (19, 33), (17, 17)):
                want = oracle.run(packed, n, k, w, canonical=True, base_offset=off)
                for nb in (0, 7, 9):
                    ws.set_blocks_per_lane(nb)
                    c = sm.canonical_minimizers(k, w).run_device(d, n, out, base_offset=off)
                    assert c == len(want) and np.array_equal(_dev(out, c), want), (k, w, off, nb, c, len(want))
                    checked += 1
    finally:
        ws.set_blocks_per_lane(0)
    assert checked == 4 * 7 * 3


def test_skip_ambiguous_chunked_window_bits(sm, oracle, gpu):
    """Late round 5: for w >= 38 the dirty walk takes its window bits in CHUNKS of sixteen dwords per lane (rows in LDS, one
    buffer, (512 - 31) / w blocks per chunk).  Every edge of that scheme against the oracle: the smallest and largest window
    sizes it serves, lanes shorter than / equal to / one block longer than a chunk and several chunks long, base and bit
    offsets (the lanes' alignments inside their first dword), window-range shards (lanes starting anywhere), Ns every few
    dozen bases so that nearly every dword of the bits matters, and reads mode (one fresh chunk per read)."""
    import torch
    rng = np.random.default_rng(512)
    ws = gpu
    checked = 0
    try:
        for (k, w) in ((20, 38), (31, 41), (31, 51), (22, 64), (21, 95), (20, 96)):
            assert (k + w - 1) % 2 == 1
            per_chunk = (512 - 31) // w
            for off in (0, 3):
                n = 700_001 + 13 * off
                a = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n + off + 8)].copy()
                a[rng.integers(0, n, size=n // 90)] = ord("N")                 # windows of l = 57 .. 114: most are skipped ...
                for s in rng.integers(0, n - 5000, size=30):                   # ... except in clean stretches
                    a[s:s + int(rng.integers(300, 4000))] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=1)]
                for s in rng.integers(0, n - 5000, size=30):
                    m = int(rng.integers(300, 4000))
                    a[s:s + m] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=m)]
                packed, amb = oracle.pack_ascii_n(a.tobytes())
                d_p, d_m = torch.from_numpy(packed).cuda(), torch.from_numpy(amb).cuda()
                out = torch.zeros(n + 8, dtype=torch.int32, device="cuda")
                b = sm.canonical_minimizers(k, w)
                want = oracle.run_skip_ambiguous(packed, amb, n, k, w, base_offset=off, amb_offset=off)
                assert len(want) > 1000, (k, w, len(want))
                for nb in (0, 1, per_chunk - 1, per_chunk, per_chunk + 1, 2 * per_chunk, 2 * per_chunk + 1):
                    if nb < 0 or (nb == 0 and off):
                        continue
                    ws.set_blocks_per_lane(max(nb, 1) if nb else 0)
                    c = b.run_skip_ambiguous_device(d_p, d_m, n, out, base_offset=off, amb_offset=off)
                    assert ws.last_path() == sm.PATH_FUSED
                    assert c == len(want) and np.array_equal(_dev(out, c), want), (k, w, off, nb, c, len(want))
                    checked += 1
                ws.set_blocks_per_lane(0)
                # window-range shards concatenate to the whole (a shard dedups against the window before it)
                nw = n - (k + w - 1) + 1
                cuts = [0, nw // 3 + 5, 2 * nw // 3 - 7, nw]
                parts = []
                for s, e in zip(cuts[:-1], cuts[1:]):
                    cc = b.run_skip_ambiguous_device(d_p, d_m, n, out, base_offset=off, amb_offset=off, win_begin=s, win_end=e)
                    parts.append(_dev(out, cc).copy())
                assert np.array_equal(np.concatenate(parts), want), (k, w, off)
                checked += 1
        # reads mode: one fresh chunk per read, reads of a few blocks
        for (k, w, n_reads, stride, read_len, off) in ((31, 51, 700, 401, 400, 1), (20, 38, 500, 300, 250, 0)):
            span = n_reads * stride + 64 + off
            a = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=span)].copy()
            a[rng.integers(0, span, size=span // 150)] = ord("N")
            packed, amb = oracle.pack_ascii_n(a.tobytes())
            d_p, d_m = torch.from_numpy(packed).cuda(), torch.from_numpy(amb).cuda()
            out = torch.zeros(n_reads * read_len, dtype=torch.int32, device="cuda")
            offs = torch.zeros(n_reads + 1, dtype=torch.int64, device="cuda")
            total = sm.run_reads_device(sm.Builder(k, w, True, 0), d_p, n_reads, stride, read_len, out, offs,
                                        base_offset=off, d_amb=d_m, amb_offset=off)
            assert ws.last_path() == sm.PATH_FUSED
            ho, hp = offs.cpu().numpy(), _dev(out, total)
            assert ho[-1] == total and total > 0
            for r in range(n_reads):
                want = oracle.run_skip_ambiguous(packed, amb, read_len, k, w, base_offset=off + r * stride,
                                                 amb_offset=off + r * stride)
                assert np.array_equal(hp[ho[r]:ho[r + 1]], want), (k, w, r)
            checked += 1
    finally:
        ws.set_blocks_per_lane(0)
    assert checked == 6 * (7 + 6 + 2) + 2


def test_bench_single_process_device_group_line(sm, gpu):
    """`bench.py --gpus 2 --single-process`: the several-device C ABI (mm_device_group_*) driven from ONE process - here two
    entries that share this GPU, a functional check of the line: the strong split of one sequence, the entries' outputs
    summing to the whole run's, and the gather timed beside it."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    n = 200_000_000
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--single-process", "--steps", "3",
                        "--warmup", "2", "--bases", str(n)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    cfg = line["config"]
    assert cfg["devices"] == [0, 0] and cfg["distinct_devices"] == 1
    assert sum(cfg["outputs_per_entry"]) == cfg["outputs"] and abs(cfg["outputs"] / n - 1 / 6) < 0.01
    assert cfg["gather_ms"] > 0
