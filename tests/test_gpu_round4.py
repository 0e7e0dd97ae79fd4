"""Round-4 GPU tests (all through the C ABI, bit-exact against the oracle):

* every PREBUILT kernel binary - each window size the library reports, canonical and forward, the four flavours
  (minimizers, closed / open syncmers, minimizers + super-k-mer indices) and the reads-mode instances - on an input
  of several tiles with a ragged last one (shape of src/test.rs:18-51);
* tie-heavy sequences (two-letter alphabets, tandem repeats with mutations, homopolymer runs): the lazy strand
  vote's slow path and the list-overflow redo, against the oracle, with window ranges that end inside tiles
  (was tools/gpu_vote_stress.py);
* the one-pass FASTA packer against the three-pass kernels on random texts
  (was tools/gpu_fasta_stress.py);
* the epoch-tagged look-back words (no clears between launches): interleaved geometries, plans and entry points on
  ONE workspace, and the wrap of the 16-bit epoch;
* one stream operation per run: counts through the kernel's own stores (device word and page-locked host word).
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(out, c):
    return out[:c].cpu().numpy().view(np.uint32)


def _range_expect(oracle, data, n, k, w, canonical, a, e):
    """positions a run over windows [a, e) returns: the per-window stream of that range, deduplicated against the
    window before it (src/collect.rs:265-271 applied at the seam)"""
    per_window = oracle.window_positions(data, n, k, w, oracle.default_hasher(canonical), canonical)
    sub = per_window[a:e]
    keep = np.ones(len(sub), dtype=bool)
    keep[1:] = sub[1:] != sub[:-1]
    if a > 0 and len(sub):
        keep[0] = sub[0] != per_window[a - 1]
    return sub[keep]


def test_every_prebuilt_instance_vs_oracle(sm, oracle, gpu):
    import torch
    n = 250_007  # a few tiles of every geometry and a ragged last one
    data = oracle.gen_packed(77, n + 64)
    d = torch.from_numpy(data).cuda()
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    sk = torch.zeros(n, dtype=torch.int32, device="cuda")
    checked = 0
    for canonical in (True, False):
        sizes = sm.prebuilt_window_sizes(canonical)
        assert len(sizes) >= 35
        for w in sizes:
            k = 20 if (canonical and w % 2 == 0) else 21
            for mode in (0, 1, 2):
                if mode == 2 and w % 2 == 0:
                    continue  # open syncmers need odd w (src/syncmers.rs:24-29)
                b = sm.Builder(k, w, canonical, mode)
                want = oracle.run(data, n, k, w, canonical=canonical, mode=mode, base_offset=1)
                c = b.run_device(d, n, out, base_offset=1)
                assert gpu.last_path() == sm.PATH_FUSED
                assert c == len(want) and np.array_equal(_dev(out, c), want), (w, canonical, mode)
                checked += 1
            want, wsk = oracle.run(data, n, k, w, canonical=canonical, base_offset=1, super_kmers=True)
            c = sm.Builder(k, w, canonical, 0).run_device(d, n, out, out_sk=sk, base_offset=1)
            assert c == len(want) and np.array_equal(_dev(out, c), want) and np.array_equal(_dev(sk, c), wsk), (w, canonical)
            checked += 1
    assert checked >= 2 * 35 * 3


def test_every_prebuilt_instance_window_ranges(sm, oracle, gpu):
    """window ranges that begin and end inside tiles (the shard-sized runs of the multi-GPU path), every instance"""
    import torch
    n = 180_011
    data = oracle.gen_packed(78, n + 64)
    d = torch.from_numpy(data).cuda()
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    rng = np.random.default_rng(4)
    for canonical in (True, False):
        for w in sm.prebuilt_window_sizes(canonical):
            k = 20 if (canonical and w % 2 == 0) else 21
            nw = n - (k + w - 1) + 1
            a, e = sorted(int(x) for x in rng.integers(1, nw, size=2))
            want = _range_expect(oracle, data, n, k, w, canonical, a, e)
            c = sm.Builder(k, w, canonical, 0).run_device(d, n, out, win_begin=a, win_end=e)
            assert c == len(want) and np.array_equal(_dev(out, c), want), (w, canonical, a, e)


def _tie_heavy(rng, n):
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    yield "two letters A/C", np.frombuffer(b"AC", dtype=np.uint8)[rng.integers(0, 2, n)]
    yield "two letters A/T", np.frombuffer(b"AT", dtype=np.uint8)[rng.integers(0, 2, n)]
    yield "two letters G/T", np.frombuffer(b"GT", dtype=np.uint8)[rng.integers(0, 2, n)]
    unit = acgt[rng.integers(0, 4, 37)]
    rep = np.tile(unit, n // 37 + 1)[:n].copy()
    mut = rng.integers(0, 200, n) == 0
    rep[mut] = acgt[rng.integers(0, 4, int(mut.sum()))]
    yield "37-base repeat, 0.5 % mutations", rep
    r = acgt[rng.integers(0, 4, n)].copy()
    r[(np.arange(n) // 5000) % 3 == 0] = ord("A")
    yield "random with 5 kbp poly-A runs", r


def test_tie_heavy_sequences_vs_oracle(sm, oracle, gpu):
    """The lazy strand vote decides only where the leftmost and the rightmost minimum differ (a tie of the 16 hash
    bits): about one window in 7 000 of a random sequence, most windows of these."""
    import torch
    rng = np.random.default_rng(11)
    n = 1_200_011
    slow_path_windows = 0
    for name, ascii_seq in _tie_heavy(rng, n):
        data = oracle.pack_ascii(ascii_seq.tobytes())
        data = np.concatenate([data, np.zeros(64, dtype=np.uint8)])
        d = torch.from_numpy(data).cuda()
        out = torch.zeros(n, dtype=torch.int32, device="cuda")
        for (k, w, mode) in ((21, 11, 0), (20, 12, 0), (15, 17, 1), (15, 17, 2), (31, 33, 0), (31, 51, 0), (20, 36, 0)):
            b = sm.Builder(k, w, True, mode)
            want = oracle.run(data, n, k, w, canonical=True, mode=mode)
            c = b.run_device(d, n, out)
            assert c == len(want) and np.array_equal(_dev(out, c), want), (name, k, w, mode)
            if mode == 0:
                nw = n - (k + w - 1) + 1
                a, e = 12_345, nw // 3 + 777
                wr = _range_expect(oracle, data, n, k, w, True, a, e)
                c = b.run_device(d, n, out, win_begin=a, win_end=e)
                assert c == len(wr) and np.array_equal(_dev(out, c), wr), (name, k, w, "range")
                # how often the two minima differ on this input (the vote's slow path): forward-leftmost != canonical
                fwd = oracle.window_positions(data, n, k, w, oracle.default_hasher(True), False)
                can = oracle.window_positions(data, n, k, w, oracle.default_hasher(True), True)
                slow_path_windows += int((fwd != can).sum())
    assert slow_path_windows > 100_000  # the inputs did reach the slow path, massively


def _random_fasta(rng, n):
    t = np.frombuffer(b"ACGTacgtNn", dtype=np.uint8)[rng.integers(0, 10, n)].copy()
    i = np.arange(n)
    width = int(rng.choice([7, 20, 31, 60, 61, 70, 80, 150, 1000, 100000]))
    crlf = rng.integers(0, 4) == 0
    step = width + (2 if crlf else 1)
    t[i % step == step - 1] = 10
    if crlf:
        t[i % step == step - 2] = 13
    every = int(rng.choice([0, 1, 2, 3, 17, 1000]))
    if every:
        t[i % (step * every) == 0] = ord(">")
    if rng.integers(0, 3) == 0:  # stray control characters and '>' inside lines
        m = rng.integers(0, 5000, n) == 0
        t[m] = np.array([9, 0, 11, 62, 13], dtype=np.uint8)[rng.integers(0, 5, int(m.sum()))]
    if rng.integers(0, 4):
        t[0] = ord(">")
    return t


def test_fasta_packers_agree_on_random_texts(sm, oracle, gpu, monkeypatch, exp_build):
    """two-pass packer == one-pass packer == three-pass kernels (packed bytes, record tables, counts) on random texts around chunk
    multiples (tests/test_gpu_fasta.py holds both against the oracle's reader)"""
    import torch
    rng = np.random.default_rng(7)
    L = sm.lib()
    ws = sm.Workspace(0, torch.cuda.current_stream().cuda_stream)

    def run(t_dev, flav):
        if flav is None:  # (the default: the two passes of mask arithmetic, mm_fasta2.hip)
            monkeypatch.delenv("MM_FASTA_ONEPASS", raising=False)
        else:
            monkeypatch.setenv("MM_FASTA_ONEPASS", flav)
        n = t_dev.numel()
        packed = torch.zeros(n // 4 + 64, dtype=torch.uint8, device="cuda")
        cap = n // 2 + 2
        rb = torch.zeros(cap + 1, dtype=torch.int64, device="cuda")
        rp = torch.zeros(cap, dtype=torch.int64, device="cuda")
        cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
        out = (C.c_uint64 * 2)()
        sm._check(L.mm_fasta_pack_device(ws.h, C.c_void_p(t_dev.data_ptr()), n, C.c_void_p(packed.data_ptr()),
                                         packed.numel() // 4 * 4, C.c_void_p(rb.data_ptr()), C.c_void_p(rp.data_ptr()),
                                         cap, C.c_void_p(cnt.data_ptr()), out))
        nb, nr = int(out[0]), int(out[1])
        return nb, nr, packed[: (nb + 3) // 4].cpu().numpy(), rb[: nr + 1].cpu().numpy(), rp[:nr].cpu().numpy()

    compared = 0
    for it in range(40):
        base = int(rng.choice([16384, 32768, 65536, 1 << 20, 1 << 22]))
        n = max(1, base * int(rng.integers(1, 4)) + int(rng.integers(-40, 40)))
        t = _random_fasta(rng, n)
        td = torch.from_numpy(t).cuda()
        a = run(td, "0")
        for b in (run(td, "1"), run(td, None)):
            assert a[0] == b[0] and a[1] == b[1], (it, n, a[:2], b[:2])
            assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4]), (it, n)
        compared += 1
    monkeypatch.delenv("MM_FASTA_ONEPASS", raising=False)
    ws.close()
    assert compared == 40


def test_epoch_tagged_status_words(sm, oracle, gpu):
    """The look-back status words are not cleared between launches (each launch tags its words with its epoch): runs
    of different geometry, plan and entry point interleaved on ONE workspace must not see one another's words."""
    import torch
    ws = sm.Workspace(0, torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(3)
    n = 1_500_013
    data = oracle.gen_packed(9, n + 64)
    d = torch.from_numpy(data).cuda()
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    plans = [(21, 11, True, 0), (21, 11, False, 0), (31, 51, True, 0), (15, 17, True, 1), (5, 3, False, 0)]
    wants = {p: oracle.run(data, n, p[0], p[1], canonical=p[2], mode=p[3]) for p in plans}
    for it in range(60):
        p = plans[int(rng.integers(0, len(plans)))]
        ws.set_blocks_per_lane(int(rng.choice([0, 0, 1, 2, 7, 13])))  # the tiles (and their words) move
        b = sm.Builder(*p).workspace(ws)
        if it % 3 == 0:
            c = b.run_device(d, n, out)
        else:
            b.run_device(d, n, out, sync=False, d_count=cnt)
            c = int(cnt.item())
        assert c == len(wants[p]) and np.array_equal(_dev(out, c), wants[p]), (it, p)
        if it % 7 == 0:  # reads mode and a batch launch share the same words
            offs = torch.zeros(41, dtype=torch.int64, device="cuda")
            tot = sm.run_reads_device(sm.Builder(21, 11, True, 0).workspace(ws), d, 40, 150, 150, out, offs)
            ho = offs.cpu().numpy()
            for r in (0, 17, 39):
                wr = oracle.run(data, 150, 21, 11, canonical=True, base_offset=150 * r)
                assert np.array_equal(_dev(out, tot)[ho[r]:ho[r + 1]], wr), (it, r)
    ws.set_blocks_per_lane(0)
    ws.check()
    ws.close()


def test_epoch_wraps(sm, oracle, gpu):
    """65 535 epochs fit the tag: the launch after that clears the words and starts over.  A long run, more than
    65 536 short ones, the long run again - its tiles meet words that carry their own epoch number from the first
    time, were it not for the clear."""
    import torch
    ws = sm.Workspace(0, torch.cuda.current_stream().cuda_stream)
    n = 2_000_003
    data = oracle.gen_packed(12, n + 64)
    d = torch.from_numpy(data).cuda()
    out = torch.zeros(n // 3, dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    b = sm.canonical_minimizers(21, 11).workspace(ws)
    want = oracle.run(data, n, 21, 11, canonical=True)
    c = b.run_device(d, n, out)
    assert c == len(want) and np.array_equal(_dev(out, c), want)
    small = oracle.run(data, 3000, 21, 11, canonical=True)
    for i in range(65_600):
        b.run_device(d, 3000, out, sync=False, d_count=cnt)
        if i % 8192 == 0:
            assert int(cnt.item()) == len(small)
    for _ in range(3):
        out.zero_()
        c = b.run_device(d, n, out)
        assert c == len(want) and np.array_equal(_dev(out, c), want)
    ws.check()
    ws.close()


def test_counts_without_copies(sm, oracle, gpu):
    """The kernel's last tile stores the run's total to the caller's device word and to the workspace's page-locked
    host word: synchronous and asynchronous runs, appending batches, empty runs and capacity errors agree with the
    oracle's counts."""
    import torch
    ws = sm.Workspace(0, torch.cuda.current_stream().cuda_stream)
    data = oracle.gen_packed(2, 400_000 + 64)
    d = torch.from_numpy(data).cuda()
    out = torch.zeros(200_000, dtype=torch.int32, device="cuda")
    cnt = torch.full((1,), -5, dtype=torch.int64, device="cuda")
    for n in (0, 5, 30, 31, 32, 1000, 77_777, 400_000):
        for canonical in (True, False):
            b = sm.Builder(21, 11, canonical, 0).workspace(ws)
            want = oracle.run(data, n, 21, 11, canonical=canonical)
            assert b.run_device(d, n, out) == len(want), (n, canonical)
            cnt.fill_(-5)
            b.run_device(d, n, out, sync=False, d_count=cnt)
            assert int(cnt.item()) == len(want), (n, canonical)
    # a capacity that is too small: the true count comes back with the error, nothing is written past the capacity
    b = sm.canonical_minimizers(21, 11).workspace(ws)
    want = oracle.run(data, 400_000, 21, 11, canonical=True)
    small = torch.full((1000 + 8,), -7, dtype=torch.int32, device="cuda")
    cnt64 = C.c_uint64()
    code = sm.lib().mm_run_device(b.plan().h, ws.h, C.c_void_p(d.data_ptr()), d.numel(), 0, 400_000, 0, sm.U64_MAX,
                                  C.c_void_p(small.data_ptr()), None, 1000, C.byref(cnt64))
    assert code == sm.ERR["CAPACITY"] and cnt64.value == len(want)
    assert np.array_equal(small[:1000].cpu().numpy().view(np.uint32), want[:1000]) and int(small[1000].item()) == -7
    ws.check()
    ws.close()


def test_bench_world2_line_is_strong(sm, gpu):
    """`bench.py --gpus 2` with no workload flag prints the strong split: ONE sequence, two window ranges, `bases`
    summing to the sequence (two ranks share this GPU over gloo: a functional check of the line, not a measurement)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["MM_BENCH_BACKEND"] = "gloo"
    n = 200_000_000
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--bases", str(n), "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    cfg = line["config"]
    assert cfg["bases_total"] == n and sum(cfg["windows_per_rank"]) == n - 31 + 1 and len(cfg["windows_per_rank"]) == 2
    assert abs(line["value"] - n * 3 / (line["ms_per_step"] * 3e-3) / 1e9) < 0.01 * line["value"]
    kinds = [e.get("scaling") for e in line.get("extra", [])]
    assert "weak" in kinds  # the weak figure rides along


def test_bench_extras_cannot_take_the_line_down(sm, gpu):
    """The two extra figures of the N > 1 default line run collectives of their own; a rank that hangs or fails inside
    them must not cost the headline.  MM_BENCH_EXTRA_TIMEOUT=0 makes the watchdog fire before they finish: rank 0 still
    prints the complete strong-split line, with an error entry in `extra`, and the job ends with exit code 0."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["MM_BENCH_BACKEND"] = "gloo"
    env["MM_BENCH_EXTRA_TIMEOUT"] = "0"
    n = 200_000_000
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--bases", str(n), "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert line["config"]["bases_total"] == n
    assert any("error" in e and "multi-GPU extras" in e.get("config", "") for e in line.get("extra", [])), line.get("extra")


def test_tapered_tail(sm, oracle, gpu, monkeypatch):
    """The last tiles of a launch walk half- and quarter-length lanes (plan_taper, mm_fused.hip).  MM_TAPER_SLOTS
    pretends the chip holds only a few workgroups, so that runs of a handful of tiles taper: every flavour, window
    ranges, lengths around the level boundaries, against the oracle; and a run large enough to taper on the real chip
    against the oracle's threaded port."""
    import torch
    rng = np.random.default_rng(17)
    n = 3_000_019
    data = oracle.gen_packed(41, n + 64)
    d = torch.from_numpy(data).cuda()
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    sk = torch.zeros(n, dtype=torch.int32, device="cuda")
    checked = 0
    for slots in (1, 2, 5):
        monkeypatch.setenv("MM_TAPER_SLOTS", str(slots))
        for (k, w, canonical, mode) in ((21, 11, True, 0), (21, 11, False, 0), (31, 51, True, 0), (15, 17, True, 1),
                                        (15, 17, True, 2), (21, 8, False, 0), (20, 32, True, 0), (5, 16, False, 0)):
            b = sm.Builder(k, w, canonical, mode)
            for m in (n, int(rng.integers(n // 3, n)), int(rng.integers(200_000, n // 3))):
                want = oracle.run(data, m, k, w, canonical=canonical, mode=mode)
                c = b.run_device(d, m, out)
                assert c == len(want) and np.array_equal(_dev(out, c), want), (slots, k, w, canonical, mode, m)
                checked += 1
            if mode == 0:
                want, wsk = oracle.run(data, n, k, w, canonical=canonical, super_kmers=True)
                c = b.run_device(d, n, out, out_sk=sk)
                assert c == len(want) and np.array_equal(_dev(out, c), want) and np.array_equal(_dev(sk, c), wsk)
                nw = n - (k + w - 1) + 1
                a, e = sorted(int(x) for x in rng.integers(1, nw, size=2))
                wr = _range_expect(oracle, data, n, k, w, canonical, a, e)
                c = b.run_device(d, n, out, win_begin=a, win_end=e)
                assert c == len(wr) and np.array_equal(_dev(out, c), wr), (slots, k, w, canonical, a, e)
                checked += 2
    # batches: the last round of the LAUNCH tapers, across the sequences' ends (fused_batch_tiles)
    for slots in (1, 3):
        monkeypatch.setenv("MM_TAPER_SLOTS", str(slots))
        for (k, w, canonical, mode) in ((21, 11, True, 0), (31, 51, True, 0), (21, 11, False, 0), (15, 17, True, 1)):
            lens = [900_001, 5, 1_200_000, 0, 333_333, 64_000, 500_017]
            starts = np.concatenate([[0], np.cumsum(np.array(lens) + 3)])[: len(lens)]
            seqs = [d[int(s0) // 4:] for s0 in starts]
            offs_b = [int(s0) % 4 for s0 in starts]
            b = sm.Builder(k, w, canonical, mode)
            offs = sm.run_batch_device(b, seqs, lens, out, base_offsets=offs_b)
            flat = out[: offs[-1]].cpu().numpy().view(np.uint32)
            for i, ln in enumerate(lens):
                want = oracle.run(data, ln, k, w, canonical=canonical, mode=mode, base_offset=int(starts[i]))
                assert np.array_equal(flat[offs[i]: offs[i + 1]], want), (slots, k, w, mode, i)
            checked += 1
    monkeypatch.delenv("MM_TAPER_SLOTS")
    assert checked >= 3 * (8 * 3 + 5 * 2) + 8
    # the real chip: 150 Mbp canonical k=21 w=11 tapers (threshold about 80 Mbp), element by element
    n2 = 150_000_001
    big = oracle.gen_packed(42, n2 + 64)
    want = oracle.run_fast(big, n2, 21, 11, canonical=True, threads=min(32, os.cpu_count() or 1))
    db = torch.from_numpy(big).cuda()
    ob = torch.zeros(int(n2 * 0.2), dtype=torch.int32, device="cuda")
    for env in (None, "1"):  # tapered and uniform tiles give the same positions
        if env:
            monkeypatch.setenv("MM_NO_TAPER", env)
        c = sm.canonical_minimizers(21, 11).run_device(db, n2, ob)
        assert c == len(want) and np.array_equal(_dev(ob, c), want), env
    monkeypatch.delenv("MM_NO_TAPER")


def test_values_u64_four_per_thread(sm, oracle, gpu):
    """values_u64 makes four values per thread since round 4: every count around the blocks of 1024 values, every
    length 1..32, both strands' choice, positions at the very end of the sequence, a position array that is only
    4-byte aligned and a sequence buffer at an odd byte offset - against the oracle (src/lib.rs:584-612)."""
    import torch
    rng = np.random.default_rng(5)
    n = 300_000
    host = oracle.gen_packed(6, n + 64)
    d = torch.from_numpy(host).cuda()
    checked = 0
    for ln in (1, 2, 5, 15, 16, 17, 21, 31, 32):
        for canonical in (True, False):
            for c in (1, 2, 3, 4, 5, 7, 1023, 1024, 1025, 2047, 4099, 50_001):
                pos = np.sort(rng.integers(0, n - ln + 1, size=c)).astype(np.uint32)
                pos[-1] = n - ln  # the last value ends with the sequence
                dp = torch.zeros(c + 3, dtype=torch.int32, device="cuda")
                shift = int(rng.integers(0, 3))  # the position array starts at any 4-byte boundary
                dp[shift: shift + c] = torch.from_numpy(pos.view(np.int32)).cuda()
                vals = torch.full((c + 2,), -1, dtype=torch.int64, device="cuda")
                sm._check(sm.lib().mm_values_u64_device_async(gpu.h, C.c_void_p(d.data_ptr()), d.numel(), 0, n, ln,
                                                              int(canonical), C.c_void_p(dp[shift:].data_ptr()), c,
                                                              C.c_void_p(vals.data_ptr())))
                gpu.sync()
                want = oracle.values_u64(host, ln, pos, canonical)
                got = vals.cpu().numpy()
                assert np.array_equal(got[:c].view(np.uint64), want), (ln, canonical, c)
                assert got[c] == -1 and got[c + 1] == -1, (ln, canonical, c)  # nothing past the count
                checked += 1
    # a sequence view that starts at an odd byte and a base offset
    for byte_off, base_off in ((1, 0), (3, 2), (2, 3)):
        m = n - 4 * byte_off - base_off
        pos = np.sort(rng.integers(0, m - 21 + 1, size=3001)).astype(np.uint32)
        dp = torch.from_numpy(pos.view(np.int32)).cuda()
        vals = torch.zeros(3001, dtype=torch.int64, device="cuda")
        sm._check(sm.lib().mm_values_u64_device_async(gpu.h, C.c_void_p(d[byte_off:].data_ptr()), d.numel() - byte_off,
                                                      base_off, m, 21, 1, C.c_void_p(dp.data_ptr()), 3001,
                                                      C.c_void_p(vals.data_ptr())))
        gpu.sync()
        want = oracle.values_u64(host, 21, pos, True, base_offset=4 * byte_off + base_off)
        assert np.array_equal(vals.cpu().numpy().view(np.uint64), want), (byte_off, base_off)
    assert checked == 9 * 2 * 12


def _group_device_lists():
    """Entries that share GPU 0 - what a one-GPU box can run - and, when the box has more GPUs, one entry per GPU and the
    pair (1, 0): hipMemcpyPeerAsync between distinct devices and a root that is not device 0."""
    import torch
    ngpu = torch.cuda.device_count()
    return [[0], [0, 0], [0, 0, 0]] + ([list(range(ngpu)), [1, 0]] if ngpu > 1 else [])


class _SameDevice:
    """Every exported function restores the calling thread's current device (ApiScope, mm_api.hip): checked around a block."""

    def __enter__(self):
        import torch
        self.dev = torch.cuda.current_device()
        return self

    def __exit__(self, *a):
        import torch
        assert torch.cuda.current_device() == self.dev, (torch.cuda.current_device(), self.dev)


def test_device_resident_shards(sm, oracle, gpu):
    """mm_device_group_upload / _adopt, mm_run_sharded_device, mm_device_group_result, mm_device_group_gather: the
    sequence resident on every entry's device, one asynchronous launch per entry, the positions left on the devices
    and gathered device-to-device.  Entries share this box's one GPU ({0, 0}, {0, 0, 0}): the shards laid end to end
    must BE the oracle's result (exact seam), for minimizers, super-k-mer indices and syncmers, odd base offsets and a
    tie-heavy sequence whose first shard overflows its expected capacity (re-run with the true count)."""
    import torch
    n = 5_000_037
    host = oracle.gen_packed(31, n + 64)
    for devices in _group_device_lists():
        root = f"cuda:{devices[0]}"  # (gather's destination lives on the root entry's device)
        with _SameDevice():
            g = sm.DeviceGroup(devices)
            g.upload(host[: (n + 3) // 4 + 1])
        for (k, w, canonical, mode, sk, off) in ((21, 11, True, 0, False, 0), (21, 11, False, 0, True, 3), (31, 51, True, 0, False, 1),
                                                 (15, 17, True, 1, False, 2), (15, 17, True, 2, False, 0)):
            m = n - off
            b = sm.Builder(k, w, canonical, mode)
            sk_list = []
            if sk:
                b = b.super_kmers(sk_list)
            with _SameDevice():
                counts = g.run_device(b, m, base_offset=off)
            if sk:
                want, wsk = oracle.run(host, m, k, w, canonical=canonical, mode=mode, base_offset=off, super_kmers=True)
            else:
                want = oracle.run(host, m, k, w, canonical=canonical, mode=mode, base_offset=off)
            assert sum(counts) == len(want), (devices, k, w, mode)
            dst = torch.full((len(want) + 8,), -3, dtype=torch.int32, device=root)
            dsk = torch.full((len(want) + 8,), -3, dtype=torch.int32, device=root) if sk else None
            torch.cuda.synchronize(root)
            with _SameDevice():
                tot = g.gather(0, dst, dsk)
            assert tot == len(want) and np.array_equal(_dev(dst, tot), want), (devices, k, w, mode)
            assert int(dst[tot].item()) == -3
            if sk:
                assert np.array_equal(_dev(dsk, tot), wsk)
            # the shards' own description: contiguous window ranges that cover every window, counts that add up
            nw = m - (k + w - 1) + 1
            prev_end = 0
            for i in range(len(devices)):
                dp, ds, cnt, wb, we = g.result(i)
                assert wb == prev_end and cnt == counts[i] and (dp != 0 or cnt == 0) and ((ds != 0) == sk or cnt == 0)
                prev_end = we
            assert prev_end == nw
            with pytest.raises(sm.MinimizerError) as e:
                g.gather(0, dst[: max(1, tot // 2)])
            assert e.value.code == sm.ERR["CAPACITY"]
        g.close()
    # adopted device buffers (the caller's own tensors) and a dense first shard: poly-A makes every window emit,
    # far above the expected density, so the shard is run again with the count the kernel reported
    asc = np.frombuffer(b"ACGT", dtype=np.uint8)[np.random.default_rng(2).integers(0, 4, 400_000)].copy()
    asc[:150_000] = ord("A")
    data = np.concatenate([oracle.pack_ascii(asc.tobytes()), np.zeros(64, dtype=np.uint8)])
    t = torch.from_numpy(data).cuda()
    g = sm.DeviceGroup([0, 0])
    g.adopt([t, t])
    counts = g.run_device(sm.minimizers(5, 3), 400_000)
    want = oracle.run(data, 400_000, 5, 3, canonical=False)
    dst = torch.zeros(len(want) + 1, dtype=torch.int32, device="cuda")
    assert g.gather(1, dst) == len(want) and np.array_equal(_dev(dst, len(want)), want) and counts[0] > 140_000
    g.close()
    # no resident sequence: refused
    g = sm.DeviceGroup([0])
    with pytest.raises(sm.MinimizerError):
        g.run_device(sm.minimizers(21, 11), 1000)
    g.close()


def _check_fastq(sm, oracle, text, max_records=1 << 16):
    import torch
    rec = sm.fasta_pack_device(text, max_records=max_records)
    want = oracle.fastq_records(bytes(text))
    assert len(rec) == len(want), (len(rec), len(want))
    packed = rec.packed.cpu().numpy()
    for i, (pos, name, seq) in enumerate(want):
        b, e = int(rec.base[i]), int(rec.base[i + 1])
        assert e - b == len(seq), (i, e - b, len(seq))
        assert int(rec.text_pos[i]) == pos, i
        if i < 50 or i % 997 == 0 or i >= len(want) - 3:
            codes = [(packed[(b + j) // 4] >> (2 * ((b + j) % 4))) & 3 for j in range(len(seq))]
            assert codes == [(c >> 1) & 3 for c in seq], i
    return rec


def test_fastq_packer(sm, oracle, gpu):
    """FASTQ text -> packed records on the device (mm_fastq.hip; mm_fasta_pack_device tells the formats apart by the
    first non-blank byte like needletail::parse_fastx_file): four-line records against the oracle's reader - CRLF, no
    trailing newline, empty reads, blank lines after the last record, reads longer than a chunk, sizes around the 4 KB
    chunks, an unaligned text pointer, the record-table limit; then reads of one length straight into the reads-mode
    kernel and of any lengths into a batch launch, every read's minimizers against the oracle."""
    import torch
    rng = np.random.default_rng(21)
    _check_fastq(sm, oracle, b"@r1 x\nACGT\n+\nIIII\n@r2\r\nTTGA\r\n+r2\r\nIIII\r\n@r3\n\n+\n\n@r4\nAC")
    _check_fastq(sm, oracle, b"@a\nAC\n+\nII\n\n\n")
    _check_fastq(sm, oracle, b"\n  @a\nACGTTTGA\n+\nIIIIIIII\n"[1:].lstrip())
    # several runs of sequence bytes inside one thread's 16 bytes: a stray '\r' inside a line, records of eight bytes
    _check_fastq(sm, oracle, b"@a\nAC\rGT\r\r\n+\nIIII\n" + b"@\nA\n+\n!\n@\nC\n+\n!\n" * 40 + b"@z\nGATTACA\n+\n!!!!!!!\n")
    acgt = np.frombuffer(b"ACGTNacgt", dtype=np.uint8)

    def make(n_reads, lens, crlf=False, final_newline=True):
        parts = []
        nl = b"\r\n" if crlf else b"\n"
        for r in range(n_reads):
            ln = int(lens[r])
            seq = acgt[rng.integers(0, 9, ln)].tobytes()
            qual = bytes(rng.integers(33, 74, ln).astype(np.uint8))
            parts.append(b"@read%d/1 len=%d" % (r, ln) + nl + seq + nl + (b"+" if r % 3 else b"+read%d" % r) + nl + qual + nl)
        text = b"".join(parts)
        return text if final_newline else text[: -len(nl)]
    for n_reads, lo, hi, crlf, fin in ((1, 1, 2, False, True), (7, 0, 30, False, False), (500, 100, 151, False, True),
                                       (300, 150, 151, True, True), (40, 3000, 9000, False, True), (3000, 0, 40, True, False)):
        text = make(n_reads, rng.integers(lo, hi, n_reads), crlf, fin)
        _check_fastq(sm, oracle, text)
    # more than one GROUP of 256 chunks (4 MB of text: the resolve step's two levels)
    _check_fastq(sm, oracle, make(27_000, rng.integers(140, 151, 27_000)), max_records=1 << 15)
    # lengths of the text around the chunk size, and an unaligned device pointer
    base = make(400, rng.integers(90, 151, 400))  # about 120 KB of text
    for cut in (4095, 4096, 4097, 8191, 8192, 12289, 16383, 16384, 16385, 32769, 65535, 65537):
        piece = base[:cut]
        _check_fastq(sm, oracle, piece)
    want_base = oracle.fastq_records(base)
    for off in (1, 2, 3):
        tt = torch.from_numpy(np.frombuffer(b"x" * off + base, dtype=np.uint8).copy()).cuda()
        rec = sm.fasta_pack_device(tt[off:], max_records=1 << 12)
        assert len(rec) == 400 and [int(x) for x in rec.text_pos[:5]] == [p for p, _, _ in want_base[:5]]
        assert rec.lengths() == [len(q) for _, _, q in want_base]
    with pytest.raises(sm.MinimizerError) as e:
        sm.fasta_pack_device(make(10, [5] * 10), max_records=4)
    assert e.value.code == sm.ERR["CAPACITY"]
    # FASTQ -> reads-mode kernel (one length) and -> batch launch (any lengths); every read against the oracle
    n_reads, ln = 2000, 150
    text = make(n_reads, [ln] * n_reads)
    rec = _check_fastq(sm, oracle, text)
    want = oracle.fastq_records(text)
    b = sm.canonical_minimizers(21, 11)
    out = torch.zeros(n_reads * ln, dtype=torch.int32, device="cuda")
    offs = torch.zeros(n_reads + 1, dtype=torch.int64, device="cuda")
    total = sm.run_reads_device(b, rec.packed, n_reads, ln, ln, out, offs)
    ho = offs.cpu().numpy()
    flat = out[:total].cpu().numpy().view(np.uint32)
    for r in (0, 1, 777, n_reads - 1):
        seq_packed = oracle.pack_ascii(want[r][2])
        w_r = oracle.run(np.concatenate([seq_packed, np.zeros(16, dtype=np.uint8)]), ln, 21, 11, canonical=True)
        assert np.array_equal(flat[ho[r]:ho[r + 1]], w_r), r
    text2 = make(300, rng.integers(31, 400, 300))
    rec2 = _check_fastq(sm, oracle, text2)
    want2 = oracle.fastq_records(text2)
    offs2 = sm.run_fasta_device(b, rec2, out)
    flat2 = out[: offs2[-1]].cpu().numpy().view(np.uint32)
    for r in (0, 5, 150, 299):
        seq_packed = oracle.pack_ascii(want2[r][2])
        w_r = oracle.run(np.concatenate([seq_packed, np.zeros(16, dtype=np.uint8)]), len(want2[r][2]), 21, 11, canonical=True)
        assert np.array_equal(flat2[offs2[r]: offs2[r + 1]], w_r), r


def test_device_resident_shares(sm, oracle, gpu):
    """mm_device_group_upload_range: every entry holds only the bytes its share of the N-way split reads (+ a halo; the
    rest of its buffer is a fill pattern), so the sequence crosses the host link once in total.  Runs of that shape with
    several plans - small and large windows, super-k-mers, syncmers, an odd base offset - equal the oracle; a run of
    another shape is refused with the ranges named."""
    import torch
    n, off = 40_000_003, 3
    host = oracle.gen_packed(17, n + off + 64)
    ngpu = torch.cuda.device_count()
    for devices in [[0, 0, 0, 0]] + ([list(range(ngpu))] if ngpu > 1 else []):
        _shares_on(sm, oracle, devices, host, n, off)


def _shares_on(sm, oracle, devices, host, n, off):
    import torch
    with _SameDevice():
        g = sm.DeviceGroup(devices)
        g.upload_range(host[: (n + off + 3) // 4 + 1], n, base_offset=off)
    root = f"cuda:{devices[0]}"
    for (k, w, canonical, mode, sk) in ((21, 11, True, 0, False), (31, 51, True, 0, False), (21, 11, False, 0, True),
                                        (15, 17, True, 1, False), (5, 3, False, 0, False)):
        b = sm.Builder(k, w, canonical, mode)
        sk_list = []
        if sk:
            b = b.super_kmers(sk_list)
        counts = g.run_device(b, n, base_offset=off)
        if sk:
            want, wsk = oracle.run(host, n, k, w, canonical=canonical, mode=mode, base_offset=off, super_kmers=True)
        else:
            want = oracle.run(host, n, k, w, canonical=canonical, mode=mode, base_offset=off)
        assert sum(counts) == len(want), (k, w, mode)
        dst = torch.zeros(len(want) + 1, dtype=torch.int32, device=root)
        dsk = torch.zeros(len(want) + 1, dtype=torch.int32, device=root) if sk else None
        torch.cuda.synchronize(root)
        with _SameDevice():
            assert g.gather(0, dst, dsk) == len(want)
        assert np.array_equal(_dev(dst, len(want)), want), (k, w, mode)
        if sk:
            assert np.array_equal(_dev(dsk, len(want)), wsk)
    with pytest.raises(sm.MinimizerError) as e:  # another shape: its shares lie elsewhere
        g.run_device(sm.minimizers(21, 11), n // 2, base_offset=off)
    assert e.value.code == sm.ERR["NULL"] and "holds bytes" in str(e.value)
    g.close()


def test_fastq_packer_large(sm, gpu):
    """More than 1 GiB of FASTQ text (more than 256 groups of 256 chunks: the second round of the resolve step's single
    workgroup), records of one shape, checked on the device: counts, every record's first base and byte offset, and the
    packed bases against mm_pack_ascii of the text's sequence bytes."""
    import ctypes as C

    import torch
    dev = torch.device("cuda:0")
    rl, name_len = 150, 19
    rec_bytes = (1 + name_len + 1) + (rl + 1) + 2 + (rl + 1)
    n_rec = int(1.1 * (1 << 30)) // rec_bytes
    n = n_rec * rec_bytes
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    t = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (n,), device=dev, generator=g)]
    t = t.view(n_rec, rec_bytes)
    t[:, 0] = ord("@")
    for p in (name_len + 1, name_len + 2 + rl, name_len + 2 + rl + 2, rec_bytes - 1):
        t[:, p] = 10
    t[:, name_len + 2 + rl + 1] = ord("+")
    seq_bytes = t[:, name_len + 2: name_len + 2 + rl].contiguous().view(-1)
    t = t.view(-1)
    ws = sm.default_workspace(0)
    L = sm.lib()
    packed = torch.empty(n_rec * rl // 4 + 64, dtype=torch.uint8, device=dev)
    rb = torch.zeros(n_rec + 1, dtype=torch.int64, device=dev)
    rp = torch.zeros(n_rec, dtype=torch.int64, device=dev)
    cnt = torch.zeros(2, dtype=torch.int64, device=dev)
    sm._check(L.mm_fastq_pack_device_async(ws.h, C.c_void_p(t.data_ptr()), n, C.c_void_p(packed.data_ptr()),
                                           packed.numel() // 4 * 4, C.c_void_p(rb.data_ptr()), C.c_void_p(rp.data_ptr()),
                                           n_rec, C.c_void_p(cnt.data_ptr())))
    ws.sync()
    ws.check()
    assert [int(x) for x in cnt.cpu()] == [n_rec * rl, n_rec]
    idx = torch.arange(n_rec + 1, device=dev, dtype=torch.int64)
    assert bool((rb == idx * rl).all()) and bool((rp == idx[:-1] * rec_bytes).all())
    want = torch.empty((n_rec * rl + 3) // 4, dtype=torch.uint8, device=dev)
    sm._check(L.mm_pack_ascii_device_async(ws.h, C.c_void_p(seq_bytes.data_ptr()), n_rec * rl, C.c_void_p(want.data_ptr())))
    ws.sync()
    assert bool((packed[: want.numel()] == want).all())


def test_device_resident_batches(sm, oracle, gpu):
    """mm_device_group_upload_batch / mm_run_batch_sharded_device / mm_device_group_gather_batch: independent sequences
    placed greedily on the entries, each resident on its entry's device only, one batch launch per entry, the positions
    left on the devices and gathered device-to-device in input order (north_star's "sharded by contig, gather of the
    positions"; entries share this box's one GPU).  Every sequence against the oracle."""
    import torch
    rng = np.random.default_rng(8)
    lens = [700_001, 40, 1_200_017, 0, 333_333, 90_000, 30, 450_123]
    offs = [0, 1, 3, 0, 2, 0, 0, 1]
    seqs = [oracle.gen_packed(300 + i, o + n + 64) for i, (n, o) in enumerate(zip(lens, offs))]
    # a dense one: poly-A emits at every window, far above the expected density of its entry
    seqs[5] = np.zeros_like(seqs[5])
    for devices in _group_device_lists():
        root = f"cuda:{devices[0]}"
        with _SameDevice():
            g = sm.DeviceGroup(devices)
            g.upload_batch([s[: (o + n + 3) // 4 + 1] for s, n, o in zip(seqs, lens, offs)])
        for (k, w, canonical, mode, sk) in ((21, 11, True, 0, False), (31, 51, True, 0, False), (5, 3, False, 0, True), (15, 17, True, 1, False)):
            b = sm.Builder(k, w, canonical, mode)
            if sk:
                b = b.super_kmers([])
            with _SameDevice():
                counts = g.run_batch_device(b, lens, base_offsets=offs)
            wants = [oracle.run(s, n, k, w, canonical=canonical, mode=mode, base_offset=o, super_kmers=sk) for s, n, o in zip(seqs, lens, offs)]
            wpos = [x[0] if sk else x for x in wants]
            assert counts == [len(x) for x in wpos], (devices, k, w, mode)
            dst = torch.full((sum(counts) + 4,), -3, dtype=torch.int32, device=root)
            dsk = torch.full((sum(counts) + 4,), -3, dtype=torch.int32, device=root) if sk else None
            torch.cuda.synchronize(root)
            with _SameDevice():
                o = g.gather_batch(0, dst, dsk)
            assert o[-1] == sum(counts) and int(dst[o[-1]].item()) == -3
            flat = _dev(dst, o[-1])
            for i in range(len(lens)):
                assert np.array_equal(flat[o[i]: o[i + 1]], wpos[i]), (devices, k, w, mode, i)
                if sk:
                    assert np.array_equal(_dev(dsk, o[-1])[o[i]: o[i + 1]], wants[i][1]), i
            with pytest.raises(sm.MinimizerError) as e:
                g.gather_batch(0, dst[: max(1, o[-1] // 3)])
            assert e.value.code == sm.ERR["CAPACITY"]
        g.close()


def test_packed_reads_of_any_lengths(sm, oracle, gpu):
    """mm_run_packed_reads_device: reads packed back to back (the FASTQ packer's layout) in ONE launch of the reads-mode
    kernel - lengths 0..400 over several tiles, super-k-mer indices, syncmers, a batch with one read too long for a lane
    (the per-read fallback), max_read_len below the longest read (cut like read_lens); every read against the oracle."""
    import torch
    rng = np.random.default_rng(33)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)

    def fastq(lens):
        parts = []
        for r, ln in enumerate(lens):
            seq = acgt[rng.integers(0, 4, int(ln))].tobytes()
            parts.append(b"@r%d\n" % r + seq + b"\n+\n" + b"I" * int(ln) + b"\n")
        return b"".join(parts)

    def check(lens, k, w, canonical, mode, sk=False, max_read_len=None, sample=None):
        text = fastq(lens)
        rec = sm.fasta_pack_device(text, max_records=len(lens) + 1)
        want = oracle.fastq_records(text)
        assert rec.lengths() == [int(x) for x in lens]
        b = sm.Builder(k, w, canonical, mode)
        out = torch.zeros(max(1, int(sum(lens))), dtype=torch.int32, device="cuda")
        osk = torch.zeros_like(out) if sk else None
        offs = torch.full((len(lens) + 1,), -1, dtype=torch.int64, device="cuda")
        total = sm.run_packed_reads_device(b, rec, out, offs, out_sk=osk, max_read_len=max_read_len)
        ho = offs.cpu().numpy()
        assert ho[0] == 0 and ho[-1] == total
        flat = out[:total].cpu().numpy().view(np.uint32)
        fsk = osk[:total].cpu().numpy().view(np.uint32) if sk else None
        for r in (sample if sample is not None else range(len(lens))):
            ln = int(lens[r]) if max_read_len is None else min(int(lens[r]), max_read_len)
            packed = np.concatenate([oracle.pack_ascii(want[r][2]), np.zeros(16, dtype=np.uint8)])
            res = oracle.run(packed, ln, k, w, canonical=canonical, mode=mode, super_kmers=sk)
            wp = res[0] if sk else res
            assert np.array_equal(flat[ho[r]: ho[r + 1]], wp), (r, ln, k, w, mode)
            if sk:
                assert np.array_equal(fsk[ho[r]: ho[r + 1]], res[1]), r
    lens = rng.integers(0, 401, 1500)
    lens[:4] = [0, 30, 31, 400]
    check(lens, 21, 11, True, 0, sample=list(range(0, 1500, 7)) + [1499])
    assert gpu.last_path() == sm.PATH_FUSED
    check(lens[:600], 21, 11, False, 0, sk=True, sample=range(0, 600, 5))
    check(lens[:300], 15, 17, True, 1, sample=range(0, 300, 3))
    check(rng.integers(100, 151, 700), 31, 19, True, 0, sample=range(0, 700, 9))
    check(lens[:200], 21, 11, True, 0, max_read_len=150, sample=range(0, 200, 3))  # longer reads are cut to 150
    long_lens = rng.integers(50, 200, 40)
    long_lens[17] = 70_001  # too long for one lane: one launch per read, same results
    check(long_lens, 21, 11, True, 0)
    check([], 21, 11, True, 0)


def test_packed_reads_from_host(sm, oracle, gpu):
    """mm_run_packed_reads_host: the one-call replacement of a per-read loop over Builder::run (src/lib.rs:378) - many short
    host sequences, one upload, one launch, one download; every read against the oracle, super-k-mer indices too."""
    rng = np.random.default_rng(44)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    reads = [acgt[rng.integers(0, 4, int(ln))].tobytes() for ln in list(rng.integers(0, 300, 900)) + [0, 30, 31, 1000]]
    for (k, w, canonical, mode, sk) in ((21, 11, True, 0, False), (21, 11, False, 0, True), (15, 17, True, 1, False)):
        b = sm.Builder(k, w, canonical, mode)
        pos, offs, skv = sm.run_reads_host(b, reads, super_kmers=sk)
        assert len(offs) == len(reads) + 1 and offs[0] == 0 and offs[-1] == len(pos)
        for r in list(range(0, len(reads), 11)) + [len(reads) - 4, len(reads) - 3, len(reads) - 2, len(reads) - 1]:
            packed = np.concatenate([oracle.pack_ascii(reads[r]), np.zeros(16, dtype=np.uint8)])
            res = oracle.run(packed, len(reads[r]), k, w, canonical=canonical, mode=mode, super_kmers=sk)
            wp = res[0] if sk else res
            assert np.array_equal(pos[offs[r]: offs[r + 1]], wp), (r, k, w, mode)
            if sk:
                assert np.array_equal(skv[offs[r]: offs[r + 1]], res[1]), r
    pos, offs, _ = sm.run_reads_host(sm.canonical_minimizers(21, 11), [])
    assert len(pos) == 0 and offs == [0]
