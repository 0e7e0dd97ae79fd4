"""Randomised differential test of the HIP path against the oracle: plans, lengths around tile
boundaries, base offsets, unaligned pointers, window ranges, lane lengths, super-k-mers, batch,
reads and skip-ambiguous entry points.  Seeded: a failure prints the case."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

_PREBUILT = {}


def _prebuilt(canonical):
    """window sizes with a prebuilt instance, from the library itself (mm_prebuilt_window_sizes)"""
    if canonical not in _PREBUILT:
        import simd_minimizers_amd as sm
        _PREBUILT[canonical] = sm.prebuilt_window_sizes(canonical)
        assert len(_PREBUILT[canonical]) >= 35
    return _PREBUILT[canonical]


def _plan(rng):
    canonical = bool(rng.integers(0, 2))
    w = int(rng.choice(_prebuilt(canonical)))
    mode = int(rng.choice([0, 0, 0, 1, 2]))
    if mode == 2 and w % 2 == 0:
        mode = 1
    k = int(rng.integers(1, 65))
    if canonical and (k + w - 1) % 2 == 0:
        k = k + 1 if k < 64 else k - 1
    return k, w, canonical, mode


def _length(rng, k, w):
    l = k + w - 1
    kind = rng.integers(0, 6)
    if kind == 0:
        return int(rng.integers(0, l + 3))
    if kind == 1:
        return int(rng.integers(l, 2000))
    if kind == 2:  # around one tile of the default geometry (256 lanes x ~12..70 blocks x w)
        return int(256 * w * rng.integers(10, 40) + l - 1 + rng.integers(-3, 4))
    return int(rng.integers(2000, 400_000))


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8])
def test_fuzz_device_runs(sm, oracle, gpu, seed):
    import torch
    rng = np.random.default_rng(1000 + seed + int(os.environ.get("MM_FUZZ_OFFSET", "0")))  # other streams on demand
    tally = dict(cases=0, bases=0, positions=0, range_checks=0, sk=0)
    for it in range(150):
        k, w, canonical, mode = _plan(rng)
        n = _length(rng, k, w)
        off = int(rng.integers(0, 40))
        shift = int(rng.integers(0, 4))  # pointer misalignment in bytes
        data = oracle.gen_packed(int(rng.integers(1 << 30)), off + n + 64)
        dev = torch.zeros(len(data) + 8, dtype=torch.uint8, device="cuda")
        dev[shift: shift + len(data)] = torch.from_numpy(data).cuda()
        d = dev[shift:]
        l = k + w - 1
        nw = max(0, n - l + 1)
        use_sk = mode == 0 and bool(rng.integers(0, 3) == 0)
        gpu.set_blocks_per_lane(int(rng.choice([0, 0, 0, 1, 2, 5, 9, 17])))
        b = sm.Builder(k, w, canonical, mode)
        out = torch.full((nw + 8,), -7, dtype=torch.int32, device="cuda")
        sk = torch.full((nw + 8,), -7, dtype=torch.int32, device="cuda") if use_sk else None
        case = dict(seed=seed, it=it, k=k, w=w, canonical=canonical, mode=mode, n=n, off=off, shift=shift, sk=use_sk)
        try:
            if use_sk:
                want, wsk = oracle.run(data, n, k, w, canonical=canonical, mode=mode, base_offset=off, super_kmers=True)
            else:
                want = oracle.run(data, n, k, w, canonical=canonical, mode=mode, base_offset=off)
            c = b.run_device(d, n, out, out_sk=sk, base_offset=off)
            assert c == len(want), case
            assert np.array_equal(out[:c].cpu().numpy().view(np.uint32), want), case
            assert int(out[c].item()) == -7, case  # nothing written past the count
            tally["cases"] += 1
            tally["bases"] += n
            tally["positions"] += c
            if use_sk:
                assert np.array_equal(sk[:c].cpu().numpy().view(np.uint32), wsk), case
                tally["sk"] += 1
            # a window sub-range equals the matching slice of the per-window stream collected from it
            if nw > 2 and mode != 0:
                a, e = sorted(int(x) for x in rng.integers(0, nw + 1, size=2))
                cc = b.run_device(d, n, out, base_offset=off, win_begin=a, win_end=e)
                sub = want[(want >= a) & (want < e)]
                assert np.array_equal(out[:cc].cpu().numpy().view(np.uint32), sub), (case, a, e)
                tally["range_checks"] += 1
        finally:
            gpu.set_blocks_per_lane(0)
    print("fuzz tally", seed, tally)
    # the loop must have done real work: every case ran, millions of bases, non-trivial outputs
    assert tally["cases"] == 150 and tally["bases"] > 5_000_000 and tally["positions"] > 300_000
    assert tally["range_checks"] >= 10 and tally["sk"] >= 10


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_fuzz_batch_reads_and_ambiguous(sm, oracle, gpu, seed):
    import torch
    rng = np.random.default_rng(2000 + seed + int(os.environ.get("MM_FUZZ_OFFSET", "0")))
    tally = dict(batches=0, batch_bases=0, reads=0, reads_checked=0, ambiguous=0, ambiguous_bases=0)
    for it in range(60):
        k, w, canonical, mode = _plan(rng)
        # ---- batch of a few sequences in one buffer
        n_seq = int(rng.integers(1, 7))
        lens = [_length(rng, k, w) for _ in range(n_seq)]
        lens = [min(x, 150_000) for x in lens]
        gaps = [int(rng.integers(0, 9)) for _ in range(n_seq)]
        starts = np.concatenate([[0], np.cumsum(np.array(lens) + np.array(gaps))])[:n_seq]
        total = int(starts[-1] + lens[-1]) + 128
        data = oracle.gen_packed(int(rng.integers(1 << 30)), total)
        big = torch.from_numpy(data).cuda()
        d = [big[int(s0) // 4:] for s0 in starts]
        offs_b = [int(s0) % 4 for s0 in starts]
        out = torch.zeros(sum(lens) + 64, dtype=torch.int32, device="cuda")
        b = sm.Builder(k, w, canonical, mode)
        case = dict(seed=seed, it=it, k=k, w=w, canonical=canonical, mode=mode, lens=lens)
        offs = sm.run_batch_device(b, d, lens, out, None, base_offsets=offs_b)
        host = out[: offs[-1]].cpu().numpy().view(np.uint32)
        for i in range(n_seq):
            want = oracle.run(data, lens[i], k, w, canonical=canonical, mode=mode, base_offset=int(starts[i]))
            assert np.array_equal(host[offs[i]:offs[i + 1]], want), (case, i)
        tally["batches"] += 1
        tally["batch_bases"] += sum(lens)
        # ---- reads (minimizer plans) and skip-ambiguous (canonical plans)
        if mode == 0:
            n_reads = int(rng.integers(1, 700))
            read_len = int(rng.integers(k + w - 1, k + w + 260))
            if it % 3 == 0:  # (round 6) reads across and far above a lane's length: the lane-table launch, mixed with short ones
                n_reads = int(rng.integers(1, 70))
                read_len = int(rng.integers(k + w + 200, 7000))
                tally["long_read_batches"] = tally.get("long_read_batches", 0) + 1
            stride = read_len + int(rng.integers(0, 9))
            span = n_reads * stride + 64
            a = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=span)].copy()
            if canonical:
                a[rng.integers(0, span, size=max(1, span // 300))] = ord("N")
            packed, amb = oracle.pack_ascii_n(a.tobytes())
            d_p, d_m = torch.from_numpy(packed).cuda(), torch.from_numpy(amb).cuda()
            lens_r = rng.integers(0, read_len + 1, size=n_reads)
            d_lens = torch.from_numpy(lens_r.astype(np.int32)).cuda()
            outr = torch.zeros(n_reads * read_len + 8, dtype=torch.int32, device="cuda")
            offr = torch.zeros(n_reads + 1, dtype=torch.int64, device="cuda")
            tot = sm.run_reads_device(b, d_p, n_reads, stride, read_len, outr, offr, read_lens=d_lens,
                                      d_amb=d_m if canonical else None)
            tally["lane_table_runs"] = tally.get("lane_table_runs", 0) + int(gpu.last_lane_table())
            ho = offr.cpu().numpy()
            hp = outr[:tot].cpu().numpy().view(np.uint32)
            for r in rng.integers(0, n_reads, size=min(n_reads, 40)):
                m = int(lens_r[r])
                if canonical:
                    want = oracle.run_skip_ambiguous(packed, amb, m, k, w, base_offset=int(r) * stride,
                                                     amb_offset=int(r) * stride)
                else:
                    want = oracle.run(packed, m, k, w, canonical=False, base_offset=int(r) * stride)
                assert np.array_equal(hp[ho[r]:ho[r + 1]], want), (case, "read", int(r), m, read_len, stride)
                tally["reads_checked"] += 1
            tally["reads"] += n_reads
        if canonical:
            n = min(_length(rng, k, w), 200_000)
            a = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n + 8)].copy()
            if n:
                a[rng.integers(0, n, size=max(1, n // 200))] = ord("N")
                s0 = int(rng.integers(0, n))
                a[s0:s0 + int(rng.integers(1, 300))] = ord("N")
            packed, amb = oracle.pack_ascii_n(a.tobytes())
            d_p, d_m = torch.from_numpy(packed).cuda(), torch.from_numpy(amb).cuda()
            outa = torch.zeros(n + 8, dtype=torch.int32, device="cuda")
            c = b.run_skip_ambiguous_device(d_p, d_m, n, outa)
            want = oracle.run_skip_ambiguous(packed, amb, n, k, w, mode=mode)
            assert np.array_equal(outa[:c].cpu().numpy().view(np.uint32), want), (case, "skip-ambiguous", n)
            tally["ambiguous"] += 1
            tally["ambiguous_bases"] += n
    print("fuzz tally", seed, tally)
    assert tally["batches"] == 60 and tally["batch_bases"] > 2_000_000
    assert tally["reads_checked"] > 300 and tally["ambiguous"] >= 15 and tally["ambiguous_bases"] > 500_000
    assert tally.get("long_read_batches", 0) >= 5 and tally.get("lane_table_runs", 0) >= 5
