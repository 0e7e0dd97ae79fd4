"""GPU tests added in round 3 (all through the C ABI, bit-exact against the oracle):

* the split path of the fused family (MM_SPLIT=1: walk kernel + persistent expander on a second stream + redo
  pass): plain runs, window ranges, slice offsets, capacity, low-complexity input (every tile overflows),
  skip-ambiguous windows, super-k-mer indices and syncmers (walk kernels specialised at run time), the
  pipelined host entry point (append mode);
* tile status words sized for the whole-rounds tuner (one workspace reused over the ascending
  CHM13-like contig lengths with an exact-size allocation), the sticky error word after a synchronous redo,
  FASTQ text refused by the FASTA packer, capacity messages of the packer.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dev(out, c):
    return out[:c].cpu().numpy().view(np.uint32)


# ------------------------------------------------------------------ split path
@pytest.mark.parametrize("canon", [False, True])
def test_split_path_matches_oracle(sm, oracle, gpu, monkeypatch, canon, exp_build):
    """walk_kernel + expander (mm_split.hip) == oracle element by element; the fused kernel gives the same."""
    import torch
    monkeypatch.setenv("MM_SPLIT", "1")
    k, w = 21, 11
    b = sm.Builder(k, w, canon, 0)
    for n, off in ((31, 0), (200_003, 0), (2_000_003, 3), (16_000_001, 1)):
        data = oracle.gen_packed(40 + off, n + off)
        want = oracle.run(data, n, k, w, canonical=canon, base_offset=off)
        d = torch.from_numpy(data).cuda()
        out = torch.zeros(n // 3 + 64, dtype=torch.int32, device="cuda")
        c = b.run_device(d, n, out, base_offset=off)
        assert gpu.last_path() == sm.PATH_SPLIT
        assert c == len(want) and np.array_equal(_dev(out, c), want), (n, off)
    # window ranges: concatenation with the seam rule == the whole run (src/collect.rs:265-271)
    n = 3_000_017
    data = oracle.gen_packed(77, n)
    want = oracle.run(data, n, k, w, canonical=canon)
    d = torch.from_numpy(data).cuda()
    out = torch.zeros(n // 3, dtype=torch.int32, device="cuda")
    nw = n - (k + w - 1) + 1
    cuts = [0, 1, 777_777, 777_778, 2_000_000, nw]
    got = []
    for a, e in zip(cuts[:-1], cuts[1:]):
        c = b.run_device(d, n, out, win_begin=a, win_end=e)
        assert gpu.last_path() == sm.PATH_SPLIT
        part = _dev(out, c)
        if got and len(part) and got[-1] == part[0]:
            part = part[1:]
        got.extend(part.tolist())
    assert np.array_equal(np.array(got, dtype=np.uint32), want)
    # a capacity that is too small: the count is still the true one, nothing is written past the capacity
    small = torch.full((1000 + 16,), -1, dtype=torch.int32, device="cuda")
    with pytest.raises(sm.MinimizerError) as e:
        b.run_device(d, n, small[:1000])
    assert e.value.code == sm.ERR["CAPACITY"] and str(len(want)) in str(e.value)
    assert np.array_equal(_dev(small, 1000), want[:1000]) and int((small[1000:] != -1).sum().item()) == 0
    monkeypatch.setenv("MM_SPLIT", "0")
    c = b.run_device(d, n, out)
    assert gpu.last_path() == sm.PATH_FUSED and np.array_equal(_dev(out, c), want)


def test_split_path_redo_and_flavours(sm, oracle, gpu, monkeypatch, exp_build):
    """Low-complexity input (lists overflow: the redo pass), skip-ambiguous windows, and plans whose walk
    kernel is specialised at run time (super-k-mer indices, closed syncmers)."""
    import torch
    monkeypatch.setenv("MM_SPLIT", "1")
    n = 1_500_000
    rng = np.random.default_rng(9)
    packed = np.zeros((n + 3) // 4, dtype=np.uint8)                 # poly-A ...
    packed[n // 16: n // 8] = rng.integers(0, 256, n // 8 - n // 16, dtype=np.uint8)  # ... with a random stretch
    packed[n // 6: n // 5] = 0x1B                                   # ... and a 4-base repeat
    d = torch.from_numpy(packed).cuda()
    out = torch.zeros(n + 64, dtype=torch.int32, device="cuda")
    for canon in (False, True):
        for (k, w) in ((21, 11), (5, 11)):
            if canon and (k + w) % 2:
                continue
            want = oracle.run(packed, n, k, w, canonical=canon)
            c = sm.Builder(k, w, canon, 0).run_device(d, n, out)
            assert gpu.last_path() == sm.PATH_SPLIT
            assert c == len(want) and np.array_equal(_dev(out, c), want), (canon, k, w)
    # skip-ambiguous windows (canonical plans; the walk kernel carries the AMBI walk)
    n = 1_000_003
    text = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)].copy()
    text[rng.integers(0, n, 300)] = ord("N")
    text[400_000:400_900] = ord("N")
    pk, amb = oracle.pack_ascii_n(text.tobytes())
    k, w = 21, 11
    want = oracle.run_skip_ambiguous(pk, amb, n, k, w)
    b = sm.canonical_minimizers(k, w)
    dp, da = torch.from_numpy(pk).cuda(), torch.from_numpy(amb).cuda()
    c = b.run_skip_ambiguous_device(dp, da, n, out)
    assert gpu.last_path() == sm.PATH_SPLIT
    assert c == len(want) and np.array_equal(_dev(out, c), want)
    # super-k-mer indices and closed syncmers: walk kernels without a prebuilt instance
    n = 700_001
    data = oracle.gen_packed(3, n)
    d = torch.from_numpy(data).cuda()
    want_p, want_s = oracle.run(data, n, k, w, canonical=True, super_kmers=True)
    sk = torch.zeros(n // 3, dtype=torch.int32, device="cuda")
    c = sm.canonical_minimizers(k, w).run_device(d, n, out, out_sk=sk)
    assert gpu.last_path() == sm.PATH_SPLIT, sm.lib().mm_last_error()
    assert c == len(want_p) and np.array_equal(_dev(out, c), want_p) and np.array_equal(_dev(sk, c), want_s)
    want = oracle.run(data, n, 15, 17, canonical=True, mode=1)
    c = sm.canonical_closed_syncmers(15, 17).run_device(d, n, out)
    assert gpu.last_path() == sm.PATH_SPLIT, sm.lib().mm_last_error()
    assert c == len(want) and np.array_equal(_dev(out, c), want)


def test_split_path_host_pipeline(sm, oracle, gpu, monkeypatch, exp_build):
    """mm_run_host on a sequence long enough for the pipelined path (chunks appended back to back: the
    expander takes the running total as its carry)."""
    monkeypatch.setenv("MM_SPLIT", "1")
    n, k, w = 60_000_000, 21, 11
    data = oracle.gen_packed(12, n)
    want = oracle.run_fast(data, n, k, w, canonical=True, threads=8)
    got, _ = sm.canonical_minimizers(k, w)._run_arrays(sm.PackedSeq(data, 0, n))
    assert gpu.last_path() == sm.PATH_SPLIT
    assert len(got) == len(want) and np.array_equal(np.asarray(got, dtype=np.uint32), want)


# ------------------------------------------------------------------ ADVICE r2
_STATUS_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/oracle")
import simd_minimizers_amd as sm
lens = [45_090_682, 51_324_926, 61_707_364, 66_210_255, 80_542_538, 84_276_897, 96_330_374, 99_753_195,
        101_161_492, 113_566_686, 127_220_663, 133_324_548, 134_758_134, 135_127_769, 146_259_331,
        150_617_247, 154_259_566, 160_567_428, 172_126_628, 182_055_711, 193_574_945, 201_105_948,
        242_696_752, 248_387_328]
n_max = max(lens)
d = sm.generate_device(n_max, 5)
out = torch.zeros(int(n_max * 0.2), dtype=torch.int32, device="cuda")
ws = sm.Workspace(0, torch.cuda.current_stream().cuda_stream)
fresh = sm.Workspace(0, torch.cuda.current_stream().cuda_stream)
for (k, w) in ((21, 11), (21, 33), (31, 51)):
    b = sm.canonical_minimizers(k, w)
    for n in lens:
        c = b.workspace(ws).run_device(d, n, out)
        s = int(out[:c].to(torch.int64).sum().item())
        c2 = b.workspace(fresh).run_device(d, n, out)
        assert (c, s) == (c2, int(out[:c2].to(torch.int64).sum().item())), (k, w, n)
print("ok")
"""


def test_status_words_cover_the_tuned_grid(sm, gpu):
    """the tile status buffer was sized before the whole-rounds tuner could raise the tile
    count by more than the 20 % margin.  One workspace reused over the ascending CHM13-like contig lengths
    with an exact-size allocation (MM_STATUS_TIGHT) and MM_STATUS_STRICT (a launch whose tuned grid does not
    fit fails instead of falling back to the default lanes)."""
    env = dict(os.environ, MM_STATUS_TIGHT="1", MM_STATUS_STRICT="1")
    r = subprocess.run([sys.executable, "-c", _STATUS_SCRIPT, ROOT], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-4000:]


# (test_sticky_error_cleared_by_synchronous_redo: part of tests/test_gpu_round2.py::test_async_status_is_observable
# since round 4 - the MM_DEBUG=32 hook lives in the experiments build, which that test loads in a child process)


def test_fasta_packer_record_table_limit(sm, gpu):
    # (FASTQ was refused here until round 4: tests/test_gpu_round4.py::test_fastq_packer)
    rec = sm.fasta_pack_device(b">a\nACGT\n>b\nTT\n")
    assert len(rec) == 2
    with pytest.raises(sm.MinimizerError) as e:
        sm.fasta_pack_device(b">a\nACGT\n>b\nTT\n>c\nA\n", max_records=2)
    assert e.value.code == sm.ERR["CAPACITY"] and "records" in str(e.value)


# ------------------------------------------------------------------ several devices behind the C ABI
def test_device_group_sharded_host(sm, oracle, gpu):
    """mm_run_sharded_host / mm_run_batch_sharded_host (multi-GPU behind the C ABI): a group of two
    workspaces on this GPU - and one entry per GPU when the box has more - gives the oracle's result: window
    ranges with the exact seam, super-k-mer indices, syncmers (no seam rule), contigs placed greedily."""
    import torch
    ngpu = torch.cuda.device_count()
    groups = [[0, 0], [0, 0, 0]] + ([list(range(ngpu))] if ngpu > 1 else [])
    n = 5_000_011
    data = oracle.gen_packed(21, n + 3)
    for devices in groups:
        g = sm.DeviceGroup(devices)
        assert len(g) == len(devices)
        for canon, k, w, mode in ((True, 21, 11, 0), (False, 21, 11, 0), (True, 15, 17, 1), (True, 31, 51, 0)):
            b = sm.Builder(k, w, canon, mode)
            for off in (0, 3):
                want = oracle.run(data, n, k, w, canonical=canon, mode=mode, base_offset=off)
                pos, _ = g.run(b, data, n, base_offset=off)
                assert np.array_equal(pos, want), (devices, canon, k, w, mode, off)
        # a homopolymer run across the seams: the same minimizer position on both sides of a cut
        flat = np.zeros((n + 3) // 4, dtype=np.uint8)
        want = oracle.run(flat, n, 21, 11, canonical=False)
        pos, _ = g.run(sm.minimizers(21, 11), flat, n)
        assert np.array_equal(pos, want)
        # super-k-mer indices
        skv = []
        b = sm.canonical_minimizers(21, 11).super_kmers(skv)
        want_p, want_s = oracle.run(data, n, 21, 11, canonical=True, super_kmers=True)
        pos, sk = g.run(b, data, n)
        assert np.array_equal(pos, want_p) and np.array_equal(sk, want_s)
        # too small a capacity is reported, not overrun
        with pytest.raises(sm.MinimizerError) as e:
            g.run(sm.canonical_minimizers(21, 11), data, n, capacity=1000)
        assert e.value.code == sm.ERR["CAPACITY"]
        # sequences shorter than a window, empty input
        pos, _ = g.run(sm.canonical_minimizers(21, 11), data, 20)
        assert len(pos) == 0
        # contigs: greedy placement, sequence-local positions in input order
        lens = [700_001, 30, 1_200_000, 0, 450_017, 2_000_003, 90_000]
        seqs = [oracle.gen_packed(50 + i, m + 2) for i, m in enumerate(lens)]
        offs_in = [0, 1, 2, 0, 3, 0, 1]
        b = sm.canonical_minimizers(21, 11)
        pos, _, o = g.run_batch(b, seqs, lens, base_offsets=offs_in)
        assert len(o) == len(lens) + 1
        for i, m in enumerate(lens):
            want = oracle.run(seqs[i], m, 21, 11, canonical=True, base_offset=offs_in[i])
            assert np.array_equal(pos[o[i]:o[i + 1]], want), (devices, i)
        g.close()


def test_config1_literal_input_on_the_gpu(sm, gpu):
    """BASELINE.json configs[0] on its literal 1 000-base ASCII input through the ASCII entry point (mm_run_host_ascii)
    and the free function the reference's doctest uses (src/lib.rs:92-99, :639): the committed vector."""
    import json
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "config1.json")))
    seq = g["ascii"].encode()
    assert sm.minimizer_positions(sm.AsciiSeq(seq), 5, 7) == g["positions"]
    assert sm.minimizers(5, 7).run_once(sm.PackedSeqVec.from_ascii(seq)) == g["positions"]
    assert gpu.last_path() == sm.PATH_FUSED
