"""GPU tests added in round 2 (all through the C ABI, bit-exact against the oracle):

* the scalar-flavour entry points (overwrite semantics), run_with_buf, pos_and_values_*;
* completion status of asynchronous runs (mm_workspace_check) and the LDS layout guard;
* BASELINE config 2 (forward k=21 w=11, 256 Mbp) and config 4's concrete input (24 CHM13-like
  contigs, canonical k=31 w=51, one batch launch);
* the sharded (multi-GPU) call path with the REAL kernel: window ranges and contig batches at
  world size 1, at world size 2 with two ranks sharing this GPU (gloo), and under NCCL when the
  box has two GPUs.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ boundary
def test_scalar_flavour_overwrites(sm, oracle, gpu):
    """run_scalar / run_scalar_once (src/lib.rs:358-376,511-543): the scalar collectors overwrite the
    output vector and truncate it to the result (src/collect.rs:15-37,39-76; src/syncmers.rs:19-48),
    where `run` appends with the last() rule (src/collect.rs:252-272).  Same HIP kernel."""
    ps = sm.PackedSeqVec.from_ascii(b"ACGTGCTCAGAGACTCAGAGGA")
    b = sm.canonical_minimizers(5, 7)
    out = [123, 0, 5, 6, 7, 8, 9, 10]
    o = b.run_scalar(ps, out)
    assert out == [0, 7, 9, 15]
    assert [int(v) for v in o.values_u64()] == [0b1011010001, 0b1100110001, 0b0100110011, 0b1100110001]
    assert b.run_scalar_once(ps) == [0, 7, 9, 15]
    assert gpu.last_path() == sm.PATH_FUSED  # served by the HIP kernel, not by the oracle
    # super-k-mer flavour (src/lib.rs:517-543, src/collect.rs:39-76; known answer src/test.rs:344-356 shape)
    sk = [9, 9, 9, 9, 9, 9, 9]
    out = [1]
    b.super_kmers(sk).run_scalar(ps, out)
    assert out == [0, 7, 9, 15] and sk == [0, 1, 8, 9]
    # a sequence without a window clears min_pos and leaves the super-k-mer vector alone (collect.rs:45-48)
    sk2, out2 = [4, 4], [1, 2, 3]
    b.super_kmers(sk2).run_scalar(ps.slice(0, 10), out2)
    assert out2 == [] and sk2 == [4, 4]
    # syncmers: overwrite as well, no dedup
    n = 5000
    data = oracle.gen_packed(5, n)
    seq = sm.PackedSeq(data, 0, n)
    for mode, builder in ((1, sm.canonical_closed_syncmers(5, 7)), (2, sm.canonical_open_syncmers(5, 7))):
        want = [int(x) for x in oracle.run(data, n, 5, 7, canonical=True, mode=mode)]
        v = [7] * 10000
        builder.run_scalar(seq, v)
        assert v == want
        v2 = [7]
        builder.run(seq, v2)
        assert v2 == [7] + want  # SIMD flavour appends; no boundary rule for syncmers (syncmers.rs:166-169)
    # random sweep: scalar flavour == oracle for short and empty inputs
    rng = np.random.default_rng(5)
    for _ in range(40):
        k, w = int(rng.integers(1, 30)), int(rng.integers(1, 20))
        ln = int(rng.integers(0, 300))
        want = [int(x) for x in oracle.run(data, ln, k, w, canonical=False)]
        v = [1, 2, 3]
        sm.minimizers(k, w).run_scalar(sm.PackedSeq(data, 0, ln), v)
        assert v == want, (k, w, ln)


def test_run_with_buf_and_pos_and_values(sm, oracle, gpu):
    """run_with_buf (src/lib.rs:553-576; Cache == Workspace), run_skip_ambiguous_windows_with_buf
    (src/lib.rs:465-496), Output::pos_and_values_u64 / _u128 (src/lib.rs:598-630)."""
    n = 100_000
    data = oracle.gen_packed(8, n)
    seq = sm.PackedSeq(data, 0, n)
    cache = sm.Workspace(0)
    for k, w, canonical in ((21, 11, True), (33, 9, False), (41, 25, True)):
        b = sm.Builder(k, w, canonical, 0)
        a, c = [], []
        b.run(seq, a)
        out = b.run_with_buf(seq, c, cache)
        assert a == c and len(a) > 0
        assert cache.last_path() == sm.PATH_FUSED
        pos, vals = out.pos_and_values_u128()
        want = oracle.values_u128(data, k, np.array(c, dtype=np.uint32), canonical)
        assert [int(p) for p in pos] == c
        assert vals == [int(lo) | (int(hi) << 64) for lo, hi in want]
        if k <= 32:
            p64, v64 = out.pos_and_values_u64()
            assert [int(p) for p in p64] == c and [int(v) for v in v64] == vals
    ascii_seq = (b"ACGTTGCAGGTTACCAGTNACGGATCCAT" * 120)[:3000]  # an N every 29 bases, windows of 11
    ns = sm.PackedNSeqVec.from_ascii(ascii_seq)
    x, y = [], []
    sm.canonical_minimizers(7, 5).run_skip_ambiguous_windows(ns, x)
    sm.canonical_minimizers(7, 5).run_skip_ambiguous_windows_with_buf(ns, y, cache)
    assert x == y and len(x) > 0
    cache.close()


def test_alternative_hashers(sm, oracle, gpu):
    """MulHasher / AntiLexHasher (src/lib.rs:71-72; src/test.rs:81-83,107-109 run the naive == product
    sweep for them too).  PARITY UNPINNED for the hash arithmetic itself (not in the reference tree, no
    known-answer vector: the tables are this engine's restatement, identical in oracle and product);
    what IS checked bit-exactly is that the HIP path with those tables - constant XOR terms folded into
    the kernels' tables - equals the oracle's definition-level flavour, on both kernel families, plus the
    reverse-complement symmetry of the canonical flavours (src/test.rs:112-152 shape)."""
    rng = np.random.default_rng(2024)
    n = 20_000
    data = oracle.gen_packed(77, n)
    seq = sm.PackedSeq(data, 0, n)
    rc_seq = sm.PackedSeqVec.from_codes((seq.codes()[::-1] ^ 2).astype(np.uint8))
    for k, w in [(5, 7), (16, 11), (21, 11), (31, 19), (8, 4), (33, 5)]:
        for canon in (False, True):
            if canon and (k + w - 1) % 2 == 0:
                continue
            for name, hp, ho in (("mul", sm.MulHasher(k, canon), oracle.mul_hasher(canon)),
                                 ("antilex", sm.AntiLexHasher(k, canon), oracle.antilex_hasher(k, canon))):
                assert list(hp.fw) == list(ho.fw) and list(hp.rc) == list(ho.rc)
                assert (hp.rot, hp.fw_xor, hp.rc_xor, hp.kind) == (ho.rot, ho.fw_xor, ho.rc_xor, ho.kind)
                want = oracle.run(data, n, k, w, hasher=ho, canonical=canon, flavour=oracle.NAIVE)
                b = sm.Builder(k, w, canon, 0).hasher(hp)
                for force_generic in (False, True):
                    gpu.force_generic(force_generic)
                    try:
                        got, _ = b._run_arrays(seq)
                    finally:
                        gpu.force_generic(False)
                    assert np.array_equal(got, want), (name, k, w, canon, force_generic)
                if canon:
                    fwd, _ = b._run_arrays(seq)
                    rev, _ = b._run_arrays(rc_seq)
                    mirrored = np.sort((n - k) - rev.astype(np.int64))
                    assert np.array_equal(np.sort(fwd.astype(np.int64)), mirrored), (name, k, w)
                # syncmers and super-k-mer indices run on the same tables
                for mode in (1, 2):
                    if mode == 2 and w % 2 == 0:
                        continue
                    want_s = oracle.run(data, n, k, w, hasher=ho, canonical=canon, mode=mode)
                    got_s, _ = sm.Builder(k, w, canon, mode).hasher(hp)._run_arrays(seq)
                    assert np.array_equal(got_s, want_s), (name, k, w, canon, mode)


# --------------------------------------------------- asynchronous completion status
_ASYNC_STATUS_SCRIPT = r"""
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "oracle"))
import numpy as np, torch
import mm_oracle as oracle
import simd_minimizers_amd as sm
n, k, w = 2_000_003, 21, 11
data = oracle.gen_packed(31, n)
want = oracle.run(data, n, k, w, canonical=True)
d = torch.from_numpy(data).cuda()
out = torch.zeros(n // 4, dtype=torch.int32, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
ws = sm.Workspace(0, torch.cuda.current_stream().cuda_stream)
b = sm.canonical_minimizers(k, w).workspace(ws)
b.run_device(d, n, out, sync=False, d_count=cnt)
ws.check()  # a clean run: no error
assert int(cnt.item()) == len(want)
os.environ["MM_DEBUG"] = "32"
b.run_device(d, n, out, sync=False, d_count=cnt)
b.run_device(d, n, out, sync=False, d_count=cnt)  # the flag is sticky across later runs
del os.environ["MM_DEBUG"]
b.run_device(d, n, out, sync=False, d_count=cnt)
try:
    ws.check()
    raise SystemExit("MM_ERR_ORDER expected")
except sm.MinimizerError as e:
    assert e.code == sm.ERR["ORDER"], e
ws.check()  # reported once
out.zero_()
b.run_device(d, n, out, sync=False, d_count=cnt)  # now in ticket mode
ws.check()
c = int(cnt.item())
assert c == len(want) and np.array_equal(out[:c].cpu().numpy().view(np.uint32), want)
# the synchronous entry point handles the same report itself: redo in ticket mode, or fail loudly
ws2 = sm.Workspace(0, torch.cuda.current_stream().cuda_stream)
os.environ["MM_DEBUG"] = "32"
try:
    sm.canonical_minimizers(k, w).workspace(ws2).run_device(d, n, out)
    raise SystemExit("a loud failure expected")
except sm.MinimizerError as e:
    assert e.code == sm.ERR["HIP"], e  # the hook fires in ticket mode too: never a silent wrong count
del os.environ["MM_DEBUG"]
c = sm.canonical_minimizers(k, w).workspace(ws2).run_device(d, n, out)
assert c == len(want) and np.array_equal(out[:c].cpu().numpy().view(np.uint32), want)
# ADVICE r2 (low): a time-out that a synchronous entry point repeats itself must not come back from the next check
ws3 = sm.Workspace(0, torch.cuda.current_stream().cuda_stream)
b3 = sm.canonical_minimizers(k, w).workspace(ws3)
os.environ["MM_DEBUG"] = "32"
try:
    b3.run_device(d, n, out)
    raise SystemExit("a loud failure expected")
except sm.MinimizerError:
    pass
del os.environ["MM_DEBUG"]
c = b3.run_device(d, n, out)
assert c == len(want) and np.array_equal(out[:c].cpu().numpy().view(np.uint32), want)
ws3.check()  # nothing asynchronous happened on this workspace: the word the redo raised is gone
print("async status ok")
"""


def test_async_status_is_observable(sm, gpu):
    """a look-back time-out in an asynchronous run must be observable.  MM_DEBUG=32 makes tile 0
    report one; mm_workspace_check returns MM_ERR_ORDER exactly once, switches the workspace to ticket mode, and the
    repeated run is right.  The hook exists only in the EXPERIMENTS build of the library (round 4: the product reads no
    switch that changes results), so the scenario runs in a child process that loads that build."""
    import subprocess
    lib = os.path.join(ROOT, "simd-minimizers_amd", "libsimd_minimizers_amd_exp.so")
    assert os.path.exists(lib), "experiments library not built (make -C simd-minimizers_amd/csrc exp)"
    env = dict(os.environ, MM_LIB_PATH=lib, MM_ENV_DYNAMIC="1")
    r = subprocess.run([sys.executable, "-c", _ASYNC_STATUS_SCRIPT, ROOT], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and "async status ok" in r.stdout, (r.stdout + r.stderr)[-4000:]


def test_product_ignores_experiment_switches(sm, oracle, gpu, monkeypatch):
    """a leaked MM_DEBUG / MM_JIT_DEFS / MM_FASTA_DEBUG must not change what the product computes."""
    import torch
    if os.path.basename(sm.LIB_PATH).endswith("_exp.so"):
        pytest.skip("MM_LIB_PATH names the experiments build, which reads these switches by design")
    n, k, w = 1_000_003, 21, 11
    data = oracle.gen_packed(5, n)
    want = oracle.run(data, n, k, w, canonical=True)
    d = torch.from_numpy(data).cuda()
    out = torch.zeros(n // 3, dtype=torch.int32, device="cuda")
    for name, val in (("MM_DEBUG", "7"), ("MM_DEBUG", "32"), ("MM_JIT_FORCE", "1"), ("MM_JIT_DEFS", "-DMM_STAGE=2"),
                      ("MM_FASTA_DEBUG", "3")):
        monkeypatch.setenv(name, val)
        c = sm.canonical_minimizers(k, w).run_device(d, n, out)
        assert c == len(want) and np.array_equal(out[:c].cpu().numpy().view(np.uint32), want), name
        rec = sm.fasta_pack_device(b">a\nACGTACGTTTGACCA\nACGT\n>b\nTTTTGGGG\n")
        assert len(rec) == 2
        monkeypatch.delenv(name)
    so = open(sm.LIB_PATH, "rb").read()
    for name in (b"MM_DEBUG", b"MM_JIT_DEFS", b"MM_JIT_FORCE", b"MM_FASTA_DEBUG", b"MM_TRACE",
                 # round 5: the cross-check kernels and their switches left the product too
                 b"MM_SPLIT", b"MM_FASTA_KERNEL", b"MM_FASTA_ONEPASS", b"fasta_lines_kernel", b"expand_kernel"):
        assert name not in so, name


def test_list_overflow_with_lds_padding(sm, oracle, gpu, monkeypatch):
    """The list-overflow redo with padded dynamic LDS (MM_LDS_PAD): entries past a list's capacity land in
    the padding instead of being dropped by the hardware; the static tables stay intact (layout guard in
    the kernel) and the redo gives the oracle's output."""
    import torch
    n = 300_000
    for pad in ("0", "4096", "20000"):
        monkeypatch.setenv("MM_LDS_PAD", pad)
        for unit in (b"A", b"ACGTT", b"G"):
            seq = (unit * (n // len(unit) + 1))[:n]
            data = oracle.pack_ascii(seq)
            d = torch.from_numpy(data).cuda()
            out = torch.zeros(n, dtype=torch.int32, device="cuda")
            for k, w, canonical, mode in [(21, 11, False, 0), (21, 11, True, 0), (15, 17, True, 1)]:
                want = oracle.run(data, n, k, w, canonical=canonical, mode=mode)
                c = sm.Builder(k, w, canonical, mode).run_device(d, n, out)
                assert gpu.last_path() == sm.PATH_FUSED
                assert c == len(want) and np.array_equal(out[:c].cpu().numpy().view(np.uint32), want), (pad, unit, k, w)
    monkeypatch.setenv("MM_LDS_PAD", "0")
    torch.zeros(1, device="cuda")
    sm.minimizers(21, 11).run_device(d, n, out)  # resets the process-wide padding for later tests


# ------------------------------------------------------------ BASELINE configs
def _device_checksum(out, c):
    import torch
    v = out[:c].to(torch.int64) & 0xFFFFFFFF
    idx = torch.arange(1, c + 1, dtype=torch.int64, device=out.device)
    return c, int(v.sum().item()), int((v * idx).sum().item()) & ((1 << 64) - 1)


def test_config2_forward_256mbp(sm, oracle, gpu):
    """BASELINE config 2: forward minimizers k=21 w=11 on 256 Mbp (generator G seed 2): the whole output
    against the oracle's multi-threaded one-pass port (itself checked against the streaming oracle in
    tests/test_oracle.py), plus fused == generic and head / tail against the streaming oracle."""
    import torch
    n, k, w = 268_435_456, 21, 11
    d = sm.generate_device(n, 2)
    out = torch.empty(int(n * 0.2), dtype=torch.int32, device="cuda")
    b = sm.minimizers(k, w)
    c = b.run_device(d, n, out)
    assert gpu.last_path() == sm.PATH_FUSED
    assert abs(c / n - 2 / 12) < 2e-3
    host = oracle.gen_packed(2, n)
    assert np.array_equal(d[: (n + 3) // 4].cpu().numpy(), host[: (n + 3) // 4])  # generator kernel == generator G
    want = oracle.run_fast(host, n, k, w, canonical=False, threads=min(16, os.cpu_count() or 1))
    got = out[:c].cpu().numpy().view(np.uint32)
    assert c == len(want) and np.array_equal(got, want)
    m = 1_000_000
    head = oracle.run(host, m, k, w, canonical=False)
    head = head[head < m - 64]
    assert np.array_equal(got[: len(head)], head)
    whole = _device_checksum(out, c)
    gpu.force_generic(True)
    try:
        cg = b.run_device(d, n, out)
        assert gpu.last_path() == sm.PATH_GENERIC and _device_checksum(out, cg) == whole
    finally:
        gpu.force_generic(False)


def test_config4_chm13_contig_batch(sm, oracle, gpu):
    """BASELINE config 4's concrete input on one GPU: the 24 CHM13-like contigs (3.1 Gbp, G seed
    100+contig), canonical k=31 w=51, ONE batch launch (mm_run_batch_device), contig-local positions.
    Per contig: fused == an independent single-sequence run (count + order-sensitive checksum); for
    three contigs the generic family agrees and head and tail equal the oracle."""
    import torch
    from simd_minimizers_amd import sharding
    lens = list(sharding.CHM13_CONTIG_LENGTHS)
    k, w = 31, 51
    d = [sm.generate_device(m, sharding.CHM13_CONTIG_SEED0 + i) for i, m in enumerate(lens)]
    total = sum(lens)
    out = torch.empty(int(total * 2 / 52 * 1.15), dtype=torch.int32, device="cuda")
    b = sm.canonical_minimizers(k, w)
    offs = sm.run_batch_device(b, d, lens, out)
    assert gpu.last_path() == sm.PATH_FUSED
    assert offs[0] == 0 and len(offs) == 25 and abs(offs[-1] / total - 2 / 52) < 1e-3
    single = torch.empty(int(max(lens) * 2 / 52 * 1.2), dtype=torch.int32, device="cuda")
    m = 400_000
    for i, n in enumerate(lens):
        seg = out[offs[i]: offs[i + 1]]
        c = b.run_device(d[i], n, single)
        assert c == offs[i + 1] - offs[i], i
        assert _device_checksum(seg, c) == _device_checksum(single, c), i
        if i in (0, 13, 20):
            gpu.force_generic(True)
            try:
                cg = b.run_device(d[i], n, single)
                assert gpu.last_path() == sm.PATH_GENERIC and _device_checksum(single, cg) == _device_checksum(seg, c), i
            finally:
                gpu.force_generic(False)
            seed = sharding.CHM13_CONTIG_SEED0 + i
            head = oracle.run(oracle.gen_packed(seed, m + 256), m + 256, k, w, canonical=True)
            head = head[head < m - 256]
            assert np.array_equal(seg[: len(head)].cpu().numpy().view(np.uint32), head), i
            tail = oracle.run(oracle.gen_packed(seed, m, first_base=n - m), m, k, w, canonical=True)
            got_tail = seg[c - len(tail) + 50:].cpu().numpy().view(np.uint32).astype(np.int64) - (n - m)
            assert np.array_equal(got_tail, tail[50:].astype(np.int64)), i


# ------------------------------------------------ sharded call path with the real kernel
def _sharded_worker(rank, world, port, backend, q):
    """One rank of the sharded path: sharding.run_sharded -> Builder.run_device(win_begin, win_end) and
    sharding.run_contig_batch_sharded -> run_batch_device, gathered to rank 0 and compared with the oracle."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    dev_id = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev_id)
    if world > 1 or backend == "nccl":
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{dev_id}"))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    import mm_oracle as o
    import simd_minimizers_amd as sm
    from simd_minimizers_amd import sharding

    ws = sm.Workspace(dev_id, torch.cuda.current_stream().cuda_stream)
    ok = True
    n, k, w = 3_000_017, 21, 11
    data = o.gen_packed(21, n)
    d = torch.from_numpy(data).cuda()
    out = torch.zeros(n // 3, dtype=torch.int32, device="cuda")
    for canonical, mode in ((True, 0), (False, 0), (True, 1)):
        b = sm.Builder(k, w, canonical, mode).workspace(ws)

        def compute(wb, we):
            c = b.run_device(d, n, out, win_begin=wb, win_end=we)
            assert ws.last_path() == sm.PATH_FUSED
            return out[:c].clone()

        local, counts, gathered = sharding.run_sharded(compute, n - (k + w - 1) + 1, gather_to=0)
        if backend == "nccl" and world == 1:
            # (one rank: run_sharded returns before its collectives - run them here, through RCCL, on device tensors)
            cnt = torch.tensor([int(local.numel())], dtype=torch.int64, device="cuda")
            counts_t = [torch.zeros(1, dtype=torch.int64, device="cuda")]
            dist.all_gather(counts_t, cnt)
            cat, parts = sharding.gather_positions_cat(local, [int(counts_t[0].item())], 0)
            ok = ok and int(counts_t[0].item()) == counts[0] and torch.equal(cat, local) and len(parts) == 1
            s_ = local[:1024].to(torch.int64).clone()
            dist.all_reduce(s_)
            ok = ok and torch.equal(s_, local[:1024].to(torch.int64))
        if rank == 0:
            want = o.run(data, n, k, w, canonical=canonical, mode=mode)
            got = gathered.cpu().numpy().view(np.uint32)
            ok = ok and sum(counts) == len(want) and np.array_equal(got, want)
    # contigs: one batch launch per rank, contig-local positions, gathered per contig
    lens = [400_003, 250_000, 180_001, 90_000, 60_000, 30, 0, 1_000]
    seqs = [o.gen_packed(100 + i, max(m, 1)) for i, m in enumerate(lens)]
    bb = sm.canonical_minimizers(31, 51).workspace(ws)
    out2 = torch.zeros(sum(lens) // 8 + 64, dtype=torch.int32, device="cuda")

    def compute_batch(idx):
        offs = sm.run_batch_device(bb, [torch.from_numpy(seqs[i]).cuda() for i in idx], [lens[i] for i in idx], out2)
        return out2, offs

    mine, _, _, counts, gathered = sharding.run_contig_batch_sharded(compute_batch, lens, gather_to=0)
    if rank == 0:
        for i, m in enumerate(lens):
            want = o.run(seqs[i], m, 31, 51, canonical=True)
            g = gathered[i].cpu().numpy().view(np.uint32)
            ok = ok and counts[i] == len(want) and np.array_equal(g, want)
        q.put(bool(ok))
    if world > 1 or backend == "nccl":
        dist.barrier()
        dist.destroy_process_group()


def _run_world(world, backend):
    import socket

    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, backend, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        ok = q.get(timeout=600)
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()
    assert ok
    assert all(p.exitcode == 0 for p in procs)


def test_sharded_path_real_kernel_world1(gpu):
    """sharding.run_sharded / run_contig_batch_sharded wired to Builder.run_device / run_batch_device."""
    _run_world(1, "gloo")


def test_sharded_path_real_kernel_world1_rccl(gpu):
    """Round 5: the same helpers with an RCCL process group of ONE rank - what a one-GPU box can execute of the N > 1
    path's communication: the communicator comes up on the device, the count exchange and the gather run through RCCL
    on device tensors (degenerate: one rank), the barrier and the teardown.  Two ranks over RCCL need two GPUs (below)."""
    _run_world(1, "nccl")


def test_sharded_path_real_kernel_world2_shared_gpu(gpu):
    """Two ranks (gloo rendezvous, both on this GPU): every rank computes its own window range / its own
    contigs with the HIP kernel; counts all-gathered, positions gathered to rank 0 == oracle."""
    _run_world(2, "gloo")


def test_sharded_path_real_kernel_world2_nccl(gpu):
    """The same over RCCL with device-resident shards (needs two GPUs)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    _run_world(2, "nccl")
