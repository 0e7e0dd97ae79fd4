"""world_size-2 gloo test of the sharded (multi-GPU) path on CPU: window-range shards computed
independently, counts all-gathered, positions gathered — the concatenation must equal the
single-process result.  The oracle stands in for the per-rank GPU run here (tests only)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, k, w, canonical, mode, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mm_oracle as o
    from simd_minimizers_amd import sharding

    data = o.gen_packed(17, n)
    l = k + w - 1
    nw = n - l + 1
    win = o.window_positions(data, n, k, w, o.default_hasher(canonical), canonical)

    def compute(b, e):
        # what the kernel does for a window range: windows [b, e) with the seam rule against b-1
        out = []
        for i in range(b, e):
            p = int(win[i])
            if mode == 0:
                if i == 0 or int(win[i - 1]) != p:
                    out.append(p)
            elif mode == 1:
                if p == i or p == i + w - 1:
                    out.append(i)
        return np.array(out, dtype=np.uint32)

    local, counts, gathered = sharding.run_sharded(compute, nw, gather_to=0)
    # the same with shards handed over as tensors (what a device-resident caller does): they are
    # sent as they are and the root receives one tensor
    import torch

    def compute_t(b, e):
        return torch.from_numpy(compute(b, e).view(np.int32))

    local_t, counts_t, gathered_t = sharding.run_sharded(compute_t, nw, gather_to=1)
    assert counts_t == counts and isinstance(local_t, torch.Tensor)
    assert (gathered_t is None) == (rank != 1)
    if rank == 1:
        want = o.run(data, n, k, w, canonical=canonical, mode=mode)
        assert np.array_equal(gathered_t.numpy().view(np.uint32), want)
    if rank == 0:
        want = o.run(data, n, k, w, canonical=canonical, mode=mode)
        q.put((counts, bool(np.array_equal(gathered, want)), len(want)))
    dist.destroy_process_group()


@pytest.mark.parametrize("k,w,canonical,mode", [(21, 11, True, 0), (5, 7, False, 0), (15, 17, True, 1)])
def test_two_rank_window_sharding(k, w, canonical, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + 7 * mode + k
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 20011, k, w, canonical, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    counts, equal, total = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert equal and sum(counts) == total and len(counts) == 2


def test_shard_helpers():
    sys.path.insert(0, ROOT)
    from simd_minimizers_amd import sharding
    assert sharding.shard_windows(10, 4) == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert sharding.shard_windows(0, 2) == [(0, 0), (0, 0)]
    assert sharding.shard_windows(3, 8)[-1] == (3, 3)
    a = sharding.assign_contigs([248, 242, 201, 193, 182, 172, 160, 146, 150, 134, 135, 133], 4)
    assert sorted(i for lst in a for i in lst) == list(range(12))
    loads = [sum([248, 242, 201, 193, 182, 172, 160, 146, 150, 134, 135, 133][i] for i in lst) for lst in a]
    assert max(loads) - min(loads) < 60


def _contig_worker(rank, world, port, lengths, k, w, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mm_oracle as o
    from simd_minimizers_amd import sharding

    def compute_contig(i):
        return o.run(o.gen_packed(100 + i, max(1, lengths[i])), lengths[i], k, w, canonical=True)

    mine, local, counts, gathered = sharding.run_contigs_sharded(compute_contig, lengths, gather_to=0)
    if rank == 0:
        ok = all(np.array_equal(gathered[i], compute_contig(i)) for i in range(len(lengths)))
        q.put((mine, counts, ok))
    dist.destroy_process_group()


def test_two_rank_contig_sharding():
    """Config 4's shape: contigs placed greedily on ranks, contig-local positions, counts exchanged,
    positions gathered to a root in contig order."""
    lengths = [5003, 40, 12001, 0, 777, 9000, 31, 2500]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + 211
    procs = [ctx.Process(target=_contig_worker, args=(r, 2, port, lengths, 31, 51, q)) for r in range(2)]
    for p in procs:
        p.start()
    mine, counts, ok = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ok and len(counts) == len(lengths) and counts[3] == 0 and counts[1] == 0 and counts[0] > 0
    assert 0 < len(mine) < len(lengths)


def _batch_worker(rank, world, port, lengths, k, w, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mm_oracle as o
    from simd_minimizers_amd import sharding

    def one(i):
        return o.run(o.gen_packed(100 + i, max(1, lengths[i])), lengths[i], k, w, canonical=True)

    def compute_batch(idx):
        # what run_batch_device returns on the GPU box: positions back to back (a tensor) + offsets
        parts = [one(i) for i in idx]
        offs = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).tolist()
        flat = np.concatenate(parts + [np.zeros(5, dtype=np.uint32)])  # a buffer larger than its content
        return torch.from_numpy(flat.view(np.int32)), offs

    mine, _, offs, counts, gathered = sharding.run_contig_batch_sharded(compute_batch, lengths, gather_to=0)
    if rank == 0:
        ok = all(np.array_equal(gathered[i].numpy().view(np.uint32), one(i)) for i in range(len(lengths)))
        q.put((mine, counts, ok))
    dist.destroy_process_group()


def test_two_rank_contig_batch_sharding():
    """The config-4 bench path (bench.py --workload contigs): one batch per rank, per-contig counts
    all-reduced, buffers gathered to rank 0 and split per contig."""
    lengths = [5003, 40, 12001, 0, 777, 9000, 31, 2500]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + 331
    procs = [ctx.Process(target=_batch_worker, args=(r, 2, port, lengths, 31, 51, q)) for r in range(2)]
    for p in procs:
        p.start()
    mine, counts, ok = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ok and len(counts) == len(lengths) and counts[3] == 0 and counts[0] > 0


def test_chm13_contig_set():
    sys.path.insert(0, ROOT)
    from simd_minimizers_amd import sharding
    lens = sharding.CHM13_CONTIG_LENGTHS
    assert len(lens) == 24 and 3.0e9 < sum(lens) < 3.2e9 and max(lens) < 2 ** 32
    for world in (1, 2, 4, 8):
        loads = [sum(lens[i] for i in p) for p in sharding.assign_contigs(lens, world)]
        assert sum(loads) == sum(lens) and max(loads) < 1.06 * sum(lens) / world


def test_bench_launcher_fails_loudly():
    """`bench.py --gpus N` must start N ranks itself or fail - never print a
    one-rank line labelled N GPUs.  Without GPUs here the ranks it starts cannot run: the command
    has to exit non-zero without a JSON line; a launcher/flag mismatch is refused before any import."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-extra"], env=env, capture_output=True, text=True, timeout=300)
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and '"n_gpus"' not in r.stdout
    env2 = dict(env, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env2, capture_output=True,
                        text=True, timeout=120)
    assert r2.returncode != 0 and "WORLD_SIZE=2" in r2.stderr and '"n_gpus"' not in r2.stdout


def test_bench_default_is_the_strong_split():
    """the line a driver-run `bench.py --gpus N` prints must be north_star's experiment - ONE
    3.1 Gbp sequence cut N ways, total work fixed - not N independent sequences.  (The N > 1 line itself is checked
    on the GPU box: tests/test_gpu_round4.py::test_bench_world2_line_is_strong.)"""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.resolve_workload(None, 1) == "headline"
    for n_gpus in (2, 4, 8):
        assert bench.resolve_workload(None, n_gpus) == "strong"
        p = bench.strong_plan(bench.N_BASES, n_gpus)
        assert p["scaling"] == "strong" and p["total_bases"] == 3_100_000_000
        assert sum(e - a for a, e in p["ranges"]) == p["windows"] == 3_100_000_000 - 31 + 1
        assert max(e - a for a, e in p["ranges"]) - min(e - a for a, e in p["ranges"]) <= n_gpus
    assert bench.resolve_workload("contigs", 8) == "contigs" and bench.resolve_workload("headline", 8) == "headline"


def _gather_edge_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from simd_minimizers_amd import sharding
    ok = True
    # (counts per rank, root): ranks without a single position, a root that holds nothing, everything on one rank,
    # buffers larger than their counts
    for counts, root in (([5, 0, 3], 0), ([0, 0, 7], 1), ([0, 4, 0], 2), ([0, 0, 0], 0), ([1, 1, 1], 1), ([40000, 3, 0], 2)):
        mine = np.arange(counts[rank] + 4, dtype=np.uint32) + 1000 * rank  # four entries more than the count
        got = sharding.gather_positions_cat(torch.from_numpy(mine.view(np.int32)), counts, root)
        if rank == root:
            cat, parts = got
            want = np.concatenate([np.arange(c, dtype=np.uint32) + 1000 * r for r, c in enumerate(counts)]) if sum(counts) else np.zeros(0, np.uint32)
            ok = ok and np.array_equal(cat.numpy().view(np.uint32), want) and [int(p.numel()) for p in parts] == counts
        else:
            ok = ok and got is None
    if rank == 0:
        q.put(ok)
    else:
        assert ok
    dist.destroy_process_group()


def test_exact_size_gather_edge_cases():
    """sharding.gather_positions_cat (round 4: every rank sends exactly its count point-to-point, the root receives into
    slices of ONE buffer): ranks with nothing to send, a root with nothing of its own, nothing at all, buffers longer than
    their counts - world size 3 over gloo."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + 577
    procs = [ctx.Process(target=_gather_edge_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ok
