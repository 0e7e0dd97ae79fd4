"""Multi-GPU sharding of the minimizer path (one process per GPU, torch.distributed).

The path partitions into independent units, so there is no data-path collective:
  * independent sequences (contigs) are assigned to ranks greedily, longest first
    (the reference's only parallel benchmark does `seqs.par_iter()` over contigs,
    bench/src/bin/paper.rs:442-459);
  * ONE long sequence is cut into window ranges [win_begin, win_end); every rank reads its
    range plus a (k + w - 2)-base halo and produces ABSOLUTE positions, and because the kernel
    also evaluates the window just before its range the dedup at the seam is exact
    (src/collect.rs:265-271): concatenating the rank outputs in rank order IS the result.
The only exchange is optional and tiny: an all-gather of the per-rank counts (offsets of each
shard in a notional concatenated buffer) and, if a caller wants everything on one rank, a
gather of the position buffers to that rank (RCCL over xGMI when the backend is nccl: N-1
point-to-point transfers into the root, each on its own link; device-resident shards are sent
from HBM to HBM without touching the host).
"""
from __future__ import annotations

from typing import Callable, Sequence

import numpy as np

# T2T-CHM13v2.0 chromosome lengths (chr1..chr22, X, Y without chrM: 3 117 275 501 bp): the contig set of BASELINE.md
# §4 config 4 ("24 contigs with CHM13-like lengths summing to ~3.1 Gbp"; the assembly itself is not
# available offline, contig i is generator G with seed 100 + i).
CHM13_CONTIG_LENGTHS = (
    248_387_328, 242_696_752, 201_105_948, 193_574_945, 182_045_439, 172_126_628, 160_567_428, 146_259_331,
    150_617_247, 134_758_134, 135_127_769, 133_324_548, 113_566_686, 101_161_492, 99_753_195, 96_330_374,
    84_276_897, 80_542_538, 61_707_364, 66_210_255, 45_090_682, 51_324_926, 154_259_566, 62_460_029,
)
CHM13_CONTIG_SEED0 = 100


def shard_windows(n_windows: int, world: int) -> list[tuple[int, int]]:
    """Equal window ranges, one per rank (the last ranks may be empty for tiny inputs)."""
    per = -(-n_windows // world) if n_windows else 0
    return [(min(r * per, n_windows), min((r + 1) * per, n_windows)) for r in range(world)]


def assign_contigs(lengths: Sequence[int], world: int) -> list[list[int]]:
    """Greedy longest-first placement of contigs on ranks; returns contig indices per rank."""
    loads = [0] * world
    out: list[list[int]] = [[] for _ in range(world)]
    for i in sorted(range(len(lengths)), key=lambda i: -lengths[i]):
        r = min(range(world), key=lambda r: loads[r])
        out[r].append(i)
        loads[r] += lengths[i]
    for lst in out:
        lst.sort()
    return out


def _as_tensor(x, dev):
    """uint32 positions as an int32 tensor on ``dev`` (device tensors stay where they are)."""
    import torch
    if isinstance(x, torch.Tensor):
        return x.to(dev).view(torch.int32).reshape(-1)
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.uint32).view(np.int32)).to(dev)


def gather_positions(local, counts: Sequence[int], gather_to: int, group=None):
    """Per-rank views of gather_positions_cat's buffer (None off the root)."""
    r = gather_positions_cat(local, counts, gather_to, group)
    return None if r is None else r[1]


def gather_positions_cat(local, counts: Sequence[int], gather_to: int, group=None):
    """Variable-size gather of position buffers to ONE rank: every rank sends exactly its `counts[rank]` positions
    to the root only - over xGMI these are N-1 point-to-point transfers into the root on separate links, not an
    all-gather that would move every shard to everybody - and the root receives them straight into their slices of
    ONE buffer of sum(counts) positions (no padding to the largest shard, no per-rank staging buffers).  Returns
    (buffer, list of per-rank int32 views of it) on the root, None elsewhere."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    counts = [int(c) for c in counts]
    t = _as_tensor(local, dev)[: counts[rank]].contiguous()
    peer = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    if rank != gather_to:
        if counts[rank] > 0:
            for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, t, peer(gather_to), group)]):
                req.wait()
        return None
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    cat = torch.empty(max(1, offs[-1]), dtype=torch.int32, device=dev)
    parts = [cat[offs[r]: offs[r + 1]] for r in range(world)]
    ops = [dist.P2POp(dist.irecv, parts[r], peer(r), group) for r in range(world) if r != gather_to and counts[r] > 0]
    parts[gather_to].copy_(t)
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return cat[: offs[-1]], parts


def run_sharded(compute: Callable[[int, int], np.ndarray], n_windows: int, group=None,
                gather_to: int | None = None):
    """Run one window-range shard per rank.

    ``compute(win_begin, win_end)`` produces this rank's positions (on the GPU box it wraps
    ``Builder.run_device(..., win_begin=, win_end=)``).  Returns ``(local, counts, gathered)``:
    this rank's positions, every rank's count (all-gather of one int64), and the concatenated
    result on rank ``gather_to`` (None elsewhere / when not requested).
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    begin, end = shard_windows(n_windows, world)[rank]
    local = compute(begin, end)
    on_device = isinstance(local, torch.Tensor)  # device-resident shards stay on the device
    if not on_device:
        local = np.ascontiguousarray(local, dtype=np.uint32)
    n_local = int(local.numel()) if on_device else len(local)
    if world == 1:
        return local, [n_local], (local if gather_to == 0 else None)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    cnt = torch.tensor([n_local], dtype=torch.int64, device=dev)
    counts_t = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts_t, cnt, group=group)
    counts = [int(c.item()) for c in counts_t]
    gathered = None
    if gather_to is not None:
        got = gather_positions_cat(local, counts, gather_to, group)
        if got is not None:
            cat = got[0]  # rank order == window order: the buffer IS the result
            gathered = cat if on_device else cat.cpu().numpy().view(np.uint32)
    return local, counts, gathered


def run_contigs_sharded(compute_contig: Callable[[int], np.ndarray], lengths: Sequence[int], group=None,
                        gather_to: int | None = None):
    """Independent sequences (contigs) spread over ranks, greedy longest first (SURVEY.md §8d config 4:
    the reference calls ``run`` once per sequence, bench/src/bin/paper.rs:410-431, so positions are
    contig-local).

    ``compute_contig(i)`` produces contig i's positions on this rank (on the GPU box it wraps
    ``run_batch_device`` / ``Builder.run_device``).  Returns ``(mine, local, counts, gathered)``:
    the contig indices of this rank, their positions, the per-contig counts of ALL contigs
    (all-gather of one int64 per contig — the only collective the path needs), and on rank
    ``gather_to`` the list of per-contig position arrays in contig order (None elsewhere).
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    placement = assign_contigs(lengths, world)
    mine = placement[rank]
    local = [np.ascontiguousarray(compute_contig(i), dtype=np.uint32) for i in mine]
    n = len(lengths)
    counts = np.zeros(n, dtype=np.int64)
    for i, p in zip(mine, local):
        counts[i] = len(p)
    if world == 1:
        return mine, local, counts.tolist(), (local if gather_to == 0 else None)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    ct = torch.from_numpy(counts).to(dev)
    dist.all_reduce(ct, op=dist.ReduceOp.SUM, group=group)  # disjoint supports: sum == gather
    counts = ct.cpu().numpy()
    gathered = None
    if gather_to is not None:
        per_rank = [int(sum(counts[i] for i in placement[r])) for r in range(world)]
        cat = np.concatenate(local) if sum(len(p) for p in local) else np.zeros(0, dtype=np.uint32)
        parts = gather_positions(cat, per_rank, gather_to, group)
        if parts is not None:
            gathered = [None] * n
            for r in range(world):
                flat = parts[r].cpu().numpy().view(np.uint32)
                off = 0
                for i in placement[r]:
                    gathered[i] = flat[off: off + int(counts[i])].copy()
                    off += int(counts[i])
    return mine, local, counts.tolist(), gathered


def run_contig_batch_sharded(compute_batch: Callable[[list], tuple], lengths: Sequence[int], group=None,
                             gather_to: int | None = None):
    """Contigs spread over ranks (greedy longest first), ONE batch call per rank.

    ``compute_batch(indices)`` runs this rank's contigs with one plan and returns ``(positions,
    offsets)``: the contig-local positions back to back (a torch tensor - device-resident on the GPU
    box, where it wraps ``run_batch_device``, i.e. ``mm_run_batch_device``: one launch - or a numpy
    array) and the ``len(indices) + 1`` offsets delimiting them.  Returns ``(mine, positions, offsets,
    counts, gathered)``: this rank's contig indices, its batch output, the per-contig counts of ALL
    contigs (one all-reduce of an int64 vector with disjoint supports - the only collective the path
    needs), and on rank ``gather_to`` the per-contig position arrays in contig order (tensors on the
    root's device when the shards are device tensors; None elsewhere / when not requested).  The
    position buffers travel as they are: with the nccl backend HBM -> HBM over xGMI."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    placement = assign_contigs(lengths, world)
    mine = placement[rank]
    positions, offsets = compute_batch(list(mine))
    offsets = [int(o) for o in offsets]
    assert len(offsets) == len(mine) + 1
    on_device = isinstance(positions, torch.Tensor)
    n = len(lengths)
    counts = np.zeros(n, dtype=np.int64)
    for j, i in enumerate(mine):
        counts[i] = offsets[j + 1] - offsets[j]
    if world > 1:
        backend = dist.get_backend(group)
        dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        ct = torch.from_numpy(counts).to(dev)
        dist.all_reduce(ct, op=dist.ReduceOp.SUM, group=group)  # disjoint supports: sum == gather
        counts = ct.cpu().numpy()
    gathered = None
    if gather_to is not None:
        flat = positions[: offsets[-1]] if on_device else np.ascontiguousarray(positions[: offsets[-1]], dtype=np.uint32)
        if world == 1:
            parts = [flat if on_device else torch.from_numpy(flat.view(np.int32))]
        else:
            per_rank = [int(sum(counts[i] for i in placement[r])) for r in range(world)]
            parts = gather_positions(flat, per_rank, gather_to, group)
        if parts is not None:
            gathered = [None] * n
            for r in range(world):
                off = 0
                for i in placement[r]:
                    seg = parts[r][off: off + int(counts[i])]
                    gathered[i] = seg if on_device else seg.cpu().numpy().view(np.uint32).copy()
                    off += int(counts[i])
    return mine, positions, offsets, counts.tolist(), gathered
