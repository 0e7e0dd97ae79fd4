"""READS_SK on fresh leases: the driver read 0.88-0.90 ms for the reads row with super-k-mer indices three
rounds running where the builder's boxes read 0.73-0.76.  Hypothesis: the kernel's TWO output streams (positions and indices,
the same offsets at the same time) alias in the memory channels when the two arrays' addresses are congruent modulo some large
power of two - which depends on where a fresh process's allocator puts them.  This script places the index array at CONTROLLED
distances from the position array inside one allocation and times the kernel (HIP events, median of 7) for each, then the
permutations the verdict lists: READS_SK before READS, outputs pre-touched, two separate allocations."""
import os, sys, statistics, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import simd_minimizers_amd as sm
ws = sm.default_workspace(0); L = sm.lib(); dev = "cuda:0"
n_reads, rl = 8_000_000, 150
n = n_reads * rl
b = sm.canonical_minimizers(21, 11).workspace(ws)
d = sm.generate_device(n, 7)
offs = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
cap = int(n * 0.2)


def time_step(step, reps=7, warm_ms=60):
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < warm_ms / 1e3:
        for _ in range(3): step()
        torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); step(); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return statistics.median(ms), min(ms), max(ms)


def run(out, sk):
    return lambda: sm.run_reads_device(b, d, n_reads, rl, rl, out, offs, d_count=cnt, sync=False, out_sk=sk)


print(f"device {torch.cuda.get_device_name(0)}; reads row: {n_reads} x {rl} bp canonical k=21 w=11; capacity {cap} positions", flush=True)
# ---- 1. fresh two allocations, as bench.py / workloads.py make them (torch.empty + empty_like)
out = torch.empty(cap, dtype=torch.int32, device=dev); sk = torch.empty_like(out)
print(f"two torch allocations: pos at {out.data_ptr():#x}, sk at {sk.data_ptr():#x}, distance {sk.data_ptr() - out.data_ptr():#x}", flush=True)
m = time_step(run(out, sk)); print(f"  READS_SK first thing in the process: {m[0]:.4f} ms (min {m[1]:.4f}, max {m[2]:.4f})", flush=True)
m = time_step(run(out, None)); print(f"  READS (no indices), same buffers:    {m[0]:.4f} ms", flush=True)
m = time_step(run(out, sk)); print(f"  READS_SK again:                      {m[0]:.4f} ms", flush=True)
out.fill_(0); sk.fill_(0)
m = time_step(run(out, sk)); print(f"  READS_SK, outputs pre-touched (fill_): {m[0]:.4f} ms", flush=True)
del out, sk; torch.cuda.empty_cache()
# ---- 2. one allocation, the index array at a controlled distance behind the position array
base_bytes = (cap * 4 + (1 << 21) - 1) >> 21 << 21   # the position array's extent, in whole 2 MiB pages
big = torch.empty(base_bytes * 2 + (1 << 30) + (1 << 22), dtype=torch.uint8, device=dev)
a0 = (big.data_ptr() + (1 << 21) - 1) >> 21 << 21
print(f"one allocation at {big.data_ptr():#x}; position array at {a0:#x}", flush=True)


def view(addr):  # an int32 view of `cap` elements at device address addr inside `big`
    o = addr - big.data_ptr()
    return big[o: o + cap * 4].view(torch.int32)


pos = view(a0)
for name, dist in (("extent rounded to 2 MiB", base_bytes), ("+ 128 B", base_bytes + 128), ("+ 4 KiB", base_bytes + 4096),
                   ("+ 64 KiB", base_bytes + (64 << 10)), ("+ 1 MiB", base_bytes + (1 << 20)),
                   ("next multiple of 16 MiB", (base_bytes + (1 << 24) - 1) >> 24 << 24),
                   ("next multiple of 256 MiB", (base_bytes + (1 << 28) - 1) >> 28 << 28),
                   ("exactly 1 GiB", 1 << 30), ("1 GiB + 4 KiB", (1 << 30) + 4096), ("1 GiB + 2 MiB", (1 << 30) + (1 << 21))):
    skv = view(a0 + dist)
    m = time_step(run(pos, skv))
    print(f"  sk at distance {dist:#12x} ({name:>26}): {m[0]:.4f} ms (min {m[1]:.4f}, max {m[2]:.4f})", flush=True)
m = time_step(run(pos, None)); print(f"  READS (no indices) in the same buffer: {m[0]:.4f} ms", flush=True)
