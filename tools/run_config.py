#!/usr/bin/env python3
"""Runs ONE secondary configuration a few times (the command rocprofv3 wraps for the per-configuration
profiles under profiles/): tools/run_config.py <C2|FWD|C4|C5|SK|READS|C3|SHARD|READS_SK|SKIP|VALUES|PACK|FASTA|FASTQ> [steps] [warmup].
Prints kernel time by HIP events (median) as one JSON line.  Measurement aid, not part of the product."""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import simd_minimizers_amd as sm  # noqa: E402
from simd_minimizers_amd import sharding  # noqa: E402

cfg = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
ws = sm.Workspace(0, torch.cuda.current_stream(dev).cuda_stream)
if os.environ.get("MM_RUN_NBLK"):  # (A/B runs: blocks per lane pinned, uniform tiles)
    ws.set_blocks_per_lane(int(os.environ["MM_RUN_NBLK"]))


def gen(n, seed):
    t = torch.zeros((n + 3) // 4 + 64, dtype=torch.uint8, device=dev)
    sm._check(sm.lib().mm_generate_device_async(ws.h, seed, 0, n, t.data_ptr()))
    return t


from simd_minimizers_amd import workloads  # noqa: E402

if cfg in workloads.COMPONENTS or cfg.startswith(("SKIP_W", "PLAIN_W")):  # the rows either side of the path
    print(json.dumps(workloads.measure(cfg, ws, dev, warm=warm, reps=steps)))
    sys.exit(0)

N = 3_100_000_000
if cfg in ("C2", "FWD", "C3", "C5", "SK"):
    b, n, seed, dens = {"C2": (sm.minimizers(21, 11), 268_435_456, 2, 2 / 12),
                        "FWD": (sm.minimizers(21, 11), N, 3, 2 / 12),
                        "C3": (sm.canonical_minimizers(21, 11), N, 3, 2 / 12),
                        "C5": (sm.canonical_closed_syncmers(15, 17), N, 3, 2 / 17),
                        # canonical minimizers + super-k-mer indices at full size (bench.py's SK row)
                        "SK": (sm.canonical_minimizers(21, 11), N, 3, 2 / 12)}[cfg]
    b = b.workspace(ws)
    d = gen(n, seed)
    out = torch.empty(int(n * dens * 1.15) + 4096, dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    sk = torch.empty_like(out) if cfg == "SK" else None

    def step():
        b.run_device(d, n, out, sync=False, d_count=cnt, out_sk=sk)

    def n_out():
        return int(cnt.item())
elif cfg == "SHARD":
    # the strong split at N = 8 (bench.py's `extra` row) on ONE GPU: the eight window ranges of the 3.1 Gbp sequence,
    # one launch each, back to back (an isolated 0.2 ms launch after an idle gap runs 15 % slower: clock ramp)
    b = sm.canonical_minimizers(21, 11).workspace(ws)
    d = gen(N, 3)
    nw = N - 31 + 1
    per = -(-nw // 8)
    ranges = [(r * per, min((r + 1) * per, nw)) for r in range(8)]
    n = nw // 8  # per launch
    out = torch.empty(int(per * 2 / 12 * 1.15) + 4096, dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)

    def step():
        for wb, we in ranges:
            b.run_device(d, N, out, win_begin=wb, win_end=we, sync=False, d_count=cnt)

    def n_out():
        return int(cnt.item())
elif cfg == "C4":
    lens = list(sharding.CHM13_CONTIG_LENGTHS)
    b = sm.canonical_minimizers(31, 51).workspace(ws)
    d = [gen(m, sharding.CHM13_CONTIG_SEED0 + i) for i, m in enumerate(lens)]
    n = sum(lens)
    out = torch.empty(int(n * 2 / 52 * 1.15) + 4096, dtype=torch.int32, device=dev)
    offs = [0]

    def step():
        offs[:] = sm.run_batch_device(b, d, lens, out)

    def n_out():
        return int(offs[-1])
elif cfg == "READS":
    n_reads, rl = 8_000_000, 150
    n = n_reads * rl
    b = sm.canonical_minimizers(21, 11).workspace(ws)
    d = gen(n, 7)
    out = torch.empty(int(n * 0.2), dtype=torch.int32, device=dev)
    offs = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)

    def step():
        sm.run_reads_device(b, d, n_reads, rl, rl, out, offs, d_count=cnt, sync=False)

    def n_out():
        return int(cnt.item())
else:
    sys.exit("unknown config " + cfg)

for _ in range(warm):
    step()
torch.cuda.synchronize()
ws.enable_timing(True)
ws.kernel_time(True)
ms = []
for _ in range(steps):
    step()
    torch.cuda.synchronize()
    t, k = ws.kernel_time(True)
    ms.append(t / max(1, k))
ws.check()
med = statistics.median(ms)
no = n_out()
alg = (n + 3) // 4 + 4 * no * (2 if cfg == "SK" else 1) + (8 * 8_000_000 if cfg == "READS" else 0)
print(json.dumps({"config": cfg, "bases": n, "outputs": no, "kernel_ms_median": round(med, 4),
                  "kernel_ms_all": [round(x, 4) for x in ms], "Gbases_per_s": round(n / med / 1e6, 1),
                  "algorithmic_bytes": alg, "frac_of_8TBps": round(alg / (med * 1e-3) / 8e12, 4)}))
