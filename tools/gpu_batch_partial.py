"""Why the 24 CHM13-like contigs cost more in one batch launch than one sequence of the same length: the same contigs
with their lengths cut to whole tiles (no partial last tile), lanes pinned to 29 blocks; canonical k=31 w=51."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
import simd_minimizers_amd as sm
from simd_minimizers_amd import sharding
dev = torch.device("cuda:0")
ws = sm.Workspace(0, torch.cuda.current_stream(dev).cuda_stream)
def gen(n, seed):
    t = torch.zeros((n + 3) // 4 + 64, dtype=torch.uint8, device=dev)
    sm._check(sm.lib().mm_generate_device_async(ws.h, seed, 0, n, t.data_ptr()))
    return t
def timed(step, warm=6, reps=6):
    for _ in range(warm): step()
    torch.cuda.synchronize(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): step()
    torch.cuda.synchronize(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / max(1, l)
k, w, nblk = 31, 51, 29
tile = 256 * nblk * w
b = sm.canonical_minimizers(k, w).workspace(ws)
lens0 = list(sharding.CHM13_CONTIG_LENGTHS)
out = torch.empty(int(sum(lens0) * 2 / (w + 1) * 1.2) + 4096, dtype=torch.int32, device=dev)
import random
random.seed(1)
whole = [((m - (k + w - 2)) // tile) * tile + (k + w - 2) for m in lens0]
cases = [("as they are, tuned lanes", lens0, 0), ("as they are, 29 blocks per lane", lens0, nblk),
         ("cut to whole tiles, 29 blocks per lane", whole, nblk)]
for f in (0.02, 0.1, 0.5, 0.9, 0.98):
    cases.append((f"whole tiles + {f} of a tile each", [m + int(tile * f) for m in whole], nblk))
cases.append(("whole tiles + a random part of a tile each", [m + int(tile * random.random()) for m in whole], nblk))
cases.append(("whole tiles + a random part, lengths made odd", [(m + int(tile * random.random())) | 1 for m in whole], nblk))
for name, lens, pin in cases:
    ws.set_blocks_per_lane(pin)
    d = [gen(m, 100 + i) for i, m in enumerate(lens)]
    res = []
    for dbg in ("0", "1", "2"):  # full / no look-back wait / no copy-out either (wrong results)
        os.environ["MM_DEBUG"] = dbg
        try:
            res.append(f"{timed(lambda: sm.run_batch_device(b, d, lens, out)):.3f}")
        except Exception as e:
            res.append("-")
    os.environ["MM_DEBUG"] = "0"
    print(f"{name}: {sum(lens)} bp, full / no look-back wait / no copy-out: {' / '.join(res)} ms", flush=True)
    del d
ws.set_blocks_per_lane(0)
