"""Blocks-per-lane sweep for one canonical (k, w) on 3.1 Gbp: gpu_nblk2.py k w n1 n2 ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import simd_minimizers_amd as sm
k, w = int(sys.argv[1]), int(sys.argv[2])
n = 3_100_000_000
d = sm.generate_device(n, 3); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
def t(b, warm=10, reps=12):
    for _ in range(warm): b.run_device(d, n, out, sync=False)
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, sync=False)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l
b = sm.canonical_minimizers(k, w)
res = []
for nb in map(int, sys.argv[3:]):
    ws.set_blocks_per_lane(nb)
    res.append((nb, t(b)))
ws.set_blocks_per_lane(0)
print(f"k={k} w={w}: " + "  ".join(f"nblk{nb}={ms:.3f}" for nb, ms in res), flush=True)
