"""BASELINE config 4 (canonical k=31 w=51, the 24 CHM13-like contigs, one batch launch) and the single 3.1 Gbp sequence
over blocks per lane (0 = the launcher's choice): gpu_c4_nblk.py n1 n2 ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import simd_minimizers_amd as sm
from simd_minimizers_amd import sharding
ws = sm.default_workspace(0)
lens = list(sharding.CHM13_CONTIG_LENGTHS)
d = [sm.generate_device(m, sharding.CHM13_CONTIG_SEED0 + i) for i, m in enumerate(lens)]
n = sum(lens)
out = torch.zeros(int(n * 2 / 52 * 1.2) + 4096, dtype=torch.int32, device="cuda")
b = sm.canonical_minimizers(31, 51)
one = sm.generate_device(3_100_000_000, 3)
def t(step, warm=8, reps=8):
    for _ in range(warm): step()
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): step()
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l
for nb in map(int, sys.argv[1:] or ["0"]):
    ws.set_blocks_per_lane(nb)
    tb = t(lambda: sm.run_batch_device(b, d, lens, out))
    t1 = t(lambda: b.run_device(one, 3_100_000_000, out, sync=False))
    print(f"blocks per lane {nb:3d}: batch of 24 contigs {tb:.3f} ms ({n / tb / 1e6:.0f} Gbases/s) | one 3.1 Gbp sequence {t1:.3f} ms ({3100 / t1:.0f} Gbases/s)", flush=True)
ws.set_blocks_per_lane(0)
