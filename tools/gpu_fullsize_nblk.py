"""Lane-length sweep at the full 3.1 Gbp size for the configurations with long default lanes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import simd_minimizers_amd as sm
n = 3_100_000_000
d = sm.generate_device(n, 3); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
def t(b, warm=8, reps=8):
    for _ in range(warm): b.run_device(d, n, out, sync=False)
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, sync=False)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l
for k, w, canon, mode, nblks in [(31, 51, True, 0, (0, 12, 16, 20, 24)), (15, 17, True, 1, (0, 16, 20, 24)), (15, 17, True, 2, (0, 28, 36, 44)),
                                 (21, 11, True, 0, (0, 24, 26, 28, 30)), (21, 11, False, 0, (0, 24, 28, 32)), (21, 33, True, 0, (0, 16, 24, 30))]:
    b = sm.Builder(k, w, canon, mode)
    for nblk in nblks:
        ws.set_blocks_per_lane(nblk)
        ms = t(b)
        print(f"k={k} w={w} canon={canon} mode={mode} nblk={nblk or 'default'}: {ms:.3f} ms {n / ms / 1e6:.0f} Gbases/s", flush=True)
ws.set_blocks_per_lane(0)
