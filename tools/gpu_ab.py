"""A/B timing of the fused kernel under MM_DEBUG switches (timing experiments only)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import simd_minimizers_amd as sm

def t(b, ws, d, n, out, reps=5):
    b.run_device(d, n, out, sync=False); ws.sync()
    ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, sync=False)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
d = sm.generate_device(n, 2); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
for (k, w, canon, nblk) in [(21, 11, True, 0), (21, 11, False, 0)]:
    ws.set_blocks_per_lane(nblk)
    b = sm.Builder(k, w, canon, 0)
    for dbg in [0, 8, 9, 1, 2, 3]:
        os.environ["MM_DEBUG"] = str(dbg)
        ms = t(b, ws, d, n, out)
        print(f"k={k} w={w} canon={canon} nblk={nblk} debug={dbg:2d}: {ms:.3f} ms {n/ms/1e6:.0f} Gbase/s", flush=True)
    os.environ["MM_DEBUG"] = "0"
