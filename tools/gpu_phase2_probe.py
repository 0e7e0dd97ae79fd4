"""Fixed per-tile costs of the fused kernel (timing experiment): MM_DEBUG=4 skips phase 1 (every list
stays empty, so nothing is copied either): what remains is workgroup launch + table setup + look-back.
MM_DEBUG=5 also skips the look-back wait.  Swept over the lane length (tile count)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")  # this script flips MM_* switches between runs (mm_env.h)
import simd_minimizers_amd as sm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_100_000_000
d = sm.generate_device(n, 3); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
def t(b, warm=10, reps=10):
    for _ in range(warm): b.run_device(d, n, out, sync=False)
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, sync=False)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l
b = sm.Builder(21, 11, False, 0)
for nblk in (6, 12, 24, 48, 96):
    ws.set_blocks_per_lane(nblk)
    tiles = (n + 256 * 11 * nblk - 1) // (256 * 11 * nblk)
    for dbg in (4, 5, 7, 0):
        os.environ["MM_DEBUG"] = str(dbg)
        ms = t(b)
        print(f"forward w=11 nblk={nblk} tiles={tiles} debug={dbg}: {ms:.3f} ms = {ms * 1e6 / tiles:.1f} ns per tile", flush=True)
os.environ["MM_DEBUG"] = "0"
ws.set_blocks_per_lane(0)
