"""Timing experiment (wrong results by design): MM_DEBUG=16 runs the kernel with lists of half the
capacity (half the LDS per workgroup, overflow ignored) to see what more resident workgroups per CU
would buy for kernels whose registers allow them (forward walks: 69 VGPRs)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")  # this script flips MM_* switches between runs (mm_env.h)
import simd_minimizers_amd as sm
n = 3_100_000_000
d = sm.generate_device(n, 3); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
def t(b, warm=15, reps=15):
    for _ in range(warm): b.run_device(d, n, out, sync=False)
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, sync=False)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l
for (k, w, canon) in [(21, 11, False), (21, 11, True), (15, 5, False), (21, 19, False)]:
    b = sm.Builder(k, w, canon, 0)
    for nblk in (0, 32, 40, 48):
        ws.set_blocks_per_lane(nblk)
        for dbg in (0, 16):
            os.environ["MM_DEBUG"] = str(dbg)
            print(f"k={k} w={w} canon={canon} nblk={nblk or 'default'} debug={dbg}: {t(b):.3f} ms", flush=True)
os.environ["MM_DEBUG"] = "0"
ws.set_blocks_per_lane(0)
