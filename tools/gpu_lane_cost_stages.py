"""Follow-up of gpu_lane_cost_bound.py: which PHASE slows down beyond 34 blocks per lane (canonical k=21 w=11, 3.1 Gbp, timing build
with half-size lists)?  MM_DEBUG 16 = whole kernel, 16+2 = no copy-out, 16+1 = no look-back, 16+3 = the walk alone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
import torch
import simd_minimizers_amd as sm
n = 3_100_000_000
d = sm.generate_device(n, 3); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
b = sm.canonical_minimizers(21, 11)
def t(warm=12, reps=10):
    for _ in range(warm): b.run_device(d, n, out, sync=False)
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, sync=False)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l
os.environ["MM_JIT_FORCE"] = "1"
os.environ["MM_JIT_DEFS"] = sys.argv[1] if len(sys.argv) > 1 else ""
for dbg, label in ((16, "whole kernel"), (18, "no copy-out"), (17, "no look-back"), (19, "walk alone")):
    os.environ["MM_DEBUG"] = str(dbg)
    row = []
    for nb in (28, 34, 37, 40, 48, 56):
        ws.set_blocks_per_lane(nb)
        row.append(f"{nb}: {t():.4f}")
    ws.set_blocks_per_lane(0)
    print(f"MM_DEBUG={dbg:2d} {label:14s} | " + " | ".join(row), flush=True)
