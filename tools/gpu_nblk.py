"""Steady-state kernel time vs lane length for the bench workload (full kernel and phase 1 only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")  # this script flips MM_* switches between runs (mm_env.h)
import simd_minimizers_amd as sm
n = 3_100_000_000
d = sm.generate_device(n, 3); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
CANON = os.environ.get("MM_NBLK_CANON", "1") == "1"
b = sm.Builder(21, 11, CANON, 0)
def t(warm=15, reps=15):
    for _ in range(warm): b.run_device(d, n, out, sync=False)
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, sync=False)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l
for nblk in [int(x) for x in sys.argv[1:]] or [14, 16, 17, 18, 19, 20, 22, 24]:
    ws.set_blocks_per_lane(nblk)
    S = 11 * nblk; cap = int(1.3 / 6 * S) + 19; lds = cap * 516
    os.environ["MM_DEBUG"] = "0"; full = t()
    os.environ["MM_DEBUG"] = "3"; p1 = t()
    os.environ["MM_DEBUG"] = "0"
    print(f"nblk={nblk} S={S} cap={cap} lds={lds/1024:.1f}KB slots={min(5, int(160*1024/(lds+400)))}: full {full:.3f} ms, phase 1 only {p1:.3f} ms", flush=True)
