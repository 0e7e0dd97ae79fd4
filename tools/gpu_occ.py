"""Phase-1-only (MM_DEBUG=3) and full kernel time vs lane length (occupancy via LDS list size)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")  # this script flips MM_* switches between runs (mm_env.h)
import simd_minimizers_amd as sm

def t(b, ws, d, n, out, reps=5):
    b.run_device(d, n, out, sync=False); ws.sync()
    ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, sync=False)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l

n = 3_100_000_000
d = sm.generate_device(n, 3); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
k, w = 21, 11
b = sm.Builder(k, w, True, 0)
for nblk in [4, 6, 8, 12, 16, 24, 32, 48, 64]:
    ws.set_blocks_per_lane(nblk)
    S = w * nblk
    cap = int(1.3 * 2 / (w + 1) * S) + 8 + w
    lds = cap * 516
    wgs = min(8, (160 * 1024) // (lds + 1024))
    steps_per_window = (S + w + k / 2.0) / S   # rough: block 0 + hash init relative cost
    res = []
    for dbg in (0, 3):
        os.environ["MM_DEBUG"] = str(dbg)
        res.append(t(b, ws, d, n, out))
    os.environ["MM_DEBUG"] = "0"
    print(f"nblk={nblk:3d} S={S:4d} lds={lds/1024:5.1f}KB wg/CU={wgs} full={res[0]:.3f} ms  phase1-only={res[1]:.3f} ms  "
          f"phase1 per step-equivalent={res[1]/steps_per_window:.3f}", flush=True)
