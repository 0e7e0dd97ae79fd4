"""A/B timing of the fused kernel under MM_DEBUG switches at steady clocks (timing experiments only).
1: fake look-back (no wait), 2: no copy-out, 3: both, 4: no phase 1, 8: copy-out without stores."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")  # this script flips MM_* switches between runs (mm_env.h)
import simd_minimizers_amd as sm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_100_000_000
d = sm.generate_device(n, 3); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
def t(b, warm=15, reps=15):
    for _ in range(warm): b.run_device(d, n, out, sync=False)
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, sync=False)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l
for (k, w, canon) in [(21, 11, True), (21, 11, False)]:
    b = sm.Builder(k, w, canon, 0)
    for dbg in [0, 8, 1, 2, 9, 3, 4]:
        os.environ["MM_DEBUG"] = str(dbg)
        ms = t(b)
        print(f"k={k} w={w} canon={canon} debug={dbg:2d}: {ms:.3f} ms {n/ms/1e6:.0f} Gbase/s", flush=True)
    os.environ["MM_DEBUG"] = "0"
