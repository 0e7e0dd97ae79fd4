"""A/B timing of the fused kernel under MM_DEBUG switches for the BASELINE configurations (timing
experiments only): 0 full, 1 no look-back wait, 2 no copy-out, 3 both (phase 1 alone)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")  # this script flips MM_* switches between runs (mm_env.h)
import simd_minimizers_amd as sm

n = 3_100_000_000
d = sm.generate_device(n, 3); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
def t(b, warm=12, reps=12):
    for _ in range(warm): b.run_device(d, n, out, sync=False)
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, sync=False)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l
cfgs = [(21, 11, True, 0), (21, 11, False, 0), (31, 51, True, 0), (15, 17, True, 1), (21, 25, False, 0)]
for (k, w, canon, mode) in cfgs:
    b = sm.Builder(k, w, canon, mode)
    res = []
    for dbg in [0, 1, 2, 3, 0]:
        os.environ["MM_DEBUG"] = str(dbg)
        res.append((dbg, t(b)))
    os.environ["MM_DEBUG"] = "0"
    print(f"k={k} w={w} canon={canon} mode={mode}: " + "  ".join(f"dbg{g}={ms:.3f}" for g, ms in res) + f"  -> {n/res[0][1]/1e6:.0f} Gbases/s", flush=True)
