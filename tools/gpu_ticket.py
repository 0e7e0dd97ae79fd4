"""Dispatch-order tile ids (default) against tile ids from the atomic ticket (MM_FORCE_TICKET=1) on 3.1 Gbp."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")  # this script flips MM_* switches between runs (mm_env.h)
import simd_minimizers_amd as sm
n = 3_100_000_000
d = sm.generate_device(n, 3); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
def t(b, warm=10, reps=12):
    for _ in range(warm): b.run_device(d, n, out, sync=False)
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, sync=False)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l
for (k, w, canon, mode) in [(21, 11, True, 0), (21, 11, False, 0), (31, 51, True, 0)]:
    b = sm.Builder(k, w, canon, mode)
    res = []
    for tk in ("0", "1", "0", "1"):
        if tk == "1": os.environ["MM_FORCE_TICKET"] = "1"
        else: os.environ.pop("MM_FORCE_TICKET", None)
        res.append(t(b))
    os.environ.pop("MM_FORCE_TICKET", None)
    print(f"k={k} w={w} canon={canon}: blockIdx {res[0]:.3f} {res[2]:.3f}  ticket {res[1]:.3f} {res[3]:.3f}", flush=True)
