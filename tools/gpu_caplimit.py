"""Lane-length sweep through MM_CAP_LIMIT (list-length limit that sizes the default lanes) for given (k, w)
canonical plans on 3.1 Gbp: kernel time by HIP events.  usage: gpu_caplimit.py k:w[,k:w...] lim1 lim2 ..."""
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
for kw in sys.argv[1].split(","):
    k, w = map(int, kw.split(":"))
    b = sm.Builder(k, w, os.environ.get("FWD") != "1", 0)
    res = []
    for lim in sys.argv[2:]:
        if lim == "0": os.environ.pop("MM_CAP_LIMIT", None)
        else: os.environ["MM_CAP_LIMIT"] = lim
        res.append((lim, t(b)))
    os.environ.pop("MM_CAP_LIMIT", None)
    print(f"k={k} w={w}: " + "  ".join(f"cap{l}={ms:.3f}" for l, ms in res), flush=True)
