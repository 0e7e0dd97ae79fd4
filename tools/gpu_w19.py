"""Experiment: canonical w = 19 / 21 kernels bounded to 128 VGPRs (4 waves per SIMD, small spills)
against the unbounded builds (134 / 138 VGPRs: 3 waves per SIMD); JIT both, same box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")  # this script flips MM_* switches between runs (mm_env.h)
import simd_minimizers_amd as sm
n = 1_000_000_000
d = sm.generate_device(n, 3); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
os.environ["MM_JIT_FORCE"] = "1"
CANON = os.environ.get("MM_CANON", "1") == "1"
MODE = int(os.environ.get("MM_MODE", "0"))
SK = os.environ.get("MM_SK", "0") == "1"
sk = torch.zeros_like(out) if SK else None
def t(b, warm=20, reps=20):
    for _ in range(warm): b.run_device(d, n, out, out_sk=sk, sync=False)
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, out_sk=sk, sync=False)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l
for k, w in [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or ((19, 19), (21, 21), (15, 23), (31, 25)):
    b = sm.Builder(k, w, CANON, MODE)
    for defs in os.environ.get("MM_DEFS_LIST", "-DMM_X=1;-DMM_MIN_BLOCKS=4;-DMM_MIN_BLOCKS=3;-DMM_MIN_BLOCKS=2").split(";"):
        os.environ["MM_JIT_DEFS"] = defs
        print(f"k={k} w={w} {defs!r:24s}: {t(b):.3f} ms per Gbp", flush=True)
