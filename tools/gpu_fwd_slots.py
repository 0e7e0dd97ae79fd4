"""Forward k=21 w=11 on 3.1 Gbp: resident workgroups per CU (LDS padding) x list-length limit, full kernel
and phase 1 alone.  The forward walk needs 36 VGPRs and 16 KB of LDS, so eight workgroups per CU are resident by
default; is that the best number now that the status words no longer contend?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")  # this script flips MM_* switches between runs (mm_env.h)
import simd_minimizers_amd as sm
n = 3_100_000_000
d = sm.generate_device(n, 3); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
CANON = os.environ.get("CANON") == "1"
b = sm.Builder(21, 11, CANON, 0)
def t(warm=12, reps=12):
    for _ in range(warm): b.run_device(d, n, out, sync=False)
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, sync=False)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l
t()
for cap in ([0, 50, 40] if not CANON else [0]):
    if cap: os.environ["MM_CAP_LIMIT"] = str(cap)
    for pad in [0, 7000, 11000, 16500, 24500, 38000]:
        os.environ["MM_LDS_PAD"] = str(pad)
        os.environ["MM_DEBUG"] = "0"; full = t()
        os.environ["MM_DEBUG"] = "3"; p1 = t()
        os.environ["MM_DEBUG"] = "0"
        print(f"cap_limit={cap or 'default'} pad={pad}: full {full:.3f} ms, phase 1 only {p1:.3f} ms", flush=True)
