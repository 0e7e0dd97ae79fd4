"""Canonical k=21 w=11 at five workgroups per CU (MM_MIN_BLOCKS=5: at most 102 VGPRs; lists short enough for five list
areas in LDS, MM_CAP_LIMIT) against the default four, both through the run-time specialisation, same box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
import simd_minimizers_amd as sm
n = 3_100_000_000
d = sm.generate_device(n, 3); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
os.environ["MM_JIT_FORCE"] = "1"
def t(b, warm=12, reps=12):
    for _ in range(warm): b.run_device(d, n, out, sync=False)
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, sync=False)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l
b = sm.Builder(21, 11, True, 0)
for defs in ("-DMM_X=1", "-DMM_MIN_BLOCKS=5"):
    os.environ["MM_JIT_DEFS"] = defs
    for cap in ("", "60", "56", "52", "44"):
        if cap: os.environ["MM_CAP_LIMIT"] = cap
        else: os.environ.pop("MM_CAP_LIMIT", None)
        os.environ["MM_DEBUG"] = "0"; full = t(b)
        os.environ["MM_DEBUG"] = "3"; walk = t(b)
        os.environ["MM_DEBUG"] = "0"
        print(f"{defs!r:22s} MM_CAP_LIMIT={cap or 'default'}: {full:.3f} ms (walk {walk:.3f})", flush=True)
