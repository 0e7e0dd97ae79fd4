"""Same kernel, prebuilt instance vs run-time specialised module: where does the difference come from?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")  # this script flips MM_* switches between runs (mm_env.h)
import simd_minimizers_amd as sm
n = 3_100_000_000
d = sm.generate_device(n, 3); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
b = sm.Builder(21, 11, True, 0)
def t(reps=6):
    b.run_device(d, n, out, sync=False); ws.sync()
    ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, sync=False)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l
for rnd in range(2):
    for force in (False, True):
        if force: os.environ["MM_JIT_FORCE"] = "1"
        else: os.environ.pop("MM_JIT_FORCE", None)
        for dbg in (0, 3, 7):
            os.environ["MM_DEBUG"] = str(dbg)
            print(f"jit={force} debug={dbg}: {t():.3f} ms", flush=True)
os.environ["MM_DEBUG"] = "0"
