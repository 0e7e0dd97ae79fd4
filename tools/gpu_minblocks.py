"""Experiment: the bench kernel compiled for 5 workgroups per CU (MM_MIN_BLOCKS=5: at most 96 VGPRs)
with lanes short enough for 5 list areas in LDS, against the default build (JIT both, same box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")  # this script flips MM_* switches between runs (mm_env.h)
import simd_minimizers_amd as sm
n = 3_100_000_000
d = sm.generate_device(n, 3); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
os.environ["MM_JIT_FORCE"] = "1"
def t(b, warm=20, reps=20):
    for _ in range(warm): b.run_device(d, n, out, sync=False)
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, sync=False)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l
for canon in (True, False):
    b = sm.Builder(21, 11, canon, 0)
    for defs in ("-DMM_X=1", "-DMM_MIN_BLOCKS=5", "-DMM_MIN_BLOCKS=5 -DMM_PF=5", "-DMM_MIN_BLOCKS=6"):
        os.environ["MM_JIT_DEFS"] = defs
        for nblk in (0, 18, 16, 13):
            ws.set_blocks_per_lane(nblk)
            print(f"canonical={canon} {defs!r:36s} nblk={nblk or 'default'}: {t(b):.3f} ms", flush=True)
ws.set_blocks_per_lane(0)
