"""Experiment: launch bounds for reads-mode kernels of larger w (JIT, even w so that no prebuilt
instance is taken)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")  # this script flips MM_* switches between runs (mm_env.h)
import simd_minimizers_amd as sm
ws = sm.default_workspace(0)
ws.enable_timing(True)
L, N = 150, 8_000_000
d = sm.generate_device(N * L, seed=1)
out = torch.empty(N * (L // 2), dtype=torch.int32, device="cuda")
offs = torch.empty(N + 1, dtype=torch.int64, device="cuda")
for k, w in ((20, 20), (20, 24), (20, 30), (20, 36), (20, 44)):
    b = sm.Builder(k, w, True, 0)
    for defs in ("-DMM_X=1", "-DMM_MIN_BLOCKS=4", "-DMM_MIN_BLOCKS=3"):
        os.environ["MM_JIT_DEFS"] = defs
        for it in range(3):
            sm.run_reads_device(b, d, N, L, L, out, offs)
        ws.kernel_time(reset=True)
        for it in range(5):
            sm.run_reads_device(b, d, N, L, L, out, offs)
        ms, n = ws.kernel_time(reset=True)
        print(f"reads k={k} w={w} {defs!r:22s}: {ms / n:.3f} ms", flush=True)
