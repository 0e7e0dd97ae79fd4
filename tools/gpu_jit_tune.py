"""Tuning experiments through the run-time specialisation: the bench kernel compiled with different
-D options (MM_JIT_FORCE / MM_JIT_DEFS), HIP-event kernel time on 3.1 Gbp at steady clocks
(20 untimed launches first), configurations interleaved over two rounds."""
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
def t(b, warm=20, reps=30):
    for _ in range(warm): b.run_device(d, n, out, sync=False)
    ws.sync()
    ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, sync=False)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l
res = {}
for rnd in range(2):
    for defs in sys.argv[1:]:
        os.environ["MM_JIT_DEFS"] = defs
        for canon in (True, False):
            b = sm.Builder(21, 11, canon, 0)
            res.setdefault((defs, canon), []).append(t(b))
for (defs, canon), v in res.items():
    print(f"{defs!r:46s} canonical={canon}: " + " ".join(f"{x:.3f}" for x in v) + f" ms  -> {n / min(v) / 1e6:.0f} Gbases/s", flush=True)
