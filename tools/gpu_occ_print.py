"""What the runtime computes as resident workgroups per CU for the headline kernels (MM_PRINT_OCC=1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MM_PRINT_OCC"] = "1"
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")  # this script flips MM_* switches between runs (mm_env.h)
import simd_minimizers_amd as sm
n = 400_000_000
d = sm.generate_device(n, 3)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
for k, w, canon, mode in [(21, 11, False, 0), (21, 11, True, 0), (31, 51, True, 0), (15, 17, True, 1), (21, 19, False, 0)]:
    print(f"k={k} w={w} canon={canon} mode={mode}", file=sys.stderr, flush=True)
    sm.Builder(k, w, canon, mode).run_device(d, n, out)
