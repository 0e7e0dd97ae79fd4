"""Is tile b always on XCD b mod 8, launch after launch, and how different are the XCDs' speeds?
Traced launches (MM_TRACE) of the bench kernel: per class c = b mod 8 the XCC ids seen and the mean
phase-1 duration."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")  # this script flips MM_* switches between runs (mm_env.h)
import simd_minimizers_amd as sm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_100_000_000
d = sm.generate_device(n, 3)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
for canon in (True, False, True):
    b = sm.Builder(21, 11, canon, 0)
    b.run_device(d, n, out)
    for rep in range(2):
        os.environ["MM_TRACE"] = "/tmp/mm_trace.bin"
        b.run_device(d, n, out)
        del os.environ["MM_TRACE"]
        t = np.fromfile("/tmp/mm_trace.bin", dtype=np.uint64).reshape(-1, 10)
        dur = (t[:, 1] - t[:, 0]).astype(np.float64) / 100.0
        xcc = ((t[:, 4] >> 32) & 15).astype(int)
        cls = np.arange(len(t)) % 8
        pure = [np.bincount(xcc[cls == c], minlength=8).max() / (cls == c).sum() for c in range(8)]
        major = [int(np.bincount(xcc[cls == c], minlength=8).argmax()) for c in range(8)]
        m = [dur[cls == c].mean() for c in range(8)]
        print(f"canon={canon} rep={rep}: class->xcc {major} purity {min(pure):.3f}; phase-1 us by class: " + " ".join(f"{x:.1f}" for x in m) + f"  (max/mean {max(m) / np.mean(m):.3f})", flush=True)
