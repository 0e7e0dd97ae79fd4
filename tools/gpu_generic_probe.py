"""Speed of the generic family (window sizes without a fused instance) vs the fused kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import simd_minimizers_amd as sm

n = 1_000_000_000
d = sm.generate_device(n, 2)
out = torch.empty(int(n * 0.45), dtype=torch.int32, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
ws = sm.default_workspace(0)

def timed(fn, reps=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for k, w, canonical in [(21, 11, True), (12, 18, True), (21, 20, False), (31, 35, True), (16, 100, True)]:
    b = sm.Builder(k, w, canonical, 0)
    for fg in (False, True):
        ws.force_generic(fg)
        t = timed(lambda: b.run_device(d, n, out, sync=False, d_count=cnt))
        print(f"k={k} w={w} canonical={canonical} force_generic={fg} path={ws.last_path()}: {t:.3f} ms "
              f"({n / t / 1e6:.0f} Gbases/s)", flush=True)
    ws.force_generic(False)
