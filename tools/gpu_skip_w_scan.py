"""Skip-ambiguous dirty walk (an isolated N every 8 kbp) against the plain walk across window sizes, prebuilt and specialised at
run time, product library: tools/gpu_skip_w_scan.py [w ...]   (kernel ms per Gbp, canonical k=31 / 30)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import simd_minimizers_amd as sm
ws_ = [int(x) for x in sys.argv[1:]] or [13, 16, 18, 20, 21, 24, 25, 28, 31, 32, 33, 34, 35, 36, 37, 38, 40, 41, 45, 51, 60, 64]
n = int(os.environ.get("MM_N", "1000000000"))
d = sm.generate_device(n, 2); ws = sm.default_workspace(0)
amb = torch.zeros((n + 7) // 8 + 64, dtype=torch.uint8, device="cuda")
g = torch.Generator(device="cuda"); g.manual_seed(1)
amb[torch.randint(0, n // 8, (n // 8000,), device="cuda", generator=g)] = 1 << 3
out = torch.zeros(int(n * 0.2), dtype=torch.int32, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
pre = set(sm.prebuilt_window_sizes(True))
def kt(step, reps=5):
    import time
    t0 = time.perf_counter()  # warm up by TIME (a run-time specialisation leaves the chip idle for seconds; 80 ms of steps were not enough, the walk read 20 % slow)
    while time.perf_counter() - t0 < 0.5:
        step(); ws.sync()
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): step()
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / max(1, l)
for w in ws_:
    k = 31 if w % 2 else 30  # (canonical windows need odd k + w - 1)
    b = sm.canonical_minimizers(k, w)
    plain = kt(lambda: b.run_device(d, n, out, sync=False, d_count=cnt))
    dirty = kt(lambda: b.run_skip_ambiguous_device(d, amb, n, out, sync=False, d_count=cnt))
    print(f"n={n} k={k} w={w:3d} {'prebuilt' if w in pre else 'run-time':8s}: plain {plain:.3f} ms | dirty {dirty:.3f} ms | x {dirty / plain:.2f}", flush=True)
