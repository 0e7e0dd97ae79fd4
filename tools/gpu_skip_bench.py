"""Throughput of the skip-ambiguous path (window-ambiguity prepass + fused kernel) vs the plain
path on the same device-resident sequence; torch events on the workspace's stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import simd_minimizers_amd as sm

n = 1_000_000_000
d = sm.generate_device(n, 2)
amb = torch.zeros((n + 7) // 8 + 64, dtype=torch.uint8, device="cuda")
g = torch.Generator(device="cuda"); g.manual_seed(1)
# 0.1% isolated Ns + 200 gaps of 50 kbp
idx = torch.randint(0, n // 8, (n // 8000,), device="cuda", generator=g)
amb[idx] = 1 << 3
for s in torch.randint(0, n // 8 - 7000, (200,), generator=g, device="cuda").tolist():
    amb[s:s + 6250] = 0xFF
out = torch.empty(int(n * 0.4), dtype=torch.int32, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")

def timed(fn, reps=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for k, w, mode in [(21, 11, 0), (31, 51, 0), (15, 17, 1)]:
    b = sm.Builder(k, w, True, mode)
    t_plain = timed(lambda: b.run_device(d, n, out, sync=False, d_count=cnt))
    c_plain = int(cnt.item())
    t_skip = timed(lambda: b.run_skip_ambiguous_device(d, amb, n, out, sync=False, d_count=cnt))
    c_skip = int(cnt.item())
    print(f"k={k} w={w} mode={mode}: plain {t_plain:.3f} ms ({n / t_plain / 1e6:.0f} Gbases/s, {c_plain} out)   "
          f"skip-ambiguous {t_skip:.3f} ms ({n / t_skip / 1e6:.0f} Gbases/s, {c_skip} out)", flush=True)
