"""Throughput of reads mode: N reads of L bases (one launch), HIP-event kernel time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import simd_minimizers_amd as sm

ws = sm.default_workspace(0)
ws.enable_timing(True)
for k, w, canonical, L, N in [(21, 11, True, 150, 8_000_000), (21, 11, True, 250, 4_000_000),
                              (21, 11, False, 150, 8_000_000), (15, 5, True, 100, 8_000_000),
                              (31, 19, True, 150, 8_000_000), (21, 11, True, 150, 100_000)]:
    stride = L
    d = sm.generate_device(N * stride, seed=1)
    out = torch.empty(N * (L // 2), dtype=torch.int32, device="cuda")
    offs = torch.empty(N + 1, dtype=torch.int64, device="cuda")
    b = sm.Builder(k, w, canonical, 0)
    for it in range(3):
        total = sm.run_reads_device(b, d, N, stride, L, out, offs)
    ws.kernel_time(reset=True)
    for it in range(5):
        total = sm.run_reads_device(b, d, N, stride, L, out, offs)
    ms, n = ws.kernel_time(reset=True)
    print(f"k={k} w={w} canon={canonical} L={L} N={N}: {ms / n:.3f} ms/launch, "
          f"{N * L / (ms / n) / 1e6:.1f} Gbases/s, {total / N:.2f} minimizers/read", flush=True)
    del d, out, offs
