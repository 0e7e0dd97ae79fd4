"""Throughput of reads mode: N reads of L bases (one launch), HIP-event kernel time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import simd_minimizers_amd as sm

ws = sm.default_workspace(0)
ws.enable_timing(True)
for k, w, canonical, L, N, mode, use_sk in [(21, 11, True, 150, 8_000_000, 0, False), (21, 11, True, 250, 4_000_000, 0, False),
                              (21, 11, False, 150, 8_000_000, 0, False), (15, 5, True, 100, 8_000_000, 0, False),
                              (31, 19, True, 150, 8_000_000, 0, False), (21, 11, True, 150, 100_000, 0, False),
                              (21, 11, True, 150, 8_000_000, 0, True), (15, 17, True, 150, 8_000_000, 1, False),
                              (15, 17, True, 150, 8_000_000, 2, False)]:
    stride = L
    d = sm.generate_device(N * stride, seed=1)
    out = torch.empty(N * (L // 2), dtype=torch.int32, device="cuda")
    sk = torch.empty_like(out) if use_sk else None
    offs = torch.empty(N + 1, dtype=torch.int64, device="cuda")
    b = sm.Builder(k, w, canonical, mode)
    for it in range(3):
        total = sm.run_reads_device(b, d, N, stride, L, out, offs, out_sk=sk)
    ws.kernel_time(reset=True)
    for it in range(5):
        total = sm.run_reads_device(b, d, N, stride, L, out, offs, out_sk=sk)
    ms, n = ws.kernel_time(reset=True)
    print(f"k={k} w={w} canon={canonical} mode={mode} sk={use_sk} L={L} N={N}: {ms / n:.3f} ms/launch, "
          f"{N * L / (ms / n) / 1e6:.1f} Gbases/s, {total / N:.2f} minimizers/read", flush=True)
    del d, out, offs
