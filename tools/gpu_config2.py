"""BASELINE.json config 2: forward minimizers k=21 w=11 on a 256 Mbp PackedSeq (kernel time by HIP
events and whole device-resident call), over a few lane lengths."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import simd_minimizers_amd as sm
n = 256 * 1024 * 1024
d = sm.generate_device(n, 2); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
d_count = torch.zeros(1, dtype=torch.int64, device="cuda")
for canon in (False, True):
    b = sm.Builder(21, 11, canon, 0)
    for nblk in (0, 8, 10, 12, 14, 18):
        ws.set_blocks_per_lane(nblk)
        for _ in range(30): b.run_device(d, n, out, sync=False, d_count=d_count)
        ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
        t0 = time.perf_counter()
        for _ in range(50): b.run_device(d, n, out, sync=False, d_count=d_count)
        ws.sync(); wall = (time.perf_counter() - t0) / 50 * 1e3
        ms, l = ws.kernel_time(True); ws.enable_timing(False)
        print(f"canonical={canon} 256 Mbp nblk={nblk or 'default'}: kernel {ms / l:.4f} ms ({n / (ms / l) / 1e6:.0f} Gbases/s), call {wall:.4f} ms ({n / wall / 1e6:.0f} Gbases/s)", flush=True)
ws.set_blocks_per_lane(0)
