import os, sys
sys.path.insert(0, "/root/repo")
import torch, time
import simd_minimizers_amd as sm
n = 3_100_000_000
d = sm.generate_device(n, 3); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.19), dtype=torch.int32, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
canon = os.environ.get("MM_CANON", "0") == "1"
b = sm.Builder(21, 11, canon, 0)
def kt(reps=8):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        b.run_device(d, n, out, sync=False, d_count=cnt); ws.sync()
    ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, sync=False, d_count=cnt)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l
for nb in [int(x) for x in (sys.argv[1:] or "0 22 20 17 14 11 8 0".split())]:
    ws.set_blocks_per_lane(nb)
    print(f"{'canonical' if canon else 'forward'} k=21 w=11 3.1 Gbp, blocks per lane {nb or 'default'}: {kt():.4f} ms", flush=True)
