import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
import torch
import simd_minimizers_amd as sm
n = 3_100_000_000
d = sm.generate_device(n, 3); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.19), dtype=torch.int32, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
def kt(b, reps=10):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        b.run_device(d, n, out, sync=False, d_count=cnt); ws.sync()
    ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, sync=False, d_count=cnt)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l
for (k, w, canon) in ((21, 11, True), (21, 11, False), (31, 51, True)):
    b = sm.Builder(k, w, canon, 0)
    row = []
    for t in ("0", "1", "0", "1"):
        if t == "1": os.environ["MM_FORCE_TICKET"] = "1"
        else: os.environ.pop("MM_FORCE_TICKET", None)
        row.append(f"{'ticket' if t == '1' else 'index '} {kt(b):.4f}")
    os.environ.pop("MM_FORCE_TICKET", None)
    print(f"k={k} w={w} canonical={canon}: " + " | ".join(row), flush=True)
