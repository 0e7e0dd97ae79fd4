"""Kernel time of the BASELINE configurations (HIP events, steady clocks); run once with MM_PIPE=0 and
once with MM_PIPE=1 (the switch is read once per process).  Timing aid, not part of the product."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import simd_minimizers_amd as sm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_100_000_000
d = sm.generate_device(n, 3); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.36) + 1024, dtype=torch.int32, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
def t(b, warm=12, reps=12):
    for _ in range(warm): b.run_device(d, n, out, sync=False, d_count=cnt)
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, sync=False, d_count=cnt)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    ws.check()
    return ms / l
cfgs = [(21, 11, True, 0), (21, 11, False, 0), (15, 10, False, 0), (5, 7, False, 0), (31, 5, True, 0), (15, 15, True, 1), (21, 16, False, 0), (13, 13, True, 2)]
tag = f"MM_PIPE={os.environ.get('MM_PIPE', '1')} FWD_BLOCKS={os.environ.get('MM_PIPE_FWD_BLOCKS', '-')} PER_CU={os.environ.get('MM_PIPE_PER_CU', '-')}"
for (k, w, canon, mode) in cfgs:
    b = sm.Builder(k, w, canon, mode)
    ms = t(b)
    c = int(cnt.item())
    print(f"{tag}: k={k} w={w} canon={canon} mode={mode}: {ms:.3f} ms -> {n/ms/1e6:.0f} Gbases/s  count={c} sum={int(out[:c].to(torch.int64).sum().item())}", flush=True)
