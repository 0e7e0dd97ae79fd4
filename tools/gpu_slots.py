"""Same tiles, fewer resident workgroups per CU (LDS padding): is the kernel limited by slots?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")  # this script flips MM_* switches between runs (mm_env.h)
import simd_minimizers_amd as sm
n = 3_100_000_000
d = sm.generate_device(n, 3); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
b = sm.Builder(21, 11, True, 0)
def t(warm=15, reps=15):
    for _ in range(warm): b.run_device(d, n, out, sync=False)
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, sync=False)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l
for nblk, pad in [(24, 0), (24, 14000), (24, 40000), (18, 0), (18, 2000), (18, 9000), (18, 22000), (12, 0), (12, 3000), (12, 8500), (12, 16000)]:
    ws.set_blocks_per_lane(nblk)
    os.environ["MM_LDS_PAD"] = str(pad)
    S = 11 * nblk; lds = (int(1.3 / 6 * S) + 19) * 516 + pad
    os.environ["MM_DEBUG"] = "0"; full = t()
    os.environ["MM_DEBUG"] = "3"; p1 = t()
    os.environ["MM_DEBUG"] = "0"
    print(f"nblk={nblk} pad={pad} lds={lds/1024:.1f}KB slots={min(5, int(160*1024/(lds+400)))}: full {full:.3f} ms, phase 1 only {p1:.3f} ms", flush=True)
