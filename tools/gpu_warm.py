"""Kernel time of consecutive launches after idle: how long does the GPU take to reach steady clocks?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import simd_minimizers_amd as sm
n = 3_100_000_000
d = sm.generate_device(n, 3); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
b = sm.canonical_minimizers(21, 11)
torch.cuda.synchronize(); time.sleep(2.0)
evs = [torch.cuda.Event(enable_timing=True) for _ in range(121)]
evs[0].record()
for i in range(120):
    b.run_device(d, n, out, sync=False)
    evs[i + 1].record()
torch.cuda.synchronize()
ts = [evs[i].elapsed_time(evs[i + 1]) for i in range(120)]
print("ms per step:", " ".join(f"{x:.2f}" for x in ts[:30]))
print("steps 30-60 mean %.3f, 60-120 mean %.3f, min %.3f" % (sum(ts[30:60]) / 30, sum(ts[60:]) / 60, min(ts)))
