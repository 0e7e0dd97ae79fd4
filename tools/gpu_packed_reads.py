"""Reads packed back to back (mm_run_packed_reads_device) against reads at a fixed stride (mm_run_reads_device): 8 M x 150 bp and
8 M reads of 100..200 bp, canonical k=21 w=11, kernel time by HIP events."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import simd_minimizers_amd as sm
ws = sm.default_workspace(0); L = sm.lib()
n_reads = 8_000_000
b = sm.canonical_minimizers(21, 11)
for name, lens in (("150 bp", torch.full((n_reads,), 150, dtype=torch.int64)), ("100..200 bp", torch.randint(100, 201, (n_reads,), dtype=torch.int64))):
    starts = torch.zeros(n_reads + 1, dtype=torch.int64); starts[1:] = torch.cumsum(lens, 0)
    total = int(starts[-1]); mx = int(lens.max())
    d = sm.generate_device(total, 7)
    ds = starts.cuda()
    out = torch.empty(int(total * 0.2), dtype=torch.int32, device="cuda")
    offs = torch.zeros(n_reads + 1, dtype=torch.int64, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    def step():
        sm._check(L.mm_run_packed_reads_device_async(b.plan().h, ws.h, C.c_void_p(d.data_ptr()), d.numel(), 0, n_reads, C.c_void_p(ds.data_ptr()),
                                                     total, mx, C.c_void_p(out.data_ptr()), None, out.numel(), C.c_void_p(offs.data_ptr()), C.c_void_p(cnt.data_ptr())))
    plan = b.plan()
    for _ in range(10): step()
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(10): step()
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    print(f"packed reads, {name}: {total} bases, kernel {ms / l:.3f} ms = {total / (ms / l) / 1e6:.0f} Gbases/s, {int(cnt.item())} positions", flush=True)
    if name == "150 bp":
        def step2():
            sm.run_reads_device(b, d, n_reads, 150, 150, out, offs, d_count=cnt, sync=False)
        for _ in range(10): step2()
        ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
        for _ in range(10): step2()
        ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
        print(f"fixed stride, 150 bp: kernel {ms / l:.3f} ms = {total / (ms / l) / 1e6:.0f} Gbases/s", flush=True)
    del d, out
