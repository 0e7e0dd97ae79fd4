"""Timing of the steps either side of the hot path: ASCII packing, values, host API (PCIe)."""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import simd_minimizers_amd as sm
ws = sm.default_workspace(0)
n = 1 << 30
asc = torch.randint(0, 4, (n,), dtype=torch.uint8, device="cuda")
lut = torch.tensor(list(b"ACTG"), dtype=torch.uint8, device="cuda")
asc = lut[asc.long()] if False else (asc * 0 + 65)  # plain 'A's are enough for bandwidth
packed = torch.zeros(n // 4 + 64, dtype=torch.uint8, device="cuda")
def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
t = timed(lambda: sm._check(sm.lib().mm_pack_ascii_device_async(ws.h, C.c_void_p(asc.data_ptr()), n, C.c_void_p(packed.data_ptr()))))
print(f"pack_ascii {n} bases: {t*1e3:.3f} ms  {n/t/1e9:.1f} Gbase/s  {(n*1.25)/t/1e9:.0f} GB/s")
d = sm.generate_device(n, 5)
out = torch.zeros(int(n * 0.2), dtype=torch.int32, device="cuda")
b = sm.canonical_minimizers(21, 11)
c = b.run_device(d, n, out)
vals = torch.zeros(c, dtype=torch.int64, device="cuda")
t = timed(lambda: sm._check(sm.lib().mm_values_u64_device_async(ws.h, C.c_void_p(d.data_ptr()), d.numel(), 0, n, 21, 1, C.c_void_p(out.data_ptr()), c, C.c_void_p(vals.data_ptr()))))
print(f"values_u64 {c} positions: {t*1e3:.3f} ms  {c/t/1e9:.2f} Gvalues/s")
# host API (PCIe inclusive) on 256 Mbp
m = 1 << 28
host = d[: m // 4 + 16].cpu().numpy()
ps = sm.PackedSeq(host, 0, m)
t0 = time.perf_counter(); pos, _ = b._run_arrays(ps); t1 = time.perf_counter()
t0 = time.perf_counter(); pos, _ = b._run_arrays(ps); t1 = time.perf_counter()
print(f"host API (H2D + kernel + D2H) {m} bases: {(t1-t0)*1e3:.1f} ms  {m/(t1-t0)/1e9:.1f} Gbase/s")
