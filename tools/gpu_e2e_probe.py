"""Where the PCIe-inclusive call's time goes on this box: each direction alone, both together on two streams, the
one-shot and the pipelined mm_run_host."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
import simd_minimizers_amd as sm
n = int(os.environ.get("MM_N", "3100000000"))
ws = sm.default_workspace(0); L = sm.lib()
d = sm.generate_device(n, 3)
nb_in = (n + 3) // 4
hp, _o1 = sm.pinned_array((nb_in + 64,), np.uint8); hp[:] = d.cpu().numpy()[: nb_in + 64]
n_out = int(n * 0.1667) + 1024
ho, _o2 = sm.pinned_array((n_out + 1024,), np.uint32); ho[:] = 0
dev_out = torch.empty(n_out, dtype=torch.int32, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
th_in = torch.from_numpy(hp[:nb_in]); th_out = torch.from_numpy(ho[:n_out].view(np.int32))
def t(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
def h2d():
    with torch.cuda.stream(s1): d[:nb_in].copy_(th_in, non_blocking=True)
def d2h():
    with torch.cuda.stream(s2): th_out.copy_(dev_out, non_blocking=True)
def both():
    h2d(); d2h()
a, b, c = t(h2d), t(d2h), t(both)
print(f"H2D {nb_in / 1e6:.0f} MB alone {a:.1f} ms ({nb_in / a / 1e6:.1f} GB/s); D2H {4 * n_out / 1e6:.0f} MB alone {b:.1f} ms ({4 * n_out / b / 1e6:.1f} GB/s); both on two streams {c:.1f} ms (sum {a + b:.1f}, max {max(a, b):.1f})")
plan = sm.canonical_minimizers(21, 11).plan(); cnt = C.c_uint64()
def run():
    sm._check(L.mm_run_host(plan.h, ws.h, hp.ctypes.data_as(C.POINTER(C.c_uint8)), 0, n, ho.ctypes.data_as(C.POINTER(C.c_uint32)), None, n_out + 1024, C.byref(cnt)))
run()
print(f"mm_run_host pipelined: {t(run):.1f} ms")
os.environ["MM_NO_PIPELINE"] = "1"
run()
print(f"mm_run_host one shot (MM_NO_PIPELINE=1): {t(run):.1f} ms")
del os.environ["MM_NO_PIPELINE"]
# what bench.py does before its end_to_end figure: a workspace of its own, timed steps, the clock probe, the 1/8 shard
ws2 = sm.Workspace(0, torch.cuda.current_stream().cuda_stream)
b = sm.canonical_minimizers(21, 11).workspace(ws2)
out = torch.empty(n_out, dtype=torch.int32, device="cuda"); dc = torch.zeros(1, dtype=torch.int64, device="cuda")
def run2():
    sm._check(L.mm_run_host(plan.h, ws2.h, hp.ctypes.data_as(C.POINTER(C.c_uint8)), 0, n, ho.ctypes.data_as(C.POINTER(C.c_uint32)), None, n_out + 1024, C.byref(cnt)))
run2(); print(f"second workspace, pipelined: {t(run2):.1f} ms")
for _ in range(20): b.run_device(d, n, out, sync=False, d_count=dc)
torch.cuda.synchronize()
print(f"after 20 device steps: {t(run2):.1f} ms")
ws2.enable_timing(True); ws2.kernel_time(True)
for _ in range(5): b.run_device(d, n, out, sync=False, d_count=dc)
torch.cuda.synchronize(); ws2.kernel_time(True); ws2.enable_timing(False)
print(f"after timed steps (events): {t(run2):.1f} ms")
ws2.clock_probe_begin(10000)
for _ in range(8): b.run_device(d, n, out, sync=False, d_count=dc)
torch.cuda.synchronize(); print("clock", ws2.clock_probe_end())
print(f"after the clock probe: {t(run2):.1f} ms")
big = [torch.empty(1 << 30, dtype=torch.uint8, device="cuda") for _ in range(6)]
print(f"with 6 GiB more allocated: {t(run2):.1f} ms")
