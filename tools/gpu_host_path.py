"""PCIe-inclusive rate of the host entry point mm_run_host (H2D + kernel + D2H), buffers preallocated
and touched, steady state."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import mm_oracle as oracle
import simd_minimizers_amd as sm
L = sm.lib()
ws = sm.default_workspace(0)
for n, pinned in [(16 << 20, False), (256 << 20, False), (256 << 20, True), (1 << 30, False), (1 << 30, True)]:
    data = oracle.gen_packed(2, n)
    if pinned:
        pd, own1 = sm.pinned_array(data.shape, np.uint8); pd[:] = data; data = pd
    plan = sm.Plan(21, 11, True, 0, None)
    cap = int(n * 0.2)
    pos = np.ones(cap, dtype=np.uint32)
    if pinned:
        pos, own2 = sm.pinned_array((cap,), np.uint32); pos[:] = 1
    cnt = C.c_uint64()
    u8p, u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)
    def run():
        sm._check(L.mm_run_host(plan.h, ws.h, data.ctypes.data_as(u8p), 0, n, pos.ctypes.data_as(u32p), None, cap, C.byref(cnt)))
    run(); run()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps): run()
    dt = (time.perf_counter() - t0) / reps
    print(f"n={n} pinned={pinned}: {dt*1e3:.2f} ms per call, {n/dt/1e9:.1f} Gbases/s, {cnt.value} positions "
          f"({(n/4 + 4*cnt.value)/dt/1e9:.1f} GB/s over PCIe)", flush=True)
