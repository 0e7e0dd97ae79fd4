"""Three mm_run_host calls on the headline sequence with page-locked caller buffers and nothing else: the program to put
behind `rocprofv3 --memory-copy-trace --kernel-trace` (tools/trace_host_path.py condenses the trace)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import simd_minimizers_amd as sm
n = int(os.environ.get("MM_N", "3100000000"))
ws = sm.default_workspace(0); L = sm.lib()
d = sm.generate_device(n, 3)
nb_in = (n + 3) // 4
hp, _o1 = sm.pinned_array((nb_in + 64,), np.uint8); hp[:] = d.cpu().numpy()[: nb_in + 64]
n_out = int(n * 0.1667) + 1024
ho, _o2 = sm.pinned_array((n_out + 1024,), np.uint32); ho[:] = 0
plan = sm.canonical_minimizers(21, 11).plan(); cnt = C.c_uint64()
for i in range(int(os.environ.get("MM_CALLS", "3"))):
    t0 = time.perf_counter()
    sm._check(L.mm_run_host(plan.h, ws.h, hp.ctypes.data_as(C.POINTER(C.c_uint8)), 0, n, ho.ctypes.data_as(C.POINTER(C.c_uint32)), None, n_out + 1024, C.byref(cnt)))
    print(f"call {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms, {cnt.value} positions", flush=True)
