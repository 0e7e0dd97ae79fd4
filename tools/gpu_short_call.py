"""Cost of ONE short sequence per call (Builder::run in a per-read loop, src/lib.rs:378; the reference: 2-20 ns/base for
reads below 1 kb, bench/results-neon.json experiment "short"): the synchronous device call and the host call on a 150 bp
and a 1 kbp read, microseconds per call over many calls; beside them the batched reads entry point."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")  # (MM_NO_SMALL_HOST is flipped between the two host columns)
import simd_minimizers_amd as sm
ws = sm.default_workspace(0)
L = sm.lib()
d = sm.generate_device(1 << 20, 3)
host = d.cpu().numpy()
out = torch.zeros(16384, dtype=torch.int32, device="cuda")
b = sm.canonical_minimizers(21, 11)
plan = b.plan()
for n in (150, 1000, 10000, 60000):
    for _ in range(200): b.run_device(d, n, out)
    t0 = time.perf_counter(); reps = 3000
    for _ in range(reps): b.run_device(d, n, out)
    dev_us = (time.perf_counter() - t0) / reps * 1e6
    # raw C-ABI call without the Python wrapper's work
    cnt = C.c_uint64()
    dp, op = C.c_void_p(d.data_ptr()), C.c_void_p(out.data_ptr())
    t0 = time.perf_counter()
    for _ in range(reps): L.mm_run_device(plan.h, ws.h, dp, d.numel(), 0, n, 0, sm.U64_MAX, op, None, 16384, C.byref(cnt))
    abi_us = (time.perf_counter() - t0) / reps * 1e6
    hp = np.ascontiguousarray(host[: (n + 3) // 4 + 8]); ho = np.zeros(16384, dtype=np.uint32)
    t0 = time.perf_counter()
    for _ in range(reps): L.mm_run_host(plan.h, ws.h, hp.ctypes.data_as(C.POINTER(C.c_uint8)), 0, n, ho.ctypes.data_as(C.POINTER(C.c_uint32)), None, 16384, C.byref(cnt))
    host_us = (time.perf_counter() - t0) / reps * 1e6
    os.environ["MM_NO_SMALL_HOST"] = "1"   # the path of rounds 1-4: upload, launch, download through the runtime
    t0 = time.perf_counter()
    for _ in range(reps): L.mm_run_host(plan.h, ws.h, hp.ctypes.data_as(C.POINTER(C.c_uint8)), 0, n, ho.ctypes.data_as(C.POINTER(C.c_uint32)), None, 16384, C.byref(cnt))
    old_us = (time.perf_counter() - t0) / reps * 1e6
    del os.environ["MM_NO_SMALL_HOST"]
    print(f"n={n}: Builder.run_device (python) {dev_us:.1f} us, mm_run_device (C ABI through ctypes) {abi_us:.1f} us = {abi_us * 1e3 / n:.0f} ns/base, mm_run_host {host_us:.1f} us = {host_us * 1e3 / n:.0f} ns/base (through the runtime's copies: {old_us:.1f} us)", flush=True)
