"""mm_run_host on a COLD link: a fresh process runs the device-resident kernel for two seconds (no host <-> device traffic,
like bench.py before its end_to_end figure), then calls mm_run_host eight times with the mechanism given on the command line
(MM_HOST_OUT = engine | blit | direct, MM_HOST_IN = engine | blit) and prints every call's time.  Round 5: with the copy
engines the first five calls of such a process take 71 ms and the sixth 41 - something outside the library ramps up under
sustained copy-engine traffic; this script asks whether copy KERNELS are exempt.
usage: gpu_host_cold.py <out mode> [<in mode>] [idle|busy]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MM_HOST_OUT"] = sys.argv[1] if len(sys.argv) > 1 else "engine"
os.environ["MM_HOST_IN"] = sys.argv[2] if len(sys.argv) > 2 else "engine"
before = sys.argv[3] if len(sys.argv) > 3 else "busy"
import numpy as np, torch
import simd_minimizers_amd as sm
n = 3_100_000_000
ws = sm.default_workspace(0); L = sm.lib()
d = sm.generate_device(n, 3)
nb_in = (n + 3) // 4
hp, _o1 = sm.pinned_array((nb_in + 64,), np.uint8)
# (the input comes from a host-side generator: no device -> host copy warms the link)
rng = np.random.default_rng(1)
blk = rng.integers(0, 256, 1 << 24, dtype=np.uint8)
for o in range(0, nb_in + 64, 1 << 24):
    m = min(1 << 24, nb_in + 64 - o); hp[o:o + m] = blk[:m]
n_cap = int(n * 0.18)
ho, _o2 = sm.pinned_array((n_cap,), np.uint32); ho[:] = 0
b = sm.canonical_minimizers(21, 11); plan = b.plan(); cnt = C.c_uint64()
out = torch.empty(n_cap, dtype=torch.int32, device="cuda")
if before == "busy":
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 2.0:
        for _ in range(16): b.run_device(d, n, out, sync=False)
        torch.cuda.synchronize()
else:
    torch.cuda.synchronize(); time.sleep(2.0)
ts = []
for i in range(8):
    t0 = time.perf_counter()
    sm._check(L.mm_run_host(plan.h, ws.h, hp.ctypes.data_as(C.POINTER(C.c_uint8)), 0, n, ho.ctypes.data_as(C.POINTER(C.c_uint32)), None, n_cap, C.byref(cnt)))
    ts.append((time.perf_counter() - t0) * 1e3)
print(f"out={os.environ['MM_HOST_OUT']:<6} in={os.environ['MM_HOST_IN']:<6} after 2 s of {'device-resident kernels' if before == 'busy' else 'idling'}: "
      + " ".join(f"{t:.1f}" for t in ts) + f" ms  ({cnt.value} positions)", flush=True)
