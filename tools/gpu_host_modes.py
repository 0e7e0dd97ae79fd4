"""mm_run_host on the headline sequence with page-locked caller buffers under every mechanism of its two legs
(MM_HOST_OUT = engine | blit | direct, MM_HOST_IN = engine | blit, MM_PIPE_CHUNKS): ms per call, outputs compared with the
device-resident run.  The A/B of VERDICT r4 item 1; run through tools/host_link_diag.sh on whatever box the call lands on."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
import numpy as np, torch
import simd_minimizers_amd as sm
n = int(os.environ.get("MM_N", "3100000000"))
ws = sm.default_workspace(0); L = sm.lib()
d = sm.generate_device(n, 3)
nb_in = (n + 3) // 4
hp, _o1 = sm.pinned_array((nb_in + 64,), np.uint8); hp[:] = d.cpu().numpy()[: nb_in + 64]
n_cap = int(n * 0.1667) + 2048
ho, _o2 = sm.pinned_array((n_cap,), np.uint32)
dev_out = torch.empty(n_cap, dtype=torch.int32, device="cuda")
b = sm.canonical_minimizers(21, 11)
want_n = b.run_device(d, n, dev_out)
want = dev_out[:want_n].cpu().numpy().view(np.uint32)
del dev_out
plan = b.plan(); cnt = C.c_uint64()
def call():
    t0 = time.perf_counter()
    sm._check(L.mm_run_host(plan.h, ws.h, hp.ctypes.data_as(C.POINTER(C.c_uint8)), 0, n, ho.ctypes.data_as(C.POINTER(C.c_uint32)), None, n_cap, C.byref(cnt)))
    return (time.perf_counter() - t0) * 1e3
# (out, in, chunks, uploads in flight, downloads in flight); None = the library's default
modes = [("engine", "engine", None, None, None), ("engine", "engine", 8, None, None), ("engine", "engine", 32, None, None),
         ("engine", "engine", 16, 1, 1), ("engine", "engine", 16, 2, 1), ("engine", "engine", 16, 1, 2), ("engine", "engine", 16, 3, 3),
         ("engine", "engine", 16, 0, 0), ("engine", "engine", 8, 0, 0), ("engine", "engine", 32, 2, 2), ("engine", "engine", 64, 2, 2),
         ("blit", "engine", None, None, None), ("blit", "engine", 32, None, None), ("blit", "engine", 16, 0, 0),
         ("direct", "engine", None, None, None), ("engine", "blit", None, None, None), ("blit", "blit", None, None, None)]
for out_m, in_m, chunks, fin, fout in modes:
    os.environ["MM_HOST_OUT"] = out_m; os.environ["MM_HOST_IN"] = in_m
    for name, v in (("MM_PIPE_CHUNKS", chunks), ("MM_PIPE_IN_FLIGHT", fin), ("MM_PIPE_OUT_FLIGHT", fout)):
        if v is None: os.environ.pop(name, None)
        else: os.environ[name] = str(v)
    ho[:] = 0
    ts = [call() for _ in range(4)]
    ok = cnt.value == want_n and np.array_equal(ho[:want_n], want)
    print(f"out={out_m:<6} in={in_m:<6} chunks={chunks or 'default':<7} in flight: uploads {'default' if fin is None else fin or 'all':<7} "
          f"downloads {'default' if fout is None else fout or 'all':<7}: {min(ts[1:]):7.2f} ms best of 3 (first {ts[0]:.1f}); "
          f"outputs {'== device run' if ok else 'DIFFER'}", flush=True)
for name in ("MM_HOST_OUT", "MM_HOST_IN", "MM_PIPE_CHUNKS", "MM_PIPE_IN_FLIGHT", "MM_PIPE_OUT_FLIGHT"):
    os.environ.pop(name, None)
os.environ["MM_NO_PIPELINE"] = "1"
ts = [call() for _ in range(3)]
print(f"one shot (MM_NO_PIPELINE=1): {min(ts[1:]):7.2f} ms")
