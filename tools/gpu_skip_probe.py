"""Instrumented single case of the skip-ambiguous path: explicit timers, counts, checksums and a
negative control, so that parity evidence is numbers rather than a pytest dot."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import mm_oracle as oracle
import simd_minimizers_amd as sm

def T():
    torch.cuda.synchronize(); return time.perf_counter()

k, w, mode, n = 21, 11, 0, 4_000_003
rng = np.random.default_rng(7)
t0 = time.perf_counter()
a = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].copy()
a[rng.integers(0, n, size=8000)] = ord("N")
for s in rng.integers(0, n - 70_000, size=12):
    a[s:s + int(rng.integers(1, 70_000))] = ord("n")
t1 = time.perf_counter()
packed, amb = oracle.pack_ascii_n(a.tobytes())
t2 = time.perf_counter()
want = oracle.run_skip_ambiguous(packed, amb, n, k, w, mode=mode)
t3 = time.perf_counter()
plain = oracle.run(packed, n, k, w, canonical=True, mode=mode)
t4 = time.perf_counter()
print(f"host: gen {t1-t0:.3f}s  pack_n {t2-t1:.3f}s  oracle skip {t3-t2:.3f}s ({(t3-t2)/n*1e9:.1f} ns/base)  "
      f"oracle plain {t4-t3:.3f}s", flush=True)
isn = (a & 0xDF)[:, None] != np.frombuffer(b"ACGT", dtype=np.uint8)[None, :]
isn = isn.all(axis=1)
l = k + w - 1
cs = np.concatenate([[0], np.cumsum(isn)])
skipped_windows = int(((cs[l:] - cs[:-l]) > 0).sum())
print(f"n={n} Ns={int(isn.sum())} windows={n-l+1} skipped_windows={skipped_windows} "
      f"|want|={len(want)} |plain|={len(plain)} checksum(want)={oracle.checksum(want)}", flush=True)
# independent numpy check of the oracle's own output: no k-mer at an output position holds an N
assert not (cs[want.astype(np.int64) + k] - cs[want.astype(np.int64)]).any()

ws = sm.default_workspace(0)
d_a = torch.from_numpy(a).cuda()
d_p = torch.zeros((n + 3) // 4 + 64, dtype=torch.uint8, device="cuda")
d_m = torch.zeros((n + 7) // 8 + 64, dtype=torch.uint8, device="cuda")
t5 = T()
sm._check(sm.lib().mm_pack_ascii_n_device_async(ws.h, C.c_void_p(d_a.data_ptr()), n,
                                                C.c_void_p(d_p.data_ptr()), C.c_void_p(d_m.data_ptr())))
t6 = T()
b = sm.Builder(k, w, True, mode)
out = torch.zeros(n, dtype=torch.int32, device="cuda")
c = b.run_skip_ambiguous_device(d_p, d_m, n, out)
t7 = T()
got = out[:c].cpu().numpy().view(np.uint32)
t8 = time.perf_counter()
print(f"gpu: pack_n {t6-t5:.4f}s  run(sync) {t7-t6:.4f}s  copy-back {t8-t7:.4f}s  path={ws.last_path()} "
      f"count={c} checksum(got)={oracle.checksum(got)}", flush=True)
print("packed equal:", np.array_equal(d_p[:(n+3)//4].cpu().numpy(), packed[:(n+3)//4]),
      " amb equal:", np.array_equal(d_m[:(n+7)//8].cpu().numpy(), amb[:(n+7)//8]))
print("first 6 want/got:", want[:6], got[:6], " last 3:", want[-3:], got[-3:])
print("PARITY:", np.array_equal(got, want), "  differs from plain run:", not np.array_equal(got, plain))
# negative control: one cleared ambiguity bit on the HOST copy only -> oracle output must change
amb2 = amb.copy()
first_n = int(np.flatnonzero(isn)[100])
amb2[first_n >> 3] &= ~np.uint8(1 << (first_n & 7))
want2 = oracle.run_skip_ambiguous(packed, amb2, n, k, w, mode=mode)
print(f"control (host amb bit {first_n} cleared): equal to GPU = {np.array_equal(got, want2)} "
      f"(must be False), |want2|={len(want2)}")
# generic family on the same input
ws.force_generic(True)
c2 = b.run_skip_ambiguous_device(d_p, d_m, n, out)
g2 = out[:c2].cpu().numpy().view(np.uint32)
ws.force_generic(False)
print("generic path:", ws.last_path(), "parity:", np.array_equal(g2, want))
