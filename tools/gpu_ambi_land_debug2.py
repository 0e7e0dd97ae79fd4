"""The failing case of test_skip_ambiguous_small_sweep (n=333, k=1, w=55) and neighbours, got vs want."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
import numpy as np
import mm_oracle as oracle
import simd_minimizers_amd as sm
from test_gpu_parity import _ascii_with_n
rng = np.random.default_rng(21)
cases = []
for n, frac, runs in [(100, 0.01, False), (100, 0.06, False), (333, 0.05, True)]:
    a = _ascii_with_n(rng, n, frac, runs)
    cases.append((n, a))
n, a = cases[2]
print("Ns at", np.flatnonzero(a == ord("N")).tolist())
nseq = sm.PackedNSeqVec.from_ascii(a.tobytes())
packed, amb = oracle.pack_ascii_n(a.tobytes())
for k, w in ((1, 55), (1, 52), (3, 55), (1, 49), (5, 43), (1, 37), (9, 55)):
    if (k + w) % 2: continue
    b = sm.Builder(k, w, True, 0)
    want = list(map(int, oracle.run_skip_ambiguous(packed, amb, n, k, w)))
    got = b.run_skip_ambiguous_windows_once(nseq)
    print(f"k={k} w={w}: {'ok' if got == want else 'DIFFER'} got tail {got[-6:]} want tail {want[-6:]} (len {len(got)} / {len(want)})", flush=True)
