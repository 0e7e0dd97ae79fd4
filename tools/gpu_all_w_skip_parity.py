"""One-off parity sweep of the SKIP-AMBIGUOUS walk over every window size the fused family serves (1 .. 128): the dirty walk takes
three different routes to its window bits (registers below w = 32, one 16-byte load to LDS per block for 32 .. 37, chunks of rows
for 38 .. 96, registers again above) and two emit bodies below w = 13.  Canonical minimizers on a sequence with isolated Ns and
N runs, two base / bit offsets, three lane lengths, against the oracle.  usage: gpu_all_w_skip_parity.py [w_from] [w_to]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import mm_oracle as oracle
import simd_minimizers_amd as sm
w0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w1 = int(sys.argv[2]) if len(sys.argv) > 2 else 128
rng = np.random.default_rng(129)
ws = sm.default_workspace(0)
n = 300_007
packs = {}
for off in (0, 3):
    a = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n + off + 8)].copy()
    a[rng.integers(0, n, size=n // 150)] = ord("N")
    for s in rng.integers(0, n - 3000, size=12):
        a[s:s + int(rng.integers(1, 400))] = ord("N")
    for s in rng.integers(0, n - 9000, size=10):                       # clean stretches (whole clean waves among dirty ones)
        m = int(rng.integers(2000, 9000))
        a[s:s + m] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=m)]
    packed, amb = oracle.pack_ascii_n(a.tobytes())
    packs[off] = (packed, amb, torch.from_numpy(packed).cuda(), torch.from_numpy(amb).cuda())
out = torch.zeros(n, dtype=torch.int32, device="cuda")
bad, done, t0 = [], 0, time.time()
for w in range(w0, w1 + 1):
    k = 19 if (19 + w - 1) % 2 == 1 else 20
    b = sm.canonical_minimizers(k, w)
    for off in (0, 3):
        packed, amb, d_p, d_m = packs[off]
        want = oracle.run_skip_ambiguous(packed, amb, n, k, w, base_offset=off, amb_offset=off)
        for nb in (0, 3, 10):
            ws.set_blocks_per_lane(nb)
            c = b.run_skip_ambiguous_device(d_p, d_m, n, out, base_offset=off, amb_offset=off)
            fused = ws.last_path() == sm.PATH_FUSED
            got = out[:c].cpu().numpy().view(np.uint32)
            if not (fused and len(got) == len(want) and np.array_equal(got, want)):
                bad.append((w, k, off, nb, fused, len(got), len(want)))
            done += 1
    ws.set_blocks_per_lane(0)
    if w % 8 == 0:
        print(f"w <= {w}: {done} runs, {len(bad)} bad, {time.time() - t0:.0f} s", flush=True)
print("bad:", bad)
print(f"{done} runs, {len(bad)} mismatches")
