"""One-off parity sweep over EVERY window size the fused family serves (1 .. 128), prebuilt or specialised at run time:
canonical and forward minimizers on a tie-heavy sequence, two base offsets, two lane lengths, against the oracle.  Round 5
found a wrong tie resolution at w = 49 / 65 - sizes no sweep of the suite had reached.  usage: gpu_all_w_parity.py [w_from] [w_to]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
import numpy as np, torch
import mm_oracle as oracle
import simd_minimizers_amd as sm
w0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w1 = int(sys.argv[2]) if len(sys.argv) > 2 else 128
rng = np.random.default_rng(128)
ws = sm.default_workspace(0)
n = 300_007
codes = rng.integers(0, 4, size=n + 8).astype(np.uint8)
for s in range(0, n, 30_000):
    m = min(12_000, n - s)
    codes[s:s + m] = rng.integers(0, 2, size=m) * 3
    m2 = max(0, min(6_000, n - s - 15_000))
    codes[s + 15_000:s + 15_000 + m2] = rng.integers(0, 2, size=m2) + 1
def pack(off):
    packed = np.zeros((n + off + 3) // 4 + 64, dtype=np.uint8)
    sh = np.concatenate([np.zeros(off, dtype=np.uint8), codes[:n]])
    for j in range(4):
        c = sh[j::4]; packed[: len(c)] |= (c << (2 * j)).astype(np.uint8)
    return packed
packs = {off: pack(off) for off in (0, 3)}
dev = {off: torch.from_numpy(p).cuda() for off, p in packs.items()}
out = torch.zeros(n, dtype=torch.int32, device="cuda")
bad, done, t0 = [], 0, time.time()
for w in range(w0, w1 + 1):
    for canonical in (True, False):
        k = 19 if (not canonical or (19 + w - 1) % 2 == 1) else 20
        b = sm.Builder(k, w, canonical, 0)
        for off in (0, 3):
            want = oracle.run(packs[off], n, k, w, canonical=canonical, base_offset=off)
            for nb in (0, 7):
                ws.set_blocks_per_lane(nb)
                c = b.run_device(dev[off], n, out, base_offset=off)
                fused = ws.last_path() == sm.PATH_FUSED
                got = out[:c].cpu().numpy().view(np.uint32)
                if not (fused and len(got) == len(want) and np.array_equal(got, want)):
                    bad.append((w, canonical, k, off, nb, fused, len(got), len(want)))
                done += 1
        ws.set_blocks_per_lane(0)
    if w % 8 == 0:
        print(f"w <= {w}: {done} runs, {len(bad)} bad, {time.time() - t0:.0f} s", flush=True)
print("bad:", bad)
print(f"{done} runs, {len(bad)} mismatches")
