"""Plain canonical runs over tie-heavy sequences for window sizes with more than two 16-base views per block (w > 32),
several lane lengths, against the oracle: is the lazy strand vote right for every view count?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
import numpy as np, torch
import mm_oracle as oracle
import simd_minimizers_amd as sm
rng = np.random.default_rng(7)
ws = sm.default_workspace(0)
n = 600_011
codes = rng.integers(0, 4, size=n).astype(np.uint8)
# stretches of a two-letter alphabet and tandem repeats: hash ties at the minimum in most windows
for s in range(0, n, 50_000):
    codes[s:s + 20_000] = rng.integers(0, 2, size=min(20_000, n - s)) * 3          # A / G only
    unit = rng.integers(0, 4, size=7)
    codes[s + 25_000:s + 35_000] = np.resize(unit, min(10_000, max(0, n - s - 25_000)))[: max(0, min(10_000, n - s - 25_000))]
packed = np.zeros((n + 3) // 4 + 64, dtype=np.uint8)
for j in range(4):
    c = codes[j::4]; packed[: len(c)] |= (c << (2 * j)).astype(np.uint8)
d = torch.from_numpy(packed).cuda()
out = torch.zeros(n, dtype=torch.int32, device="cuda")
bad = 0
for (k, w) in ((19, 33), (19, 35), (19, 39), (19, 41), (19, 49), (19, 51), (19, 63), (19, 65), (21, 79), (19, 81), (19, 97), (20, 64), (21, 127)):
    want = oracle.run(packed, n, k, w, canonical=True)
    res = []
    for nb in (0, 3, 4, 6, 9):
        ws.set_blocks_per_lane(nb)
        c = sm.canonical_minimizers(k, w).run_device(d, n, out)
        got = out[:c].cpu().numpy().view(np.uint32)
        ok = len(got) == len(want) and np.array_equal(got, want)
        res.append(f"{nb or 'default'}:{'ok' if ok else f'DIFF({len(got)}/{len(want)})'}")
        bad += 0 if ok else 1
    ws.set_blocks_per_lane(0)
    print(f"k={k} w={w}: " + " ".join(res), flush=True)
print("bad", bad)
