"""One-off parity sweep of the OTHER emit flavours of the fused family over the window sizes: minimizers with super-k-mer
indices (packed 16-bit list entries whose shift depends on w), closed and open syncmers - canonical and forward, a tie-heavy
sequence, two lane lengths, positions (and indices) against the oracle.  usage: gpu_all_w_modes_parity.py [w_from] [w_to]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import mm_oracle as oracle
import simd_minimizers_amd as sm
w0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w1 = int(sys.argv[2]) if len(sys.argv) > 2 else 72
rng = np.random.default_rng(130)
ws = sm.default_workspace(0)
n = 250_003
codes = rng.integers(0, 4, size=n + 8).astype(np.uint8)
for s in range(0, n, 30_000):
    m = min(10_000, n - s)
    codes[s:s + m] = rng.integers(0, 2, size=m) * 3
off = 1
packed = np.zeros((n + off + 3) // 4 + 64, dtype=np.uint8)
sh = np.concatenate([np.zeros(off, dtype=np.uint8), codes[:n]])
for j in range(4):
    c = sh[j::4]; packed[: len(c)] |= (c << (2 * j)).astype(np.uint8)
dev = torch.from_numpy(packed).cuda()
out = torch.zeros(n, dtype=torch.int32, device="cuda")
sk = torch.zeros(n, dtype=torch.int32, device="cuda")
bad, done, t0 = [], 0, time.time()
for w in range(w0, w1 + 1):
    for canonical in (True, False):
        k = 15 if (not canonical or (15 + w - 1) % 2 == 1) else 16
        for mode, use_sk in ((0, True), (1, False), (2, False)):
            try:
                if use_sk:
                    want, want_sk = oracle.run(packed, n, k, w, canonical=canonical, super_kmers=True, base_offset=off)
                else:
                    want, want_sk = oracle.run(packed, n, k, w, canonical=canonical, mode=mode, base_offset=off), None
            except ValueError:
                continue  # (a combination the reference refuses, e.g. open syncmers over an even w)
            b = sm.Builder(k, w, canonical, mode)
            for nb in (0, 5):
                ws.set_blocks_per_lane(nb)
                try:
                    c = b.run_device(dev, n, out, out_sk=sk if use_sk else None, base_offset=off)
                except sm.MinimizerError as e:
                    bad.append((w, canonical, mode, use_sk, nb, str(e)[:60])); done += 1
                    continue
                ok = ws.last_path() == sm.PATH_FUSED and c == len(want) and np.array_equal(out[:c].cpu().numpy().view(np.uint32), want)
                if ok and use_sk:
                    ok = np.array_equal(sk[:c].cpu().numpy().view(np.uint32), want_sk)
                if not ok:
                    bad.append((w, canonical, k, mode, use_sk, nb, c, len(want)))
                done += 1
    ws.set_blocks_per_lane(0)
    if w % 8 == 0:
        print(f"w <= {w}: {done} runs, {len(bad)} bad, {time.time() - t0:.0f} s", flush=True)
print("bad:", bad[:40])
print(f"{done} runs, {len(bad)} mismatches")
