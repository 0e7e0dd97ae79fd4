"""One-off parity sweep of the LANE-TABLE launch (round 6) over every window size 1 .. 128: canonical and forward minimizers, reads at
a fixed stride with per-read lengths (empty, shorter than a window, full), the table forced (MM_LANE_TABLE=1) with lanes of THREE
blocks - a read of about 150 windows takes up to 50 lanes at w = 1 and one at w = 51 and above, so every seam rule and every walk
flavour (two-body, range-checked, run-time specialised) is met - against the oracle run on every read.
usage: gpu_all_w_lanes_parity.py [w_from] [w_to]; MM_SWEEP_MODES=1: closed / open syncmers and super-k-mer indices on a subset of
the window sizes instead (their reads-mode kernels are specialised at first use)"""
import os, sys, time
os.environ["MM_LANE_TABLE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import mm_oracle as oracle
import simd_minimizers_amd as sm
w0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w1 = int(sys.argv[2]) if len(sys.argv) > 2 else 128
rng = np.random.default_rng(131)
ws = sm.default_workspace(0)
ws.set_blocks_per_lane(3)
n_reads = 700
bad, done, t0 = [], 0, time.time()
MODES = os.environ.get("MM_SWEEP_MODES") == "1"
WS = [w for w in (list(range(1, 17)) + [19, 25, 31, 33, 41, 51, 64, 100, 128]) if w0 <= w <= w1] if MODES else range(w0, w1 + 1)
for w in WS:
    for (canonical, mode, sk) in ([(True, 1, False), (False, 2, False), (True, 0, True), (False, 0, True)] if MODES else [(True, 0, False), (False, 0, False)]):
        if mode == 2 and w % 2 == 0:
            continue
        k = 15 if (not canonical or (15 + w - 1) % 2 == 1) else 16
        l = k + w - 1
        read_len = l + 150 + (w % 7)
        stride = read_len + 1 + (w % 3)
        off = w % 4
        span = n_reads * stride + 64 + off
        codes = rng.integers(0, 4, size=span).astype(np.uint8)
        codes[span // 3: span // 3 + 20_000] = rng.integers(0, 2, size=20_000) * 3      # tie-heavy reads
        packed = np.zeros((span + 3) // 4 + 64, dtype=np.uint8)
        for j in range(4):
            c = codes[j::4]; packed[: len(c)] |= (c << (2 * j)).astype(np.uint8)
        d_p = torch.from_numpy(packed).cuda()
        lens = rng.integers(0, read_len + 1, size=n_reads)
        lens[::9] = read_len; lens[1::50] = 0; lens[2::50] = l - 1; lens[3::50] = l
        d_lens = torch.from_numpy(lens.astype(np.int32)).cuda()
        out = torch.zeros(n_reads * read_len, dtype=torch.int32, device="cuda")
        offs = torch.zeros(n_reads + 1, dtype=torch.int64, device="cuda")
        b = sm.Builder(k, w, canonical, mode)
        osk = torch.zeros_like(out) if sk else None
        total = sm.run_reads_device(b, d_p, n_reads, stride, read_len, out, offs, read_lens=d_lens, base_offset=off, out_sk=osk)
        hsk = osk[:total].cpu().numpy().view(np.uint32) if sk else None
        fused = ws.last_path() == sm.PATH_FUSED and ws.last_lane_table()
        ho, hp = offs.cpu().numpy(), out[:total].cpu().numpy().view(np.uint32)
        ok = fused and ho[-1] == total
        for r in range(n_reads):
            if not ok: break
            want = oracle.run(packed, int(lens[r]), k, w, canonical=canonical, mode=mode, base_offset=off + r * stride, super_kmers=sk)
            if sk:
                want, wsk = want
                if not np.array_equal(hsk[ho[r]:ho[r + 1]], wsk):
                    ok = False
                    bad.append((w, canonical, k, r, int(lens[r]), "sk"))
            if not np.array_equal(hp[ho[r]:ho[r + 1]], want):
                ok = False
                bad.append((w, canonical, k, r, int(lens[r]), fused))
        if not ok and (not bad or bad[-1][0] != w): bad.append((w, canonical, k, -1, -1, fused))
        done += 1
    if w % 8 == 0 or MODES:
        print(f"w <= {w}: {done} batches of {n_reads} reads, {len(bad)} bad, {time.time() - t0:.0f} s", flush=True)
print("bad:", bad[:40])
print(f"{done} batches, {len(bad)} mismatches")
