import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
import numpy as np, torch
import mm_oracle as oracle
import simd_minimizers_amd as sm
rng = np.random.default_rng(55)
ws = sm.default_workspace(0)
for (k, w) in ((1, 55), (1, 33), (3, 51), (3, 41), (19, 33), (21, 35), (31, 51), (21, 63), (19, 65), (21, 81), (31, 33)):
    for n in (333, 5_003, 120_007, 1_500_013):
        a = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n + 8)].copy()
        if k <= 3 and n > 1000:
            a[n // 3: n // 3 + 900] = ord("A")
        a[rng.integers(0, n, size=max(1, n // 250))] = ord("N")
        s0 = int(rng.integers(0, n))
        a[s0:s0 + int(rng.integers(1, 200))] = ord("N")
        if (k, w, n) != (19, 65, 1_500_013): continue
        packed, amb = oracle.pack_ascii_n(a.tobytes())
        d_p, d_m = torch.from_numpy(packed).cuda(), torch.from_numpy(amb).cuda()
        out = torch.zeros(n + 8, dtype=torch.int32, device="cuda")
        want = oracle.run_skip_ambiguous(packed, amb, n, k, w)
        for nb in (0, 4, 6, 3):
            ws.set_blocks_per_lane(nb)
            c = sm.canonical_minimizers(k, w).run_skip_ambiguous_device(d_p, d_m, n, out)
            got = out[:c].cpu().numpy().view(np.uint32)
            m = min(len(got), len(want))
            diff = np.flatnonzero(got[:m] != want[:m])
            l = k + w - 1
            msg = "ok" if (len(got) == len(want) and not len(diff)) else f"{len(got)} vs {len(want)}"
            print(f"blocks per lane {nb or 'default'}: {msg}")
            if len(diff):
                i = int(diff[0])
                p = int(want[i])
                ns = np.flatnonzero(a[max(0, p - 3 * l): p + 3 * l] == ord("N")) + max(0, p - 3 * l)
                print(f"   first difference at output {i}: got {got[max(0,i-2):i+3]} want {want[max(0,i-2):i+3]}; Ns near {ns.tolist()}; p={p} p/w={p/w:.2f} p mod (256*w)={p % (256*w)}")
                # second mismatch region
                j = i
                g2, w2 = list(got), list(want)
        ws.set_blocks_per_lane(0)
        ws.force_generic(True)
        c = sm.canonical_minimizers(k, w).run_skip_ambiguous_device(d_p, d_m, n, out)
        got = out[:c].cpu().numpy().view(np.uint32)
        print("generic family:", "ok" if len(got) == len(want) and np.array_equal(got, want) else f"{len(got)} vs {len(want)}")
        ws.force_generic(False)
        # the same region through other window sizes / k (same l): which part is sensitive?
        for (k2, w2) in ((21, 63), (17, 67), (19, 65), (23, 61), (35, 49), (33, 51)):
            want2 = oracle.run_skip_ambiguous(packed, amb, n, k2, w2)
            c = sm.canonical_minimizers(k2, w2).run_skip_ambiguous_device(d_p, d_m, n, out)
            got = out[:c].cpu().numpy().view(np.uint32)
            print(f"k={k2} w={w2}:", "ok" if len(got) == len(want2) and np.array_equal(got, want2) else f"{len(got)} vs {len(want2)}")
        # without ambiguity bits at all (plain canonical run on the same codes)
        want3 = oracle.run(packed, n, k, w, canonical=True)
        c = sm.canonical_minimizers(k, w).run_device(d_p, n, out)
        got = out[:c].cpu().numpy().view(np.uint32)
        print("plain run on the same codes:", "ok" if len(got) == len(want3) and np.array_equal(got, want3) else f"{len(got)} vs {len(want3)}")
        zero = torch.zeros_like(d_m)
        c = sm.canonical_minimizers(k, w).run_skip_ambiguous_device(d_p, zero, n, out)
        got = out[:c].cpu().numpy().view(np.uint32)
        print("skip entry, no ambiguity bit set:", "ok" if len(got) == len(want3) and np.array_equal(got, want3) else f"{len(got)} vs {len(want3)}")
