"""Per-tile timeline of the fused kernel (MM_TRACE): are workgroups in lockstep? does phase 2 overlap?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")  # this script flips MM_* switches between runs (mm_env.h)
import simd_minimizers_amd as sm

n = 3_100_000_000
d = sm.generate_device(n, 3)
out = torch.zeros(int(n * 0.2), dtype=torch.int32, device="cuda")
b = sm.Builder(21, 11, os.environ.get("FWD") != "1", 0)
b.run_device(d, n, out)
for spec in sys.argv[1:] or ["0:0", "4:0"]:
    stagger, dbg = spec.split(":")
    os.environ["MM_STAGGER"] = stagger
    os.environ["MM_DEBUG"] = dbg
    os.environ["MM_TRACE"] = "/tmp/mm_trace.bin"
    b.run_device(d, n, out)
    del os.environ["MM_TRACE"]
    t = np.fromfile("/tmp/mm_trace.bin", dtype=np.uint64).reshape(-1, 10)
    t0 = t[:, 0].min()
    start, p1, lb, end = [(t[:, i] - t0).astype(np.float64) / 100.0 for i in range(4)]  # us (100 MHz)
    hw = t[:, 4]
    cu = ((hw >> 32) & 15) * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 8) & 15)
    print(f"== MM_STAGGER={stagger} MM_DEBUG={dbg}: tiles={len(t)} kernel span={end.max():.1f} us")
    bar = (t[:, 5] - t0).astype(np.float64) / 100.0
    print(f"   mean phase1 (wave 0) {np.mean(p1 - start):.1f} us, barrier wait {np.mean(bar - p1):.1f} us, "
          f"look-back {np.mean(lb - bar):.1f} us, copy-out {np.mean(end - lb):.1f} us; p50/p90/p99 look-back "
          f"{np.percentile(lb - bar, 50):.1f}/{np.percentile(lb - bar, 90):.1f}/{np.percentile(lb - bar, 99):.1f}")
    # at sample instants: fraction of resident tiles that are in phase 2
    for ts in np.linspace(0.2, 0.8, 4) * end.max():
        res = (start <= ts) & (end > ts)
        in_p2 = res & (p1 <= ts)
        xc = ((t[:, 4] >> 32) & 15).astype(int)
        print(f"   t={ts:7.1f} us: resident={res.sum():4d} in phase 2={in_p2.sum():4d}; resident by XCD "
              f"{np.bincount(xc[res], minlength=8)} walking by XCD {np.bincount(xc[res & ~in_p2], minlength=8)}")
    # one CU's slots over time
    c0 = cu[0]
    idx = np.flatnonzero(cu == c0)[40:46]
    for i in idx:
        print(f"   CU {c0}: tile {i:6d} start {start[i]:7.1f} p1 {p1[i]:7.1f} lb {lb[i]:7.1f} end {end[i]:7.1f}")
    # who are the stragglers?  per-XCD phase-1 duration and lag of the running maximum
    xcc = ((hw >> 32) & 15).astype(int)
    dur = p1 - start
    print("   per-XCD mean/p90 phase-1 us:", " ".join(f"{x}:{dur[xcc == x].mean():.1f}/{np.percentile(dur[xcc == x], 90):.1f}" for x in range(8)))
    fin = bar  # all waves of the tile done with phase 1
    runmax = np.maximum.accumulate(fin)
    lag = runmax - fin  # how long a tile's aggregate chain is held back by an earlier tile
    print(f"   wait for earlier tiles (running max of phase-1 finish - own finish): mean {lag.mean():.1f} us, p50 {np.percentile(lag, 50):.1f}, p90 {np.percentile(lag, 90):.1f}")
    late = fin - np.concatenate([[0], runmax[:-1]])  # > 0: this tile itself raises the maximum (a straggler)
    st = late > 2.0
    print(f"   stragglers (> 2 us past every earlier tile): {st.sum()} tiles; by XCD {np.bincount(xcc[st], minlength=8)}; "
          f"their mean phase-1 {dur[st].mean():.1f} us vs {dur.mean():.1f}")
    slot_wave = (hw & 15).astype(int)
    print("   phase-1 us by wave slot id:", " ".join(f"{w}:{dur[slot_wave == w].mean():.1f}" for w in range(8) if (slot_wave == w).any()))
