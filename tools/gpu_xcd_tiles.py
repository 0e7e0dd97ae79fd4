"""Tiles sized by XCD (MM_XCD_W): per-XCD walk times from a traced launch -> relative speeds -> kernel time with
proportional tiles against uniform tiles, same box, same process; outputs compared.  usage: gpu_xcd_tiles.py [fwd|can|c4]..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import simd_minimizers_amd as sm

n = 3_100_000_000
d = sm.generate_device(n, 3); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")

def timed(b, warm=8, reps=10):
    for _ in range(warm): b.run_device(d, n, out, sync=False, d_count=cnt)
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, sync=False, d_count=cnt)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l

def traced(b):
    os.environ["MM_TRACE"] = "/tmp/mm_trace.bin"
    b.run_device(d, n, out)
    del os.environ["MM_TRACE"]
    t = np.fromfile("/tmp/mm_trace.bin", dtype=np.uint64).reshape(-1, 10)
    ok = t[:, 1] > 0
    t = t[ok]
    xcc = ((t[:, 4] >> 32) & 15).astype(int)
    dur = (t[:, 1] - t[:, 0]).astype(np.float64) / 100.0
    bar = t[:, 5].astype(np.float64) / 100.0
    lag = np.maximum.accumulate(bar) - bar
    bid = np.flatnonzero(ok)
    c = (xcc - bid) & 7
    lb = (t[:, 2] - t[:, 5]).astype(np.float64) / 100.0  # look-back: all waves done -> exclusive prefix known
    traced.lb = np.array([lb[xcc == x].mean() for x in range(8)])
    return np.array([dur[xcc == x].mean() for x in range(8)]), lag.mean(), np.bincount(c, minlength=8), len(t)

def digest():
    m = int(cnt.item())
    v = out[:m].to(torch.int64)
    return m, int(v.sum().item()), int((v * torch.arange(1, m + 1, device="cuda", dtype=torch.int64) % 1000003).sum().item())

for which in sys.argv[1:] or ["can", "fwd"]:
    k, w, canon = {"can": (21, 11, True), "fwd": (21, 11, False), "c4": (31, 51, True)}[which]
    b = sm.Builder(k, w, canon, 0)
    os.environ.pop("MM_XCD_W", None)
    timed(b)
    t_uni = timed(b)
    ref = digest()
    dur, lag, cs, nt = traced(b)
    print(f"== {which}: uniform {t_uni:.3f} ms; tiles {nt}; per-XCD walk us {np.round(dur, 1)}; max/mean {dur.max() / dur.mean():.3f}; "
          f"wait for earlier tiles {lag:.1f} us; c histogram {cs}", flush=True)
    ALPHA = float(os.environ.get("ALPHA", "0.5"))
    wts = np.ones(8)
    lbw = traced.lb
    for it in range(int(os.environ.get("ITERS", "6"))):
        # controller: an XCD whose tiles wait longer than average in the look-back is ahead - give it more
        wts = wts * (1.0 + ALPHA * (lbw - lbw.mean()) / dur.mean())
        wts = np.clip(wts / wts.max(), 0.8, 1.0)
        os.environ["MM_XCD_W"] = ",".join(f"{x:.4f}" for x in wts)
        t_x = timed(b)
        got = digest()
        dur2, lag2, cs2, nt2 = traced(b)
        # same protocol for both (a traced launch leaves the clocks low): A B A B
        ab = []
        for rep in range(2):
            os.environ["MM_XCD_W"] = ",".join(f"{x:.4f}" for x in wts)
            ab.append(timed(b, warm=12))
            os.environ.pop("MM_XCD_W")
            ab.append(timed(b, warm=12))
        t_x, t_u2 = min(ab[0], ab[2]), min(ab[1], ab[3])
        print("            A/B ms (tiles by XCD, uniform) x 2:", " ".join(f"{x:.3f}" for x in ab), flush=True)
        print(f"   iter {it}: weights {np.round(wts, 3)} -> {t_x:.3f} ms (uniform again {t_u2:.3f}); tiles {nt2}; per-XCD walk us "
              f"{np.round(dur2, 1)} max/mean {dur2.max() / dur2.mean():.3f}; wait {lag2:.1f} us; c {cs2}; outputs "
              f"{'identical' if got == ref else 'DIFFER ' + str(got) + ' ' + str(ref)}", flush=True)
        lbw = traced.lb
        print(f"            look-back us by XCD {np.round(lbw, 1)}", flush=True)
