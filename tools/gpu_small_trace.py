"""Timeline of a SHORT run (a shard of a strong split: 387.5 M windows canonical, or BASELINE config 2: 256 Mbp forward)
from the per-tile trace (MM_TRACE, experiments library): how many tiles are resident / walking / in phase 2 in every
10 us bin, when the first tiles start and how the run tails off.  Usage: MM_LIB_PATH=.../libsimd_minimizers_amd_exp.so
python tools/gpu_small_trace.py [n] [fwd|canon] [nblk]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
import simd_minimizers_amd as sm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 387_500_000
canon = (sys.argv[2] if len(sys.argv) > 2 else "canon") == "canon"
nblk = int(sys.argv[3]) if len(sys.argv) > 3 else 0
k, w = (int(os.environ.get("K", 21)), int(os.environ.get("W", 11)))
ws = sm.default_workspace(0)
d = sm.generate_device(n, 3)
out = torch.zeros(int(n * 0.2) + 4096, dtype=torch.int32, device="cuda")
b = sm.Builder(k, w, canon, 0)
ws.set_blocks_per_lane(nblk)
for _ in range(200): b.run_device(d, n, out, sync=False)
ws.sync()
ws.enable_timing(True); ws.kernel_time(True)
for _ in range(50): b.run_device(d, n, out, sync=False)
ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
print(f"n={n} canonical={canon} k={k} w={w} nblk={nblk or 'default'}: kernel {ms / l * 1e3:.1f} us by events")
os.environ["MM_TRACE"] = "/tmp/mm_trace.bin"
b.run_device(d, n, out)
del os.environ["MM_TRACE"]
t = np.fromfile("/tmp/mm_trace.bin", dtype=np.uint64).reshape(-1, 10)
t0 = t[:, 0].min()
start, p1, lb, end = [(t[:, i] - t0).astype(np.float64) / 100.0 for i in range(4)]
bar = (t[:, 5] - t0).astype(np.float64) / 100.0
span = end.max()
print(f"tiles={len(t)} traced span={span:.1f} us; mean walk (wave 0) {np.mean(p1 - start):.1f}, look-back {np.mean(lb - bar):.1f}, copy-out {np.mean(end - lb):.1f}, slot cycle {np.mean(end - start):.1f} us")
print(f"first-round starts: tile 0 at {start[0]:.1f}, tile 255 at {start[min(255, len(t) - 1)]:.1f}, tile 1023 at {start[min(1023, len(t) - 1)]:.1f} us; last tile starts {start[-1]:.1f}, last end {span:.1f}")
bins = np.arange(0, span + 10, 10)
print("  t_us resident walking phase2")
for a in bins[:-1]:
    mid = a + 5
    res = (start <= mid) & (end > mid)
    walk = res & (p1 > mid)
    print(f"  {a:5.0f} {res.sum():5d} {walk.sum():5d} {(res & ~walk).sum():5d}")
# work-weighted: integral of walking tiles over time / (tiles * mean walk) and idle slot time
walk_time = np.sum(p1 - start)
print(f"sum of walk times {walk_time / 1e3:.2f} ms-tiles; span x 1024 slots = {span * 1024 / 1e3:.2f}; walking share {walk_time / (span * 1024):.3f}")
ws.set_blocks_per_lane(0)
