"""Per-tile timeline (MM_TRACE) of a batch launch whose contigs end in partial tiles: how long the partial tiles and
their neighbours walk.  Canonical k=31 w=51, 29 blocks per lane, 6 contigs of 200 whole tiles + a fraction of a tile."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
import simd_minimizers_amd as sm
dev = torch.device("cuda:0")
ws = sm.Workspace(0, torch.cuda.current_stream(dev).cuda_stream)
def gen(n, seed):
    t = torch.zeros((n + 3) // 4 + 64, dtype=torch.uint8, device=dev)
    sm._check(sm.lib().mm_generate_device_async(ws.h, seed, 0, n, t.data_ptr()))
    return t
k, w, nblk = 31, 51, 29
tile = 256 * nblk * w
b = sm.canonical_minimizers(k, w).workspace(ws)
ws.set_blocks_per_lane(nblk)
for f in [float(x) for x in os.environ.get('FS', '0,0.02,0.5,0.9').split(',')]:
    lens = [200 * tile + int(tile * f) + (k + w - 2)] * 6
    d = [gen(m, 100 + i) for i, m in enumerate(lens)]
    out = torch.empty(int(sum(lens) * 2 / (w + 1) * 1.2) + 4096, dtype=torch.int32, device=dev)
    for _ in range(3): sm.run_batch_device(b, d, lens, out)
    os.environ["MM_TRACE"] = "/tmp/mm_trace.bin"
    sm.run_batch_device(b, d, lens, out)
    del os.environ["MM_TRACE"]
    t = np.fromfile("/tmp/mm_trace.bin", dtype=np.uint64).reshape(-1, 10)
    t0 = t[:, 0].min()
    start, p1, lb, end = [(t[:, i] - t0).astype(np.float64) / 100.0 for i in range(4)]
    per = 200 + (1 if f > 0 else 0)
    last = np.array([per * (c + 1) - 1 for c in range(6)])
    walk = p1 - start
    normal = np.ones(len(t), bool); normal[last] = False
    prv = last - 1
    print(f"f={f}: span {end.max():.0f} us; ordinary walk {walk[normal].mean():.1f}; last tiles: walk - walk of the tile before "
          f"{np.round(walk[last] - walk[prv], 1)}; phase 2 of the last tiles {np.round(end[last] - p1[last], 1)} (ordinary {np.mean(end[normal] - p1[normal]):.1f})", flush=True)
    del d, out
