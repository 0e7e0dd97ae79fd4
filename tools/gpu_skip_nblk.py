"""Skip-ambiguous runs with scattered Ns (every wave dirty): kernel time against the lane length (blocks per lane).  The
dirty walk streams a third array - one bit per window - beside the two sequence streams; profiles/r05_skip_dirty_walk.txt
shows its L2 misses at 3.2 x the plain walk's for k=31 w=33: do shorter lanes (smaller resident spans) pay for it?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
import simd_minimizers_amd as sm
n = 1_000_000_000
d = sm.generate_device(n, 2); ws = sm.default_workspace(0)
amb = torch.zeros((n + 7) // 8 + 64, dtype=torch.uint8, device="cuda")
g = torch.Generator(device="cuda"); g.manual_seed(1)
amb[torch.randint(0, n // 8, (n // 8000,), device="cuda", generator=g)] = 1 << 3
out = torch.zeros(int(n * 0.2), dtype=torch.int32, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
def kt(step, reps=6):
    import time
    t0 = time.perf_counter()  # (warm up by time: a few steps after an idle second read up to 10 % slow)
    while time.perf_counter() - t0 < 0.1:
        step(); ws.sync()
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): step()
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / max(1, l)
CASES = ((31, 33, (0, 13, 11, 9, 8, 7, 6, 5)), (31, 51, (0, 27, 20, 16, 12, 10, 8, 6)), (21, 25, (0, 14, 11, 9, 7)), (21, 19, (0, 20, 16, 12, 9)), (15, 17, (0, 28, 20, 14, 10)))
if len(sys.argv) > 1:  # k,w,nb,nb,... per argument (0 = the library's default lanes)
    CASES = tuple((int(a.split(",")[0]), int(a.split(",")[1]), tuple(int(x) for x in a.split(",")[2:])) for a in sys.argv[1:])
for (k, w, sweep) in CASES:
    b = sm.canonical_minimizers(k, w)
    row = []
    for nb in sweep:
        ws.set_blocks_per_lane(nb)
        row.append(f"{nb or 'default'}: {kt(lambda: b.run_skip_ambiguous_device(d, amb, n, out, sync=False, d_count=cnt)):.3f}")
    ws.set_blocks_per_lane(0)
    plain = kt(lambda: b.run_device(d, n, out, sync=False, d_count=cnt))
    print(f"k={k} w={w}: plain {plain:.3f} ms | dirty walk by blocks per lane: " + "  ".join(row), flush=True)
