"""Is a step of the hot path capturable in a HIP graph (memsets + kernel + count copy on one stream)?
Times 20 replays against 20 plain launches of the bench workload."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import simd_minimizers_amd as sm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_100_000_000
dev = torch.device("cuda:0")
side = torch.cuda.Stream(dev)
with torch.cuda.stream(side):
    ws = sm.Workspace(0, side.cuda_stream)
    b = sm.canonical_minimizers(21, 11).workspace(ws)
    d = torch.zeros((n + 3) // 4 + 64, dtype=torch.uint8, device=dev)
    sm._check(sm.lib().mm_generate_device_async(ws.h, 3, 0, n, d.data_ptr()))
    out = torch.empty(int(n * 2.3 / 12) + 4096, dtype=torch.int32, device=dev)
    d_count = torch.zeros(1, dtype=torch.int64, device=dev)
    for _ in range(20):
        b.run_device(d, n, out, sync=False, d_count=d_count)
    side.synchronize()
    want = int(d_count.item())
    t0 = time.perf_counter()
    for _ in range(20):
        b.run_device(d, n, out, sync=False, d_count=d_count)
    side.synchronize()
    plain = (time.perf_counter() - t0) / 20 * 1e3
    g = torch.cuda.CUDAGraph()
    d_count.zero_()
    side.synchronize()
    with torch.cuda.graph(g, stream=side):
        b.run_device(d, n, out, sync=False, d_count=d_count)
    for _ in range(5):
        g.replay()
    side.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    side.synchronize()
    graph = (time.perf_counter() - t0) / 20 * 1e3
    got = int(d_count.item())
print(f"n={n}: plain {plain:.4f} ms/step, graph replay {graph:.4f} ms/step, count {got} == {want}: {got == want}")
