"""Throughput against input size (forward and canonical k=21 w=11; optionally k=31 w=51): kernel time by HIP events and
whole asynchronous call (wall clock over many calls), default lanes and a sweep of blocks per lane.  The regime every
rank of a strong split lives in (387 Mbp per GPU at N = 8), and BASELINE config 2 (256 Mbp)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import simd_minimizers_amd as sm
sizes = [int(x) for x in os.environ.get("SIZES", "33554432,67108864,134217728,268435456,387500000,536870912,1073741824,3100000000").split(",")]
plans = [(21, 11, False), (21, 11, True)] + ([(31, 51, True)] if os.environ.get("BIGW") else [])
nblks = [int(x) for x in os.environ.get("NBLKS", "0").split(",")]
ws = sm.default_workspace(0)
nmax = max(sizes)
d = sm.generate_device(nmax, 3)
out = torch.zeros(int(nmax * 0.2) + 4096, dtype=torch.int32, device="cuda")
d_count = torch.zeros(1, dtype=torch.int64, device="cuda")
for (k, w, canon) in plans:
    b = sm.Builder(k, w, canon, 0)
    for n in sizes:
        for nblk in nblks:
            ws.set_blocks_per_lane(nblk)
            reps = max(10, min(200, int(2e10 / n)))
            for _ in range(max(10, reps // 2)): b.run_device(d, n, out, sync=False, d_count=d_count)
            ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
            t0 = time.perf_counter()
            for _ in range(reps): b.run_device(d, n, out, sync=False, d_count=d_count)
            ws.sync(); wall = (time.perf_counter() - t0) / reps * 1e3
            ms, l = ws.kernel_time(True); ws.enable_timing(False)
            # the same without event records between the calls
            t0 = time.perf_counter()
            for _ in range(reps): b.run_device(d, n, out, sync=False, d_count=d_count)
            ws.sync(); wall2 = (time.perf_counter() - t0) / reps * 1e3
            # synchronous calls (count read back every time)
            t0 = time.perf_counter()
            for _ in range(reps): b.run_device(d, n, out)
            wall3 = (time.perf_counter() - t0) / reps * 1e3
            print(f"k={k} w={w} canonical={canon} n={n} nblk={nblk or 'default'}: kernel {ms / l:.4f} ms ({n / (ms / l) / 1e6:.0f} Gbases/s), "
                  f"async call {wall2:.4f} ms ({n / wall2 / 1e6:.0f}), with events {wall:.4f}, sync call {wall3:.4f} ms ({n / wall3 / 1e6:.0f})", flush=True)
ws.set_blocks_per_lane(0)
