"""Small lane-table batches: whole-call device time of mm_run_packed_reads_device_async for n reads of 10 kbp (the table's four kernels + the
walk are five launches on the stream) against one sequence of the same total length through mm_run_device_async (one launch)."""
import os, sys, statistics, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import simd_minimizers_amd as sm
ws = sm.default_workspace(0); L = sm.lib(); dev = "cuda:0"
b = sm.canonical_minimizers(21, 11)
def med(step, warm=30, reps=15):
    for _ in range(warm): step()
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); step(); e1.record(); torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
    return statistics.median(ms)
for n_reads in (10, 100, 1000, 10_000, 100_000):
    ln = 10_000
    total = n_reads * ln
    starts = torch.arange(0, n_reads + 1, dtype=torch.int64, device=dev) * ln
    d = sm.generate_device(total, 7)
    out = torch.empty(int(total * 0.19) + 4096, dtype=torch.int32, device=dev)
    offs = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev); cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    t_reads = med(lambda: sm._check(L.mm_run_packed_reads_device_async(b.plan().h, ws.h, C.c_void_p(d.data_ptr()), d.numel(), 0, n_reads, C.c_void_p(starts.data_ptr()),
                                                                    total, ln, C.c_void_p(out.data_ptr()), None, out.numel(), C.c_void_p(offs.data_ptr()), C.c_void_p(cnt.data_ptr()))))
    lt = ws.last_lane_table()
    t_seq = med(lambda: b.run_device(d, total, out, sync=False, d_count=cnt))
    print(f"{n_reads:7d} reads x {ln} bp ({total / 1e6:8.1f} Mbp): lane-table call {t_reads * 1e3:8.1f} us (lane_table={lt}) | one sequence of that length {t_seq * 1e3:8.1f} us", flush=True)
