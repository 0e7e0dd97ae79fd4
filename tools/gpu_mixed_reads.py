"""A MIXED batch: 4 M reads of 150 bp and 40 k reads of 15 kbp shuffled into one packed buffer (0.6 + 0.6 Gbp), canonical k=21 w=11, one
call of mm_run_packed_reads_device - against the two halves run on their own (one lane per read; lane table): what the short reads lose
by taking a lane each inside a lane-table launch."""
import os, sys, statistics, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import simd_minimizers_amd as sm
ws = sm.default_workspace(0); L = sm.lib(); dev = "cuda:0"
b = sm.canonical_minimizers(21, 11)
g = torch.Generator(device=dev); g.manual_seed(4)
def run(lens, label):
    n_reads = lens.numel()
    starts = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev); starts[1:] = torch.cumsum(lens, 0)
    n = int(starts[-1].item()); mx = int(lens.max().item())
    d = sm.generate_device(n, 7)
    out = torch.empty(int(n * 0.19) + 4096, dtype=torch.int32, device=dev)
    offs = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev); cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    def step():
        sm._check(L.mm_run_packed_reads_device_async(b.plan().h, ws.h, C.c_void_p(d.data_ptr()), d.numel(), 0, n_reads, C.c_void_p(starts.data_ptr()),
                                                     n, mx, C.c_void_p(out.data_ptr()), None, out.numel(), C.c_void_p(offs.data_ptr()), C.c_void_p(cnt.data_ptr())))
    for _ in range(60): step()
    torch.cuda.synchronize()
    ms = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); step(); e1.record(); torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
    m = statistics.median(ms)
    print(f"{label:52s} {n / 1e9:.2f} Gbp  {m:.4f} ms = {n / m / 1e6:.0f} Gbases/s  lane_table={ws.last_lane_table()}", flush=True)
    return m
short = torch.full((4_000_000,), 150, dtype=torch.int64, device=dev)
long_ = torch.full((40_000,), 15_000, dtype=torch.int64, device=dev)
mixed = torch.cat([short, long_])[torch.randperm(4_040_000, device=dev, generator=g)]
a = run(short, "4 M x 150 bp alone")
c = run(long_, "40 k x 15 kbp alone")
m = run(mixed, "the two shuffled into one batch")
print(f"sum of the halves {a + c:.4f} ms; the mixed batch costs {m / (a + c):.3f} x that", flush=True)
