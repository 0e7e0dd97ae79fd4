"""Upper bound of any length-sorting scheme for variable-length short reads: 8 M reads of 100..200 bp,
canonical k=21 w=11, packed back to back - as they come, sorted by length over the WHOLE batch (every wave homogeneous), and
sorted within groups of 256 / 1024 reads (what a tile-level sort could see)."""
import os, sys, statistics, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
import simd_minimizers_amd as sm
ws = sm.default_workspace(0); L = sm.lib(); dev = "cuda:0"
b = sm.canonical_minimizers(21, 11)
g = torch.Generator(device=dev); g.manual_seed(6)
n_reads = 8_000_000
base = torch.randint(100, 201, (n_reads,), device=dev, generator=g)
def run(lens, label):
    starts = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev); starts[1:] = torch.cumsum(lens, 0)
    n = int(starts[-1].item())
    d = sm.generate_device(n, 7)
    out = torch.empty(int(n * 0.19) + 4096, dtype=torch.int32, device=dev)
    offs = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev); cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    def step():
        sm._check(L.mm_run_packed_reads_device_async(b.plan().h, ws.h, C.c_void_p(d.data_ptr()), d.numel(), 0, n_reads, C.c_void_p(starts.data_ptr()),
                                                     n, 200, C.c_void_p(out.data_ptr()), None, out.numel(), C.c_void_p(offs.data_ptr()), C.c_void_p(cnt.data_ptr())))
    for _ in range(60): step()
    torch.cuda.synchronize()
    ms = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); step(); e1.record(); torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
    m = statistics.median(ms)
    print(f"{label:46s} {m:.4f} ms = {n / m / 1e6:.0f} Gbases/s", flush=True)
for pol in (None, "1"):
    if pol: os.environ["MM_LANE_TABLE"] = pol
    print("MM_LANE_TABLE =", pol, "(one lane per read, every tile walks the longest read's blocks)" if not pol else "(lane table: a wave walks its own longest lane's blocks)", flush=True)
    run(base, "as they come")
    run(torch.sort(base)[0], "sorted over the whole batch")
    for grp in (256, 1000, 4000):
        run(torch.sort(base.view(-1, grp), dim=1)[0].reshape(-1), f"sorted within groups of {grp} reads")
    run(torch.full_like(base, 150), "all 150 bp (packed-starts entry)")
