"""Lane-table launches repeated: the 2.5 Gbp long-read batch 60 times (dispatch-order tile ids, the atomic ticket, pinned lane lengths of
3 / 7 / 0 blocks) - the same count, offsets checksum and order-sensitive output checksum every time."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
import numpy as np, torch
import simd_minimizers_amd as sm
ws = sm.default_workspace(0); L = sm.lib(); dev = "cuda:0"
b = sm.canonical_minimizers(21, 11)
rng = np.random.default_rng(61)
n_reads = 200_000
lens = np.exp(rng.uniform(np.log(1000), np.log(50_000), n_reads)).astype(np.int64)
starts = np.zeros(n_reads + 1, dtype=np.int64); starts[1:] = np.cumsum(lens)
total = int(starts[-1])
d = sm.generate_device(total, 7); ds = torch.from_numpy(starts).cuda()
out = torch.empty(int(total * 0.18), dtype=torch.int32, device=dev)
offs = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev); cnt = torch.zeros(1, dtype=torch.int64, device=dev)
idx = torch.arange(1, out.numel() + 1, dtype=torch.int64, device=dev)
ref = None
for it in range(60):
    mode = it % 4
    if mode == 1: os.environ["MM_FORCE_TICKET"] = "1"
    else: os.environ.pop("MM_FORCE_TICKET", None)
    ws.set_blocks_per_lane({2: 3, 3: 7}.get(mode, 0))
    out.fill_(-1); offs.fill_(-1)
    sm._check(L.mm_run_packed_reads_device_async(b.plan().h, ws.h, C.c_void_p(d.data_ptr()), d.numel(), 0, n_reads, C.c_void_p(ds.data_ptr()),
                                                 total, int(lens.max()), C.c_void_p(out.data_ptr()), None, out.numel(), C.c_void_p(offs.data_ptr()), C.c_void_p(cnt.data_ptr())))
    ws.check()
    c = int(cnt.item())
    sig = (c, int(offs.sum().item()), int((out[:c].to(torch.int64) * idx[:c]).sum().item()), int(out[c].item()))
    if ref is None: ref = sig
    assert sig == ref and ws.last_lane_table(), (it, mode, sig, ref)
ws.set_blocks_per_lane(0)
print("60 lane-table runs of", total, "bases,", ref[0], "positions: identical outputs (dispatch order / ticket / 3- and 7-block lanes)")
