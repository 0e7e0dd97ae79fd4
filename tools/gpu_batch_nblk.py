"""The 24 CHM13-like contigs in one batch launch, canonical k=31 w=51: kernel time over pinned blocks per lane against
the whole-rounds tuner's choice (0 = tuned)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
import simd_minimizers_amd as sm
from simd_minimizers_amd import sharding
dev = torch.device("cuda:0")
ws = sm.Workspace(0, torch.cuda.current_stream(dev).cuda_stream)
def gen(n, seed):
    t = torch.zeros((n + 3) // 4 + 64, dtype=torch.uint8, device=dev)
    sm._check(sm.lib().mm_generate_device_async(ws.h, seed, 0, n, t.data_ptr()))
    return t
def timed(step, warm=6, reps=8):
    for _ in range(warm): step()
    torch.cuda.synchronize(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): step()
    torch.cuda.synchronize(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / max(1, l)
lens = list(sharding.CHM13_CONTIG_LENGTHS)
for (k, w, cand) in ((31, 51, (0, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33)), (21, 11, (0, 28, 30, 32, 34, 36))):
    b = sm.canonical_minimizers(k, w).workspace(ws)
    d = [gen(m, 100 + i) for i, m in enumerate(lens)]
    out = torch.empty(int(sum(lens) * 2 / (w + 1) * 1.2) + 4096, dtype=torch.int32, device=dev)
    for nblk in cand:
        ws.set_blocks_per_lane(nblk)
        print(f"k={k} w={w} blocks per lane {nblk or 'tuned'}: {timed(lambda: sm.run_batch_device(b, d, lens, out)):.3f} ms", flush=True)
    ws.set_blocks_per_lane(0)
    del d, out
