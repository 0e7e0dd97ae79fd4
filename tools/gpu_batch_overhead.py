"""Batch launch against single-sequence launch of the same bases (canonical k=31 w=51 and k=21 w=11): the 24 CHM13-like
contigs in one mm_run_batch_device call, ONE sequence of the same total length through the batch entry, and through
mm_run_device."""
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
def timed(step, warm=8, reps=8):
    for _ in range(warm): step()
    torch.cuda.synchronize(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): step()
    torch.cuda.synchronize(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / max(1, l)
lens = list(sharding.CHM13_CONTIG_LENGTHS)
n = sum(lens)
for (k, w) in ((31, 51), (21, 11)):
    b = sm.canonical_minimizers(k, w).workspace(ws)
    out = torch.empty(int(n * 2 / (w + 1) * 1.2) + 4096, dtype=torch.int32, device=dev)
    d = [gen(m, sharding.CHM13_CONTIG_SEED0 + i) for i, m in enumerate(lens)]
    t_batch = timed(lambda: sm.run_batch_device(b, d, lens, out))
    del d
    one = gen(n, 3)
    t_b1 = timed(lambda: sm.run_batch_device(b, [one], [n], out))
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    t_single = timed(lambda: b.run_device(one, n, out, sync=False, d_count=cnt))
    print(f"k={k} w={w}: 24 contigs in one batch {t_batch:.3f} ms ({n / t_batch / 1e6:.0f} Gbases/s) | one sequence through the batch entry "
          f"{t_b1:.3f} ms | one sequence, mm_run_device {t_single:.3f} ms ({n / t_single / 1e6:.0f})", flush=True)
    del one
