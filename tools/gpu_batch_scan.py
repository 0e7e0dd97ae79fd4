"""Batch launch cost against the number of sequences: one 3.1 Gbp sequence cut into n equal contigs, canonical k=31 w=51
and k=21 w=11."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
import simd_minimizers_amd as sm
dev = torch.device("cuda:0")
ws = sm.Workspace(0, torch.cuda.current_stream(dev).cuda_stream)
def gen(n, seed):
    t = torch.zeros((n + 3) // 4 + 64, dtype=torch.uint8, device=dev)
    sm._check(sm.lib().mm_generate_device_async(ws.h, seed, 0, n, t.data_ptr()))
    return t
def timed(step, warm=6, reps=6):
    for _ in range(warm): step()
    torch.cuda.synchronize(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): step()
    torch.cuda.synchronize(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / max(1, l)
N = 3_100_000_000
for (k, w) in ((31, 51), (21, 11)):
    b = sm.canonical_minimizers(k, w).workspace(ws)
    out = torch.empty(int(N * 2 / (w + 1) * 1.2) + 4096, dtype=torch.int32, device=dev)
    for parts in (1, 2, 8, 24, 96):
        m = (N // parts) // 4 * 4
        d = [gen(m, 100 + i) for i in range(parts)]
        lens = [m] * parts
        res = []
        res.append(timed(lambda: sm.run_batch_device(b, d, lens, out)))
        print(f"k={k} w={w}: {parts:3d} contigs of {m} bp: {res[0]:.3f} ms", flush=True)
        del d
