"""Cost of many small sequences with one plan (mm_run_batch_device): one launch per sequence today."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import simd_minimizers_amd as sm
b = sm.canonical_minimizers(21, 11)
for n_seq, length in [(24, 130_000_000), (2000, 500_000), (20000, 10_000), (20000, 1000)]:
    total = n_seq * length
    big = sm.generate_device(total + 64, 5)
    stride_bytes = length // 4
    d = [big[i * stride_bytes: i * stride_bytes + stride_bytes + 16] for i in range(n_seq)]
    lens = [length] * n_seq
    out = torch.zeros(int(total * 0.2) + n_seq * 4, dtype=torch.int32, device="cuda")
    sm.run_batch_device(b, d, lens, out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    offs = sm.run_batch_device(b, d, lens, out)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{n_seq} sequences x {length} bp: {dt*1e3:.2f} ms, {total/dt/1e9:.1f} Gbases/s, {dt/n_seq*1e6:.1f} us per sequence", flush=True)
    del big, d, out
