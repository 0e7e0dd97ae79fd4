import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
import simd_minimizers_amd as sm
n = 1_000_000_000
d = sm.generate_device(n, 2); ws = sm.default_workspace(0)
amb = torch.zeros((n + 7) // 8 + 64, dtype=torch.uint8, device="cuda")
g = torch.Generator(device="cuda"); g.manual_seed(1)
amb[torch.randint(0, n // 8, (n // 8000,), device="cuda", generator=g)] = 1 << 3
for s in torch.randint(0, n // 8 - 7000, (200,), generator=g, device="cuda").tolist():
    amb[s:s + 6250] = 0xFF
zero = torch.zeros_like(amb)
out = torch.zeros(int(n * 0.2), dtype=torch.int32, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
def kt(step, warm=5, reps=6):
    for _ in range(warm): step()
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): step()
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / max(1, l)
for (k, w) in ((21, 11), (31, 51), (31, 33)):
    b = sm.canonical_minimizers(k, w)
    print(f"k={k} w={w}: plain kernel {kt(lambda: b.run_device(d, n, out, sync=False, d_count=cnt)):.3f} ms | skip kernel, Ns as in the bench "
          f"{kt(lambda: b.run_skip_ambiguous_device(d, amb, n, out, sync=False, d_count=cnt)):.3f} | skip kernel, no N at all "
          f"{kt(lambda: b.run_skip_ambiguous_device(d, zero, n, out, sync=False, d_count=cnt)):.3f}", flush=True)
# a genome-like pattern: 200 gaps of 50 kbp and nothing between them (round 4: clean waves take the plain walk)
gaps = torch.zeros_like(amb)
for s in torch.randint(0, n // 8 - 7000, (200,), generator=g, device="cuda").tolist():
    gaps[s:s + 6250] = 0xFF
for (k, w) in ((21, 11), (31, 51), (31, 33), (15, 17)):
    b = sm.canonical_minimizers(k, w)
    print(f"k={k} w={w}: skip kernel, 200 gaps of 50 kbp only {kt(lambda: b.run_skip_ambiguous_device(d, gaps, n, out, sync=False, d_count=cnt)):.3f} ms", flush=True)
