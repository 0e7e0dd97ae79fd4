import os, sys
sys.path.insert(0, "/root/repo")
import torch
import simd_minimizers_amd as sm
n = 1_000_000_000
d = sm.generate_device(n, 2); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda"); sk = torch.zeros_like(out)
def t(b, warm=20, reps=20):
    for _ in range(warm): b.run_device(d, n, out, out_sk=sk, sync=False)
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, out_sk=sk, sync=False)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l
for k, w, canon in [(21, 11, True), (21, 11, False), (31, 51, True)]:
    b = sm.Builder(k, w, canon, 0)
    for nblk in ([0, 6, 8, 10, 12, 16, 20] if w == 11 else [0, 3, 4, 6, 8]):
        ws.set_blocks_per_lane(nblk)
        print(f"k={k} w={w} canon={canon} SK nblk={nblk or 'default'}: {t(b):.3f} ms", flush=True)
ws.set_blocks_per_lane(0)
