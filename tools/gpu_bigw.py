import os, sys
sys.path.insert(0, "/root/repo")
import torch
import simd_minimizers_amd as sm
n = 1_000_000_000
d = sm.generate_device(n, 2); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.45) + 1024, dtype=torch.int32, device="cuda")
def t(b, warm=15, reps=15):
    for _ in range(warm): b.run_device(d, n, out, sync=False)
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, sync=False)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l
for k, w, canon, mode, nblks in [(31, 51, True, 0, [0, 16, 24, 32, 40]), (19, 19, True, 0, [0, 16, 20, 24, 28]), (21, 31, True, 0, [0, 16, 20, 25]),
                                 (15, 17, True, 1, [0, 14, 18, 22]), (21, 31, False, 0, [0, 16, 24]), (31, 51, False, 0, [0, 16, 24])]:
    b = sm.Builder(k, w, canon, mode)
    for nblk in nblks:
        ws.set_blocks_per_lane(nblk)
        print(f"k={k} w={w} canon={canon} mode={mode} nblk={nblk or 'default'}: {t(b):.3f} ms", flush=True)
ws.set_blocks_per_lane(0)
