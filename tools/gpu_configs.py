"""Steady-state kernel time (HIP events, 20 untimed + 20 timed launches) of the SURVEY configs and a
few others on 1 Gbp; the default lane length and a couple of alternatives."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import simd_minimizers_amd as sm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
d = sm.generate_device(n, 2); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.45) + 1024, dtype=torch.int32, device="cuda")
sk = torch.zeros_like(out)
def t(b, use_sk=False, warm=20, reps=20):
    for _ in range(warm): b.run_device(d, n, out, out_sk=sk if use_sk else None, sync=False)
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, out_sk=sk if use_sk else None, sync=False)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l
for k, w, canon, mode, use_sk, nblks in [(21, 11, True, 0, False, [0, 20, 28]), (21, 11, False, 0, False, [0, 32]),
                                         (31, 51, True, 0, False, [0, 6, 12]), (15, 17, True, 1, False, [0, 12, 20]),
                                         (15, 17, True, 2, False, [0]), (5, 7, False, 0, False, [0]),
                                         (31, 5, True, 0, False, [0]), (19, 19, True, 0, False, [0]),
                                         (21, 11, True, 0, True, [0]), (15, 10, False, 0, False, [0])]:
    b = sm.Builder(k, w, canon, mode)
    for nblk in nblks:
        ws.set_blocks_per_lane(nblk)
        ms = t(b, use_sk)
        print(f"k={k} w={w} canonical={canon} mode={mode} sk={use_sk} nblk={nblk or 'default'}: {ms:.3f} ms  "
              f"{n / ms / 1e6:.0f} Gbases/s", flush=True)
    ws.set_blocks_per_lane(0)
