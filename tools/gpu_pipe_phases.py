"""Where the pipelined kernel's time goes: MM_PIPE_DEBUG 0 full, 1 no look-back loads, 2 no copy-out, 3 both."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import simd_minimizers_amd as sm
n = 3_100_000_000
d = sm.generate_device(n, 3); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
def t(b, warm=10, reps=10):
    for _ in range(warm): b.run_device(d, n, out, sync=False)
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, sync=False)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l
for (k, w, canon) in [(21, 11, True), (21, 11, False)]:
    b = sm.Builder(k, w, canon, 0)
    for nb in [int(x) for x in os.environ.get("NBLKS", "0").split(",")]:
        ws.set_blocks_per_lane(nb)
        res = []
        for dbg in [0, 1, 2, 3]:
            os.environ["MM_PIPE_DEBUG"] = str(dbg)
            res.append(t(b))
        os.environ["MM_PIPE_DEBUG"] = "0"
        print(f"k={k} w={w} canon={canon} nblk={nb}: full {res[0]:.3f}  no-lookback {res[1]:.3f}  no-copy {res[2]:.3f}  neither {res[3]:.3f}", flush=True)
ws.set_blocks_per_lane(0)
