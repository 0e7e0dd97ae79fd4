"""Wide sequence loads (16 bytes per lane and group of W-blocks, MM_WIDE_LOADS=1) against the 8-byte loads per block
(=0): both through the run-time specialisation, same box, 3.1 Gbp; outputs must be identical."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")  # this script flips MM_* switches between runs (mm_env.h)
import simd_minimizers_amd as sm
n = int(os.environ.get("MM_N", "3100000000"))
d = sm.generate_device(n, 3); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.26) + 1024, dtype=torch.int32, device="cuda")
os.environ["MM_JIT_FORCE"] = "1"
def t(b, warm=10, reps=10):
    for _ in range(warm): b.run_device(d, n, out, sync=False)
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, sync=False)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l
def chk(b):
    out.zero_()
    c = b.run_device(d, n, out)
    v = out[:c].to(torch.int64)
    return c, int((v * torch.arange(1, c + 1, device="cuda")).sum().item())
cfgs = [(21, 11, False, 0), (21, 11, True, 0), (31, 51, True, 0), (15, 17, True, 1), (21, 25, True, 0), (21, 7, False, 0), (31, 33, True, 0)]
if len(sys.argv) > 1: cfgs = [cfgs[int(x)] for x in sys.argv[1].split(",")]
for (k, w, canon, mode) in cfgs:
    b = sm.Builder(k, w, canon, mode)
    res = {}
    for wide in ("0", "1"):
        os.environ["MM_JIT_DEFS"] = f"-DMM_WIDE_LOADS={wide}"
        os.environ["MM_DEBUG"] = "0"
        res[wide] = (chk(b), t(b))
        os.environ["MM_DEBUG"] = "3"
        res[wide] += (t(b),)
    os.environ["MM_DEBUG"] = "0"
    same = res["0"][0] == res["1"][0]
    print(f"k={k} w={w} canon={canon} mode={mode}: 8-byte loads {res['0'][1]:.3f} ms (walk {res['0'][2]:.3f}) | wide loads {res['1'][1]:.3f} ms "
          f"(walk {res['1'][2]:.3f}) | {n / res['1'][1] / 1e6:.0f} Gbases/s | outputs {'SAME' if same else 'DIFFERENT'} {res['1'][0][0]}", flush=True)
