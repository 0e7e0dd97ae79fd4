"""A/B of two builds of the fused kernel on one box: tools/gpu_defs_ab.py "<defs A>" "<defs B>" [config indices].
Both go through the run-time specialisation (MM_JIT_FORCE=1, MM_JIT_DEFS=<defs>), 3.1 Gbp; prints kernel time of the
whole kernel and of the walk alone (MM_DEBUG=3) and checks that the outputs are identical."""
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
cfgs = [(21, 11, False, 0), (21, 11, True, 0), (31, 51, True, 0), (15, 17, True, 1), (21, 25, True, 0), (21, 7, False, 0),
        (31, 33, True, 0), (20, 12, True, 0), (21, 20, False, 0), (15, 17, True, 2), (31, 41, True, 0), (21, 37, True, 0), (31, 35, True, 0), (20, 36, True, 0)]
A, B = sys.argv[1], sys.argv[2]
if len(sys.argv) > 3: cfgs = [cfgs[int(x)] for x in sys.argv[3].split(",")]
print(f"A = '{A}'   B = '{B}'")
for (k, w, canon, mode) in cfgs:
    b = sm.Builder(k, w, canon, mode)
    res = []
    for defs in (A, B, A, B):
        os.environ["MM_JIT_DEFS"] = defs
        os.environ["MM_DEBUG"] = "0"
        r = (chk(b), t(b))
        os.environ["MM_DEBUG"] = "3"
        res.append(r + (t(b),))
    os.environ["MM_DEBUG"] = "0"
    same = res[0][0] == res[1][0]
    print(f"k={k} w={w} canon={canon} mode={mode}: A {res[0][1]:.3f} / {res[2][1]:.3f} ms (walk {res[0][2]:.3f}) | B {res[1][1]:.3f} / {res[3][1]:.3f} ms "
          f"(walk {res[1][2]:.3f}) | B {n / min(res[1][1], res[3][1]) / 1e6:.0f} Gbases/s | outputs {'SAME' if same else 'DIFFERENT'} {res[1][0][0]}", flush=True)
