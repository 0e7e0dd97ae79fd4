"""JIT A/B for one (k, w, canonical, mode): kernel time on 3.1 Gbp for several MM_JIT_DEFS strings.
usage: gpu_jit_w.py k w canon mode "defs1" "defs2" ...   ("" = prebuilt-equivalent defaults, also JIT-compiled)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")  # this script flips MM_* switches between runs (mm_env.h)
import simd_minimizers_amd as sm
k, w, canon, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3] == "1", int(sys.argv[4])
n = 3_100_000_000
d = sm.generate_device(n, 3); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
os.environ["MM_JIT_FORCE"] = "1"
os.environ["MM_JIT_CACHE_DIR"] = ""
def t(b, warm=10, reps=12):
    for _ in range(warm): b.run_device(d, n, out, sync=False, d_count=cnt)
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, sync=False, d_count=cnt)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l
res = {}
for rnd in range(2):
    for defs in sys.argv[5:]:
        os.environ["MM_JIT_DEFS"] = defs
        b = sm.Builder(k, w, canon, mode)
        res.setdefault(defs, []).append(t(b))
        res.setdefault(defs + "#count", []).append(int(cnt.item()))
for defs in sys.argv[5:]:
    v = res[defs]
    print(f"k={k} w={w} canon={canon} mode={mode} {defs!r:50s}: " + " ".join(f"{x:.3f}" for x in v) + f" ms -> {n / min(v) / 1e6:.0f} Gbases/s count={res[defs + '#count'][0]}", flush=True)
