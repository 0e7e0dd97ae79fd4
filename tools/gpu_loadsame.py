"""Is the walk's load path (64 lanes x 8 bytes, about 31 cache lines per instruction) a limiter?  Timing builds with
every lane of a wave reading the SAME line (MM_EXP_LOADSAME, wrong results) against the real addresses, for the
decode-only stage and the whole walk.  3.1 Gbp k=21 w=11."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")  # this script flips MM_* switches between runs (mm_env.h)
import simd_minimizers_amd as sm
n = 3_100_000_000
d = sm.generate_device(n, 3); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
os.environ["MM_JIT_FORCE"] = "1"
def t(b, warm=8, reps=8):
    for _ in range(warm): b.run_device(d, n, out, sync=False)
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, sync=False)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l
for k, w, canon in ((21, 11, False), (21, 11, True), (31, 51, True)):
    b = sm.Builder(k, w, canon, 0)
    for defs, dbg, name in (("-DMM_STAGE=1", "3", "decode only"), ("-DMM_STAGE=1 -DMM_EXP_LOADSAME", "3", "decode only, one line per wave-load"),
                            ("", "3", "walk"), ("-DMM_EXP_LOADSAME", "3", "walk, one line per wave-load"),
                            ("", "0", "full kernel"), ("-DMM_EXP_LOADSAME", "0", "full kernel, one line per wave-load")):
        if defs: os.environ["MM_JIT_DEFS"] = defs
        else: os.environ.pop("MM_JIT_DEFS", None)
        os.environ["MM_DEBUG"] = dbg
        print(f"k={k} w={w} canon={canon} {name:40s} {t(b):.3f} ms", flush=True)
