"""what would lanes beyond the 31-block L2 bound buy the headline kernel (canonical k=21 w=11, 3.1 Gbp) if the
sequence streams were staged through LDS rows and the lists stayed small?  An UPPER BOUND from the experiments build's timing
hooks, before any of it is built (WRONG RESULTS by design, times only):
  MM_DEBUG=16            lists of half the capacity, overflow ignored: lanes of 40 .. 56 blocks keep four workgroups per CU
                         (what one-byte list entries would give)
  -DMM_EXP_LOADSAME      every lane of a wave reads the same cache line: no L2 footprint of the lanes' spans at all - better than
                         any staging could be, and without its LDS or its instructions
Run with MM_LIB_PATH=.../libsimd_minimizers_amd_exp.so."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
import torch
import simd_minimizers_amd as sm
n = 3_100_000_000
d = sm.generate_device(n, 3); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
b = sm.canonical_minimizers(21, 11)


def t(warm=12, reps=10):
    for _ in range(warm): b.run_device(d, n, out, sync=False)
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, sync=False)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l


print("product kernel, default lanes (28 blocks, tapered tail):", f"{t():.4f} ms", flush=True)
os.environ["MM_JIT_FORCE"] = "1"
for defs, label in (("", "MM_DEBUG=16 (half-size lists)"), ("-DMM_EXP_LOADSAME", "MM_DEBUG=16 + every lane reads the same line")):
    os.environ["MM_JIT_DEFS"] = defs
    os.environ["MM_DEBUG"] = "16"
    row = []
    for nb in (28, 34, 40, 48, 56, 64):
        ws.set_blocks_per_lane(nb)
        row.append(f"{nb}: {t():.4f}")
    ws.set_blocks_per_lane(0)
    print(f"{label:48s} blocks per lane (uniform tiles) -> ms | " + " | ".join(row), flush=True)
os.environ["MM_DEBUG"] = "0"
os.environ["MM_JIT_DEFS"] = ""
row = []
for nb in (22, 28, 31):
    ws.set_blocks_per_lane(nb)
    row.append(f"{nb}: {t():.4f}")
ws.set_blocks_per_lane(0)
print(f"{'run-time specialised product kernel, pinned lanes':48s} blocks per lane (uniform tiles) -> ms | " + " | ".join(row), flush=True)
