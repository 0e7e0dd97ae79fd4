"""What phase 2 costs (3.1 Gbp, k=21 w=11, timing build, default lanes): MM_DEBUG 0 whole kernel, 8 copy-out without its stores
(the instructions stay, the descriptor drops the stores), 2 no copy-out, 1 no look-back, 3 the walk alone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
import torch
import simd_minimizers_amd as sm
n = 3_100_000_000
d = sm.generate_device(n, 3); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
os.environ["MM_JIT_FORCE"] = "1"
os.environ["MM_JIT_DEFS"] = sys.argv[1] if len(sys.argv) > 1 else ""
for name, b in (("canonical", sm.canonical_minimizers(21, 11)), ("forward", sm.minimizers(21, 11))):
    def t(warm=12, reps=10):
        for _ in range(warm): b.run_device(d, n, out, sync=False)
        ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
        for _ in range(reps): b.run_device(d, n, out, sync=False)
        ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
        return ms / l
    row = []
    for dbg in (0, 8, 2, 1, 3, 0):
        os.environ["MM_DEBUG"] = str(dbg)
        row.append(f"MM_DEBUG={dbg}: {t():.4f}")
    os.environ["MM_DEBUG"] = "0"
    print(f"{name:10s} " + " | ".join(row), flush=True)
