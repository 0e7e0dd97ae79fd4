"""Forward k=21 w=11 on 3.1 Gbp over the list-length limit that sizes the lanes (MM_CAP_LIMIT), for the library as
built (8-bit lists, S + w <= 255) or a -DMM_NO_E8 build (16-bit lists, longer lanes possible)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
import torch
import simd_minimizers_amd as sm
n = 3_100_000_000
d = sm.generate_device(n, 3); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
b = sm.minimizers(21, 11)
def t(warm=10, reps=12):
    for _ in range(warm): b.run_device(d, n, out, sync=False)
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, sync=False)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l
for cap in sys.argv[1:] or ["0"]:
    if cap != "0": os.environ["MM_CAP_LIMIT"] = cap
    else: os.environ.pop("MM_CAP_LIMIT", None)
    print(f"lib {os.path.basename(sm.LIB_PATH)} cap limit {cap}: {t():.3f} ms", flush=True)
