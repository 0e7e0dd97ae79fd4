"""Stress of the scan: the same 1 Gbp run with very short lanes (hundreds of thousands of tiles), in
ticket mode, and repeated; all outputs must be identical (count + order-sensitive checksum)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")  # this script flips MM_* switches between runs (mm_env.h)
import simd_minimizers_amd as sm
n = 1_000_000_000
d = sm.generate_device(n, 4); ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.55), dtype=torch.int32, device="cuda")
def checksum(c):
    v = out[:c].to(torch.int64) & 0xFFFFFFFF
    idx = torch.arange(1, c + 1, dtype=torch.int64, device="cuda")
    return c, int(v.sum().item()), int((v * idx).sum().item())  # wraps mod 2^64 consistently
for k, w, canon, mode in [(21, 11, True, 0), (21, 11, False, 0), (15, 17, True, 1), (5, 3, False, 0)]:
    b = sm.Builder(k, w, canon, mode)
    ref = None
    for nblk, ticket, rep in [(0, 0, 0), (1, 0, 0), (2, 0, 0), (3, 0, 0), (7, 0, 0), (0, 1, 0), (1, 1, 0), (0, 0, 1), (0, 0, 2), (1, 0, 1)]:
        ws.set_blocks_per_lane(nblk)
        if ticket: os.environ["MM_FORCE_TICKET"] = "1"
        else: os.environ.pop("MM_FORCE_TICKET", None)
        cs = checksum(b.run_device(d, n, out))
        if ref is None: ref = cs
        status = "ok" if cs == ref else "MISMATCH"
        print(f"k={k} w={w} canon={canon} mode={mode} nblk={nblk or 'default'} ticket={ticket} rep={rep}: count={cs[0]} {status}", flush=True)
        assert cs == ref
ws.set_blocks_per_lane(0)
print("stress ok")
