"""Forward and canonical k=21 w=11 on 3.1 Gbp and C2 (256 Mbp): kernel ms (HIP events), count and an order-sensitive checksum -
one line per configuration, so that two builds of the library (MM_LIB_PATH) can be compared run by run."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import simd_minimizers_amd as sm
ws = sm.default_workspace(0)
print("library:", sm.LIB_PATH, flush=True)
for (name, b, n, seed) in (("forward 3.1 Gbp", sm.minimizers(21, 11), 3_100_000_000, 3), ("C2 forward 256 Mbp", sm.minimizers(21, 11), 268_435_456, 2),
                           ("canonical 3.1 Gbp", sm.canonical_minimizers(21, 11), 3_100_000_000, 3), ("forward w=7 3.1 Gbp", sm.minimizers(21, 7), 3_100_000_000, 3)):
    if len(sys.argv) > 1 and not any(a in name for a in sys.argv[1:]):
        continue
    d = sm.generate_device(n, seed)
    out = torch.zeros(int(n * 0.27) + 1024, dtype=torch.int32, device="cuda")
    c = b.run_device(d, n, out)
    v = out[:c].to(torch.int64)
    chk = 0
    for a in range(0, c, 1 << 27):
        e = min(c, a + (1 << 27))
        chk = (chk + int((v[a:e] * torch.arange(a + 1, e + 1, device="cuda")).sum().item())) & ((1 << 64) - 1)
    del v
    res = []
    for rep in range(3):
        for _ in range(40): b.run_device(d, n, out, sync=False)
        ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
        for _ in range(10): b.run_device(d, n, out, sync=False)
        ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
        res.append(ms / l)
    print(f"{name:22s} count {c} checksum {chk:#018x} kernel ms {' '.join(f'{x:.4f}' for x in res)} -> {n / min(res) / 1e6:.0f} Gbases/s", flush=True)
    del d, out
