"""Skip-ambiguous dirty walk (an isolated N every 8 kbp: every wave dirty) under compile-time options of the fused kernel:
tools/gpu_skip_defs_ab.py "<defs A>" "<defs B>" ... [--cfg k,w ...]   (experiments library: MM_LIB_PATH=..._exp.so)
Every build goes through the run-time specialisation (MM_JIT_FORCE=1, MM_JIT_DEFS=<defs>); prints kernel ms per Gbp of the
plain walk and of the dirty walk and an order-sensitive checksum of the dirty walk's output (must not move)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
import simd_minimizers_amd as sm
args = sys.argv[1:]
cfgs = [(31, 51), (31, 33)]
if "--cfg" in args:
    i = args.index("--cfg")
    cfgs = [tuple(int(x) for x in a.split(",")) for a in args[i + 1:]]
    args = args[:i]
n = 1_000_000_000
d = sm.generate_device(n, 2); ws = sm.default_workspace(0)
amb = torch.zeros((n + 7) // 8 + 64, dtype=torch.uint8, device="cuda")
g = torch.Generator(device="cuda"); g.manual_seed(1)
amb[torch.randint(0, n // 8, (n // 8000,), device="cuda", generator=g)] = 1 << 3
out = torch.zeros(int(n * 0.2), dtype=torch.int32, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
os.environ["MM_JIT_FORCE"] = "1"
def kt(step, warm=6, reps=6):
    for _ in range(warm): step()
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): step()
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / max(1, l)
def chk(c):
    v = out[:c].to(torch.int64)
    return c, int((v * torch.arange(1, c + 1, device="cuda")).sum().item())
for (k, w) in cfgs:
    b = sm.canonical_minimizers(k, w)
    ref = None
    for rnd in range(2):
        for defs in args:
            os.environ["MM_JIT_DEFS"] = defs
            out.zero_()
            c = chk(b.run_skip_ambiguous_device(d, amb, n, out))
            ref = ref or c
            dirty = kt(lambda: b.run_skip_ambiguous_device(d, amb, n, out, sync=False, d_count=cnt))
            plain = kt(lambda: b.run_device(d, n, out, sync=False, d_count=cnt))
            print(f"k={k} w={w} defs='{defs}': plain {plain:.3f} ms | dirty {dirty:.3f} ms | outputs {'SAME' if c == ref else 'DIFFERENT'} {c[0]}", flush=True)
