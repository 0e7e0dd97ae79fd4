"""Split path (walk kernel + expander on a second stream, MM_SPLIT=1) against the fused kernel: outputs must be
identical; kernel time by the workspace's events (walk start -> join) and by wall clock over several steps."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")  # this script flips MM_* switches between runs (mm_env.h)
import simd_minimizers_amd as sm

def checksum(out, c):
    v = out[:c].to(torch.int64)
    idx = torch.arange(1, c + 1, device=out.device, dtype=torch.int64)
    return int(c), int((v * idx).sum().item()), int(v.sum().item())

def run(b, d, n, out, split, env=None):
    os.environ["MM_SPLIT"] = "1" if split else "0"
    for k_, v_ in (env or {}).items(): os.environ[k_] = v_
    out.zero_()
    c = b.run_device(d, n, out)
    path = b._ws().last_path()
    for k_ in (env or {}): del os.environ[k_]
    return checksum(out, c), path

def timeit(b, d, n, out, split, warm=10, reps=12, env=None):
    os.environ["MM_SPLIT"] = "1" if split else "0"
    for k_, v_ in (env or {}).items(): os.environ[k_] = v_
    ws = b._ws()
    for _ in range(warm): b.run_device(d, n, out, sync=False)
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    t0 = time.perf_counter()
    for _ in range(reps): b.run_device(d, n, out, sync=False)
    ws.sync(); wall = (time.perf_counter() - t0) / reps * 1e3
    ms, l = ws.kernel_time(True); ws.enable_timing(False)
    for k_ in (env or {}): del os.environ[k_]
    return ms / l, wall

sizes = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["16000000", "3100000000"])]
cfgs = [(21, 11, False, 0), (21, 11, True, 0)]
for n in sizes:
    d = sm.generate_device(n, 3)
    out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
    for (k, w, canon, mode) in cfgs:
        b = sm.Builder(k, w, canon, mode)
        ref, p0 = run(b, d, n, out, False)
        got, p1 = run(b, d, n, out, True)
        print(f"n={n} k={k} w={w} canon={canon}: fused {ref} path {p0} | split {got} path {p1} | {'SAME' if ref == got else 'DIFFERENT'}", flush=True)
        if n >= 100_000_000:
            f = timeit(b, d, n, out, False)
            line = f"   fused {f[0]:.3f} ms (wall {f[1]:.3f})"
            for E in ("128", "256", "384", "512", "768"):
                s = timeit(b, d, n, out, True, env={"MM_SPLIT_E": E})
                line += f" | split E={E}: {s[0]:.3f} (wall {s[1]:.3f})"
            s = timeit(b, d, n, out, True, env={"MM_SPLIT_NO_REDO": "1"})
            line += f" | split, no redo launch: {s[0]:.3f} (wall {s[1]:.3f})"
            f = timeit(b, d, n, out, False)
            line += f" | fused again {f[0]:.3f}"
            print(line, flush=True)
    del d, out
# low-complexity input: every tile overflows -> redo pass
n = 4_000_000
packed = np.zeros(n // 4, dtype=np.uint8)  # poly-A
packed[n // 16: n // 8] = np.random.default_rng(1).integers(0, 256, n // 8 - n // 16, dtype=np.uint8)
d = torch.from_numpy(packed).cuda()
out = torch.zeros(n + 1024, dtype=torch.int32, device="cuda")
for (k, w, canon, mode) in cfgs:
    b = sm.Builder(k, w, canon, mode)
    ref, p0 = run(b, d, n, out, False)
    got, p1 = run(b, d, n, out, True)
    print(f"poly-A n={n} k={k} w={w} canon={canon}: fused {ref} | split {got} path {p1} | {'SAME' if ref == got else 'DIFFERENT'}", flush=True)
