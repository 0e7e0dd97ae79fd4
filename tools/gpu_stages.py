"""Stage-incremental timing of the walk on the GPU, mirroring the reference's incremental
experiment (bench/src/bin/paper.rs:231-300: gather -> +nthash -> +sliding_min -> +canonical strand -> +collect ->
+dedup).  Stages 1-4 are timing builds of the kernel (MM_JIT_DEFS=-DMM_STAGE=n through the run-time specialisation;
wrong results by design), stage 5 is the product kernel without phase 2 (MM_DEBUG=3), stage 6 the product kernel.
All six go through the run-time specialisation so that they compare like with like.  3.1 Gbp, k=21 w=11."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")  # this script flips MM_* switches between runs (mm_env.h)
import simd_minimizers_amd as sm

n = int(os.environ.get("MM_N", "3100000000"))
d = sm.generate_device(n, 3)
ws = sm.default_workspace(0)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
os.environ["MM_JIT_FORCE"] = "1"

def t(b, warm=10, reps=10):
    for _ in range(warm): b.run_device(d, n, out, sync=False)
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(reps): b.run_device(d, n, out, sync=False)
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    return ms / l

ref = {"forward": [("gather2", 0.2996), ("nthash", 0.3185), ("sliding_min", 0.9037), (None, None), ("fwd-collect", 1.4847), ("fwd-dedup", 1.6129)],
       "canonical": [("gather2", 0.2996), ("canonical-nthash (incl. sliding_min)", 1.0359), (None, None), ("canonical-strand", 1.5283), ("canonical-collect", 2.0282), ("canonical-dedup", 2.1976)]}
names = ["1 loads + 2-bit decode", "2 + table look-ups, hash roll", "3 + keys, leftmost sliding min",
         "4 + rightmost min, strand vote", "5 + emit to the lane lists (walk complete)", "6 + look-back, copy-out (full kernel)"]
res = {}
for label, canon in (("forward", False), ("canonical", True)):
    b = sm.Builder(21, 11, canon, 0)
    rows = []
    for st in range(1, 7):
        if st <= 4:
            os.environ["MM_JIT_DEFS"] = f"-DMM_STAGE={st}"
            os.environ["MM_DEBUG"] = "3"
        else:
            os.environ.pop("MM_JIT_DEFS", None)
            os.environ["MM_DEBUG"] = "3" if st == 5 else "0"
        if st == 4 and not canon:
            rows.append(None)
            continue
        ms = t(b)
        rows.append(ms)
        r = ref[label][st - 1]
        print(f"{label:9s} stage {names[st - 1]:45s} {ms:7.3f} ms  {ms / n * 1e9:7.2f} ps/base  (+{(ms - (([x for x in rows[:-1] if x] or [0])[-1])):6.3f} ms)"
              + (f"   reference, 1 CPU thread (bench/results.json, not measured here): {r[0]} {r[1]} ns/base" if r[0] else ""), flush=True)
    res[label] = rows
os.environ["MM_DEBUG"] = "0"
print(json.dumps({"n": n, "k": 21, "w": 11, "ms": res}))
