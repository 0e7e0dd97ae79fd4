"""A/B of two builds of the READS-mode kernel on one box (run-time specialisation, experiments build): tools/gpu_reads_defs_ab.py
"<defs A>" "<defs B>" - the rows READS (8 M x 150 bp), READS_VAR, LONGREADS, BATCH10K of simd_minimizers_amd.workloads, A / B / A / B."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
import torch
import simd_minimizers_amd as sm
from simd_minimizers_amd import workloads
ws = sm.default_workspace(0)
os.environ["MM_JIT_FORCE"] = "1"
A, B = sys.argv[1], sys.argv[2]
names = sys.argv[3].split(",") if len(sys.argv) > 3 else ["READS", "READS_VAR", "LONGREADS", "BATCH10K"]
print(f"A = '{A}'   B = '{B}'", flush=True)
for nme in names:
    res = []
    for defs in (A, B, A, B):
        os.environ["MM_JIT_DEFS"] = defs
        r = workloads.measure(nme, ws, "cuda:0")
        res.append(r["ms"])
    print(f"{nme:10s} A {res[0]:.4f} / {res[2]:.4f} ms | B {res[1]:.4f} / {res[3]:.4f} ms | B/A {min(res[1], res[3]) / min(res[0], res[2]):.3f}", flush=True)
