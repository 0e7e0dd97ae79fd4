"""The reads-side rows of bench.py alone (READS, READS_SK, READS_VAR, LONGREADS, BATCH10K and the ladder): quick A/B of the
reads-mode kernels."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import simd_minimizers_amd as sm
from simd_minimizers_amd import workloads
ws = sm.default_workspace(0)
names = sys.argv[1:] or ["READS", "READS_SK", "READS_VAR", "LONGREADS", "BATCH10K", "LADDER"]
for nme in names:
    if nme == "LADDER":
        for r in workloads.ladder(ws, "cuda:0")["rows"]:
            print(json.dumps(r), flush=True)
    else:
        r = workloads.measure(nme, ws, "cuda:0")
        print(json.dumps({k: v for k, v in r.items() if k != "what"}), flush=True)
