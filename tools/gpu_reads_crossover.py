"""Reads policy crossover: one lane per read (MM_LANE_TABLE=0) against the lane table (=1) on rungs of the ladder (lengths uniform in
[n, 2n), 2^30 bases), forward and canonical k=21 w=11; whole-call device ms."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
import simd_minimizers_amd as sm
from simd_minimizers_amd import workloads
ws = sm.default_workspace(0)
for n in (64, 128, 192, 256, 384, 512, 1024):
    row = []
    for fl in ("F", "C"):
        for pol in ("0", "1", None):
            if pol is None: os.environ.pop("MM_LANE_TABLE", None)
            else: os.environ["MM_LANE_TABLE"] = pol
            try:
                r = workloads.measure(f"LADDER_{n}_{fl}", ws, "cuda:0")
                row.append(f"{fl} {'one lane' if pol == '0' else 'table' if pol == '1' else 'policy'} {r['ms']:.4f}{'*' if r['lane_table'] else ''}")
            except Exception as e:
                row.append(f"{fl} {pol} error {str(e)[:40]}")
    os.environ.pop("MM_LANE_TABLE", None)
    print(f"n={n:5d}: " + " | ".join(row), flush=True)
