"""Whole synchronous call of mm_run_batch_device on 20 000 contigs of 10 kbp (host loops, uploads of the starts and lengths, the lane
table's kernels, the walk, the offsets back): wall clock per call with the ctypes arrays prebuilt."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import simd_minimizers_amd as sm
from simd_minimizers_amd import workloads
ws = sm.default_workspace(0)
c = workloads.component("BATCH10K", ws, "cuda:0")
for _ in range(5): c["step"]()
t0 = time.perf_counter()
for _ in range(20): c["step"]()
print("mm_run_batch_device, 20000 x 10 kbp, whole synchronous call (ctypes arrays prebuilt):", (time.perf_counter() - t0) / 20 * 1e3, "ms")
