"""Probe with an alternate library (MM_LIB env) and selected configs."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import simd_minimizers_amd as sm
if os.environ.get("MM_LIB"):
    sm.LIB_PATH = os.environ["MM_LIB"]
from tools.gpu_probe import run
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
run(n, 21, 11, False, 0, [12, 16, 20, 24], reps=5)
run(n, 21, 11, True, 0, [12, 16, 20, 24], reps=5)
run(n, 31, 51, True, 0, [3, 4, 6, 8], reps=5)
run(n, 15, 17, True, 1, [6, 8, 10, 12], reps=5)
