"""Sweep blocks-per-lane for several (k, w) configs."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.gpu_probe import run
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
run(n, 21, 11, True, 0, [20, 24, 28, 32, 40], reps=5)
run(n, 21, 11, False, 0, [24, 32, 40], reps=5)
run(n, 31, 51, True, 0, [8, 10, 12], reps=5)
run(n, 15, 17, True, 1, [12, 16, 20], reps=5)
run(n, 5, 7, False, 0, [20, 30, 40], reps=5)
run(n, 31, 5, True, 0, [30, 50, 70], reps=5)
run(n, 19, 19, True, 0, [8, 12, 16], reps=5)
