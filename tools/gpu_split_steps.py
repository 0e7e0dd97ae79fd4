"""A few steps of one configuration on 3.1 Gbp, for rocprofv3 --kernel-trace (split path: MM_SPLIT=1, MM_SPLIT_E=..).
usage: gpu_split_steps.py <canon 0|1> <k> <w> [steps]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import simd_minimizers_amd as sm
canon, k, w = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 8
n = int(os.environ.get("MM_N", "3100000000"))
d = sm.generate_device(n, 3)
out = torch.zeros(int(n * 0.2) + 1024, dtype=torch.int32, device="cuda")
b = sm.Builder(k, w, bool(canon), 0)
for _ in range(steps): b.run_device(d, n, out, sync=False)
b._ws().sync()
print("path", b._ws().last_path())
