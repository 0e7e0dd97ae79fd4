import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import simd_minimizers_amd as sm, mm_oracle as o
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
host = o.gen_packed(1, n)
d = torch.from_numpy(host).cuda()
out = torch.zeros(4 * n, dtype=torch.int32, device="cuda")
for k, w, canonical, mode in [(21, 11, False, 0), (21, 11, True, 0), (21, 11, True, 1), (31, 51, True, 0)]:
    b = sm.Builder(k, w, canonical, mode)
    try:
        c = b.run_device(d, n, out)
    except Exception as e:
        print(k, w, canonical, mode, "ERR", e); continue
    got = out[:c].cpu().numpy().view(np.uint32)
    want = o.run(host, n, k, w, canonical=canonical, mode=mode)
    eq = c == len(want) and np.array_equal(got, want)
    print(k, w, canonical, mode, "equal:", eq, c, len(want))
    if not eq:
        m = min(len(got), len(want)); bad = np.nonzero(got[:m] != want[:m])[0]
        print(" first mismatch idx", bad[:3], "got", got[bad[:3][0]-2:bad[:3][0]+6] if len(bad) else None, "want", want[bad[:3][0]-2:bad[:3][0]+6] if len(bad) else None)
