import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import simd_minimizers_amd as sm, mm_oracle as o
n = 200_000
host = o.gen_packed(1, n)
d = torch.from_numpy(host).cuda()
out = torch.zeros(4 * n, dtype=torch.int32, device="cuda")
os.environ["MM_DEBUG"] = sys.argv[1] if len(sys.argv) > 1 else "8"
ws = sm.default_workspace(0)
for canonical in (False, True):
    b = sm.Builder(21, 11, canonical, 0)
    out.zero_()
    try: b.run_device(d, n, out)
    except Exception as e: print("err", e)
    r = out[:256 + 256 * 8].cpu().numpy()
    print("canon", canonical, "counts lanes 0..7:", r[:8], " heads lane0:", r[256:264], "lane1:", r[264:272], "lane2:", r[272:280])
    S = 264
    win = o.window_positions(host, n, 21, 11, o.default_hasher(canonical), canonical)
    for L in range(3):
        a = L * S
        w = win[a:a + S]; prevp = win[a - 1] if a > 0 else -1
        ent = [int(x) for i, x in enumerate(w) if x != (w[i - 1] if i > 0 else prevp)]
        print("  want lane", L, "count", len(ent), "first e':", [x - (a - 1) for x in ent[:8]])
