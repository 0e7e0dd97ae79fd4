import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import simd_minimizers_amd as sm, mm_oracle as o
n = 1_000_000
host = o.gen_packed(1, n)
d = torch.from_numpy(host).cuda()
out = torch.zeros(n, dtype=torch.int32, device="cuda")
ws = sm.default_workspace(0)
for k, w, canonical, mode in [(31, 5, False, 0), (21, 11, True, 0)]:
    b = sm.Builder(k, w, canonical, mode)
    c = b.run_device(d, n, out)
    got = out[:c].cpu().numpy().view(np.uint32)
    want = o.run(host, n, k, w, canonical=canonical, mode=mode)
    print(k, w, "pos equal:", c == len(want) and np.array_equal(got, want), c, len(want))
    vals = torch.zeros(c, dtype=torch.int64, device="cuda")
    sm._check(sm.lib().mm_values_u64_device_async(ws.h, C.c_void_p(d.data_ptr()), d.numel(), 0, n, k, int(canonical), C.c_void_p(out.data_ptr()), c, C.c_void_p(vals.data_ptr())))
    ws.sync()
    gv = vals.cpu().numpy().view(np.uint64)
    wv = o.values_u64(host, k, got, canonical)
    bad = np.nonzero(gv != wv)[0]
    print("values mismatches:", len(bad), bad[:10], [int(got[i]) for i in bad[:10]])
    for i in bad[:5]:
        print(hex(int(gv[i])), hex(int(wv[i])))
