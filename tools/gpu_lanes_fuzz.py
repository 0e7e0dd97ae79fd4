"""Randomised differential test of the LANE-TABLE launch against the oracle (one-off; tests/test_gpu_fuzz.py carries a share of it):
random plans (any prebuilt window size and a few run-time specialised ones, canonical / forward, minimizers / closed / open syncmers,
super-k-mer indices), read-length mixes (empty, below a window, around multiples of the lane length, long), pinned lane lengths, base
offsets and unaligned pointers, packed-starts and fixed-stride layouts, batches through mm_run_batch_device.  usage: gpu_lanes_fuzz.py seed iterations"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
os.environ["MM_LANE_TABLE"] = "1"
import numpy as np, torch
import mm_oracle as oracle
import simd_minimizers_amd as sm
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(9000 + seed)
ws = sm.default_workspace(0); L = sm.lib()
pre = {c: sm.prebuilt_window_sizes(c, True) for c in (True, False)}
tally = dict(cases=0, reads=0, bases=0, positions=0, multi_lane_reads=0)
for it in range(iters):
    canonical = bool(rng.integers(0, 2))
    w = int(rng.choice(pre[canonical])) if rng.integers(0, 5) else int(rng.choice([18, 20, 35, 36, 37, 49, 64, 65, 97]))
    mode = int(rng.choice([0, 0, 0, 1, 2]))
    if mode == 2 and w % 2 == 0: mode = 1
    k = int(rng.integers(1, 65))
    if canonical and (k + w - 1) % 2 == 0: k = k + 1 if k < 64 else k - 1
    sk = mode == 0 and bool(rng.integers(0, 3) == 0)
    l = k + w - 1
    nb = int(rng.choice([0, 0, 1, 2, 3, 5, 9]))
    plan6 = (C.c_uint64 * 6)()
    assert L.mm_debug_lane_plan(k, w, int(canonical), 3 if sk else mode, 1, 1000, nb, plan6) == 0
    S = int(plan6[1])
    n_reads = int(rng.integers(1, 60))
    kinds = rng.integers(0, 7, n_reads)
    lens = np.where(kinds == 0, rng.integers(0, l + 2, n_reads),
           np.where(kinds == 1, l - 1 + S * rng.integers(1, 4, n_reads) + rng.integers(-2, 3, n_reads),
           np.where(kinds == 2, rng.integers(l, l + 3 * S + 5, n_reads), rng.integers(0, min(40 * S + 200, 30_000), n_reads)))).astype(np.int64)
    lens = np.maximum(lens, 0)
    off, shift = int(rng.integers(0, 4)), int(rng.integers(0, 4))
    layout = int(rng.integers(0, 3))  # 0 packed starts, 1 fixed stride + lengths, 2 batch of device sequences
    if layout == 1:
        read_len = int(lens.max()) if lens.max() > 0 else 1
        stride = read_len + int(rng.integers(0, 9))
        starts = np.arange(n_reads + 1, dtype=np.int64) * stride
        total = n_reads * stride
    else:
        gaps = rng.integers(0, 6, n_reads) if layout == 2 else np.zeros(n_reads, dtype=np.int64)
        starts = np.zeros(n_reads + 1, dtype=np.int64); starts[1:] = np.cumsum(lens + gaps)
        total = int(starts[-1])
    data = oracle.gen_packed(int(rng.integers(1 << 30)), off + total + 64)
    dev = torch.zeros(len(data) + 8, dtype=torch.uint8, device="cuda")
    dev[shift: shift + len(data)] = torch.from_numpy(data).cuda()
    d = dev[shift:]
    b = sm.Builder(k, w, canonical, mode)
    cap = max(1, int(lens.sum()))
    out = torch.full((cap + 8,), -7, dtype=torch.int32, device="cuda")
    osk = torch.zeros_like(out) if sk else None
    ws.set_blocks_per_lane(nb)
    case = dict(seed=seed, it=it, k=k, w=w, canonical=canonical, mode=mode, sk=sk, nb=nb, S=S, layout=layout, off=off, shift=shift, n_reads=n_reads)
    try:
        if layout == 2:
            seqs = [d[(off + int(s)) // 4:] for s in starts[:-1]]
            boffs = [(off + int(s)) % 4 for s in starts[:-1]]
            ho = np.array(sm.run_batch_device(b, seqs, [int(x) for x in lens], out[:cap], osk[:cap] if sk else None, base_offsets=boffs), dtype=np.int64)
            tot = int(ho[-1])
        else:
            offs = torch.full((n_reads + 1,), -1, dtype=torch.int64, device="cuda")
            if layout == 0:
                ds = torch.from_numpy(starts).cuda(); cnt = C.c_uint64()
                sm._check(L.mm_run_packed_reads_device(b.plan().h, ws.h, C.c_void_p(d.data_ptr()), d.numel(), off, n_reads, C.c_void_p(ds.data_ptr()),
                                                       total, int(max(1, lens.max())), C.c_void_p(out.data_ptr()), C.c_void_p(osk.data_ptr()) if sk else None,
                                                       cap, C.c_void_p(offs.data_ptr()), C.byref(cnt)))
                tot = int(cnt.value)
            else:
                dl = torch.from_numpy(lens.astype(np.int32)).cuda()
                tot = sm.run_reads_device(b, d, n_reads, stride, read_len, out[:cap], offs, read_lens=dl, base_offset=off, out_sk=osk[:cap] if sk else None)
            ho = offs.cpu().numpy()
        assert ws.last_lane_table(), case
    finally:
        ws.set_blocks_per_lane(0)
    assert ho[0] == 0 and ho[-1] == tot and int(out[tot].item()) == -7, (case, ho[:3], tot)
    flat = out[:tot].cpu().numpy().view(np.uint32)
    fsk = osk[:tot].cpu().numpy().view(np.uint32) if sk else None
    for r in range(n_reads):
        res = oracle.run(data, int(lens[r]), k, w, canonical=canonical, mode=mode, base_offset=off + int(starts[r]), super_kmers=sk)
        wp = res[0] if sk else res
        assert np.array_equal(flat[ho[r]: ho[r + 1]], wp), (case, r, int(lens[r]))
        if sk: assert np.array_equal(fsk[ho[r]: ho[r + 1]], res[1]), (case, r, "sk")
        tally["multi_lane_reads"] += int(max(0, int(lens[r]) - l + 1) > S)
    tally["cases"] += 1; tally["reads"] += n_reads; tally["bases"] += int(lens.sum()); tally["positions"] += tot
print("lane-table fuzz seed", seed, tally, flush=True)
