"""Round 6: lane-table launches (mm_lanes.hip).  Correctness of mixed read lengths against the oracle, then the kernel and
whole-call times of (a) 200 k reads log-uniform in 1 .. 50 kbp, (b) 20 000 x 10 kbp through mm_run_batch_device, (c) 8 M reads
of 100 .. 200 bp, each with the lane table on and off (MM_LANE_TABLE)."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
import numpy as np, torch
import simd_minimizers_amd as sm
import mm_oracle as oracle
ws = sm.default_workspace(0); L = sm.lib()
what = sys.argv[1:] or ["check", "a", "b", "c"]


def packed_reads(lens, seed):
    lens = np.asarray(lens, dtype=np.int64)
    starts = np.zeros(len(lens) + 1, dtype=np.int64); starts[1:] = np.cumsum(lens)
    total = int(starts[-1])
    d = sm.generate_device(max(total, 1), seed)
    return d, torch.from_numpy(starts).cuda(), starts, total


def run_packed(b, d, ds, n, total, mx, out, offs, cnt, sk=None):
    sm._check(L.mm_run_packed_reads_device_async(b.plan().h, ws.h, C.c_void_p(d.data_ptr()), d.numel(), 0, n, C.c_void_p(ds.data_ptr()),
                                                 total, mx, C.c_void_p(out.data_ptr()), C.c_void_p(sk.data_ptr()) if sk is not None else None,
                                                 out.numel(), C.c_void_p(offs.data_ptr()), C.c_void_p(cnt.data_ptr())))


def check(lens, k, w, canonical, mode, sk=False, sample=None, seed=5):
    d, ds, starts, total = packed_reads(lens, seed)
    host = d.cpu().numpy()
    b = sm.Builder(k, w, canonical, mode)
    out = torch.zeros(max(1, total), dtype=torch.int32, device="cuda")
    osk = torch.zeros_like(out) if sk else None
    offs = torch.full((len(lens) + 1,), -1, dtype=torch.int64, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    run_packed(b, d, ds, len(lens), total, int(max(lens)), out, offs, cnt, osk)
    ws.sync(); sm._check(L.mm_workspace_check(ws.h))
    tot = int(cnt.item()); ho = offs.cpu().numpy()
    assert ho[0] == 0 and ho[-1] == tot, (ho[:4], ho[-1], tot)
    assert np.all(np.diff(ho) >= 0)
    flat = out[:tot].cpu().numpy().view(np.uint32)
    fsk = osk[:tot].cpu().numpy().view(np.uint32) if sk else None
    bad = 0
    for r in (sample if sample is not None else range(len(lens))):
        s0 = int(starts[r]); ln = int(lens[r])
        res = oracle.run(host[s0 // 4:], ln, k, w, canonical=canonical, mode=mode, super_kmers=sk, base_offset=s0 % 4)
        wp = res[0] if sk else res
        got = flat[ho[r]: ho[r + 1]]
        if not np.array_equal(got, wp):
            bad += 1
            if bad < 4:
                print("MISMATCH read", r, "len", ln, "got", len(got), "want", len(wp), got[:6], wp[:6], flush=True)
        if sk and not np.array_equal(fsk[ho[r]: ho[r + 1]], res[1]):
            bad += 1
    print(f"check k={k} w={w} canon={canonical} mode={mode} sk={sk}: {len(lens)} reads, {total} bases, {tot} positions, lane_table={ws.last_lane_table()}, mismatches {bad}", flush=True)
    return bad


def timeit(step, total, label, n=10):
    for _ in range(3): step()
    ws.sync(); ws.enable_timing(True); ws.kernel_time(True)
    for _ in range(n): step()
    ws.sync(); ms, l = ws.kernel_time(True); ws.enable_timing(False)
    t0 = time.perf_counter()
    for _ in range(n): step()
    ws.sync(); wall = (time.perf_counter() - t0) / n * 1e3
    print(f"{label}: walk kernel {ms / max(l, 1):.3f} ms x {l / n:.1f} launches = {total / (ms / n) / 1e6:.0f} Gbases/s; whole call {wall:.3f} ms = {total / wall / 1e6:.0f} Gbases/s; lane_table={ws.last_lane_table()}", flush=True)


rng = np.random.default_rng(61)
if "check" in what:
    bad = 0
    lens = rng.integers(0, 3000, 400); lens[:6] = [0, 30, 31, 70_001, 309, 5000]
    bad += check(lens, 21, 11, True, 0)
    bad += check(lens, 21, 11, False, 0)
    bad += check(lens[:200], 21, 11, True, 0, sk=True)
    bad += check(lens[:150], 15, 17, True, 1)
    bad += check(lens[:150], 15, 17, True, 2)
    bad += check(rng.integers(0, 9000, 300), 31, 51, True, 0)
    bad += check(rng.integers(0, 2000, 300), 5, 7, False, 0)
    os.environ["MM_LANE_TABLE"] = "1"
    bad += check(rng.integers(0, 400, 3000), 21, 11, True, 0, sample=range(0, 3000, 5))
    bad += check(rng.integers(0, 400, 2000), 31, 19, False, 0, sample=range(0, 2000, 5))
    del os.environ["MM_LANE_TABLE"]
    print("TOTAL MISMATCHES", bad, flush=True)

b = sm.canonical_minimizers(21, 11)
if "a" in what:
    n = 200_000
    lens = np.exp(rng.uniform(np.log(1000), np.log(50_000), n)).astype(np.int64)
    d, ds, starts, total = packed_reads(lens, 7)
    out = torch.empty(int(total * 0.18), dtype=torch.int32, device="cuda")
    offs = torch.zeros(n + 1, dtype=torch.int64, device="cuda"); cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    step = lambda: run_packed(b, d, ds, n, total, int(lens.max()), out, offs, cnt)
    timeit(step, total, f"(a) {n} reads log-uniform 1-50 kbp, {total / 1e9:.2f} Gbp, packed reads")
    host = d[: 4_000_000].cpu().numpy(); ho = offs.cpu().numpy(); bad = 0
    for r in range(0, 200):
        s0 = int(starts[r])
        if s0 // 4 + int(lens[r]) // 4 + 8 > len(host): break
        want = oracle.run(host[s0 // 4:], int(lens[r]), 21, 11, canonical=True, base_offset=s0 % 4)
        got = out[ho[r]: ho[r + 1]].cpu().numpy().view(np.uint32)
        bad += not np.array_equal(got, want)
    print("(a) first reads against the oracle: mismatches", bad, "count", int(cnt.item()), flush=True)
    del d, out
if "b" in what:
    n, ln = 20_000, 10_000
    d = sm.generate_device(n * ln, 9)
    seqs = [d[(i * ln) // 4:] for i in range(n)]
    out = torch.empty(int(n * ln * 0.18), dtype=torch.int32, device="cuda")
    for pol in ("1", "0"):
        os.environ["MM_LANE_TABLE"] = pol
        step = lambda: sm.run_batch_device(b, seqs, [ln] * n, out)
        timeit(step, n * ln, f"(b) {n} x {ln} bp, mm_run_batch_device, MM_LANE_TABLE={pol}", n=5)
    del os.environ["MM_LANE_TABLE"]
    offs = sm.run_batch_device(b, seqs, [ln] * n, out)
    host = d.cpu().numpy(); bad = 0
    for s in range(0, n, 997):
        want = oracle.run(host[(s * ln) // 4:], ln, 21, 11, canonical=True, base_offset=(s * ln) % 4)
        got = out[offs[s]: offs[s + 1]].cpu().numpy().view(np.uint32)
        bad += not np.array_equal(got, want)
    print("(b) sampled sequences against the oracle: mismatches", bad, "lane_table", ws.last_lane_table(), flush=True)
    del d, out
if "c" in what:
    n = 8_000_000
    lens = rng.integers(100, 201, n)
    d, ds, starts, total = packed_reads(lens, 11)
    out = torch.empty(int(total * 0.2), dtype=torch.int32, device="cuda")
    offs = torch.zeros(n + 1, dtype=torch.int64, device="cuda"); cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    for pol in ("0", "1"):
        os.environ["MM_LANE_TABLE"] = pol
        step = lambda: run_packed(b, d, ds, n, total, 200, out, offs, cnt)
        timeit(step, total, f"(c) 8 M reads of 100..200 bp, MM_LANE_TABLE={pol}")
    del os.environ["MM_LANE_TABLE"]
