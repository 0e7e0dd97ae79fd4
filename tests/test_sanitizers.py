"""Sanitizer passes on the CPU side (GPU sanitizers are not available on this pool):

* the oracle - the checker every parity claim rests on - built with -fsanitize=address,undefined
  (oracle/Makefile `asan`) and driven through its whole CPU test file;
* (GPU box) the HOST side of the product - workspaces, tile tables of batch and reads runs, the pipelined and the
  sharded host entry points, capacity arithmetic - from a library whose host objects are built with clang's
  AddressSanitizer + UBSan (csrc/Makefile `hostasan`; device code is the normal build).
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gcc_runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_under_asan_ubsan():
    asan, ubsan = _gcc_runtime("libasan.so"), _gcc_runtime("libubsan.so")
    if not asan:
        pytest.skip("gcc has no libasan here")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True, capture_output=True)
    env = dict(os.environ, MM_ORACLE_LIB=os.path.join(ROOT, "oracle", "libmm_oracle_asan.so"),
               LD_PRELOAD=":".join(x for x in (asan, ubsan) if x),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle.py"), "-x", "-q",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and "passed" in r.stdout, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail


_HOST_SCRIPT = r"""
import os, sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/oracle")
import simd_minimizers_amd as sm
import mm_oracle as oracle
rng = np.random.default_rng(3)
k, w = 21, 11
b = sm.canonical_minimizers(k, w)
# batches: many tile tables of different shapes on one workspace (grow / reuse), empty and short sequences
for trial in range(6):
    lens = [int(x) for x in rng.integers(0, 400_000, size=int(rng.integers(1, 40)))] + [0, 5, 31]
    seqs = [oracle.gen_packed(100 + i, m + 3) for i, m in enumerate(lens)]
    d = [torch.from_numpy(s).cuda() for s in seqs]
    out = torch.zeros(sum(lens) // 3 + 64, dtype=torch.int32, device="cuda")
    offs = sm.run_batch_device(b, d, lens, out)
    for i in (0, len(lens) // 2, len(lens) - 1):
        want = oracle.run(seqs[i], lens[i], k, w, canonical=True)
        assert np.array_equal(out[offs[i]:offs[i + 1]].cpu().numpy().view(np.uint32), want)
# reads: fixed and per-read lengths, super-k-mer indices
n_reads, rl = 5000, 150
data = oracle.gen_packed(9, n_reads * rl)
d = torch.from_numpy(data).cuda()
out = torch.zeros(n_reads * rl // 2, dtype=torch.int32, device="cuda")
sk = torch.zeros_like(out)
offs = torch.zeros(n_reads + 1, dtype=torch.int64, device="cuda")
lens = torch.from_numpy(rng.integers(0, rl + 1, size=n_reads).astype(np.uint32)).cuda()
for kw in ({}, {"read_lens": lens}, {"out_sk": sk}):
    sm.run_reads_device(b, d, n_reads, rl, rl, out, offs, **kw)
# host entry points: one shot, pipelined (long), sharded over two workspaces, contigs over three
for n in (1000, 3_000_017, 60_000_000):
    data = oracle.gen_packed(5, n)
    got, _ = b._run_arrays(sm.PackedSeq(data, 0, n))
    if n < 10_000_000:
        assert np.array_equal(np.asarray(got, dtype=np.uint32), oracle.run(data, n, k, w, canonical=True))
g = sm.DeviceGroup([0, 0, 0])
data = oracle.gen_packed(6, 4_000_003)
pos, _ = g.run(b, data, 4_000_003)
assert np.array_equal(pos, oracle.run(data, 4_000_003, k, w, canonical=True))
lens = [300_000, 7, 0, 1_000_001, 250_000]
seqs = [oracle.gen_packed(60 + i, m + 3) for i, m in enumerate(lens)]
pos, _, o = g.run_batch(b, seqs, lens)
assert np.array_equal(pos[o[3]:o[4]], oracle.run(seqs[3], lens[3], k, w, canonical=True))
try:
    g.run(b, data, 4_000_003, capacity=100)
    raise SystemExit("capacity error expected")
except sm.MinimizerError:
    pass
g.close()
# FASTA text, values
text = b">a\nACGTACGTTTGACCA\nACGT\n>b\n\n>c\nTTTTGGGG\n" * 50
rec = sm.fasta_pack_device(text)
assert len(rec) == 150
# round 4: device-resident shards and batches of a group, FASTQ, packed reads (device and host), the launch planner
g = sm.DeviceGroup([0, 0])
data = oracle.gen_packed(6, 4_000_003)
g.upload(data[: (4_000_003 + 3) // 4 + 1])
for _ in range(3):
    counts = g.run_device(b, 4_000_003)
want = oracle.run(data, 4_000_003, k, w, canonical=True)
dst = torch.zeros(len(want) + 8, dtype=torch.int32, device="cuda")
assert g.gather(1, dst) == len(want) and np.array_equal(dst[: len(want)].cpu().numpy().view(np.uint32), want)
g.upload_range(data[: (4_000_003 + 3) // 4 + 1], 4_000_003)   # every entry only its share (+ halo) of the bytes
g.run_device(b, 4_000_003)
assert g.gather(0, dst) == len(want) and np.array_equal(dst[: len(want)].cpu().numpy().view(np.uint32), want)
try:
    g.run_device(b, 1_000_000)   # another shape: refused, the ranges named
    raise SystemExit("a run outside the resident ranges was accepted")
except sm.MinimizerError as e_:
    assert "holds bytes" in str(e_)
lens = [300_000, 7, 0, 1_000_001, 250_000]
seqs = [oracle.gen_packed(60 + i, m + 3) for i, m in enumerate(lens)]
g.upload_batch([s_[: (m + 3) // 4 + 1] for s_, m in zip(seqs, lens)])
cc = g.run_batch_device(b, lens)
o = g.gather_batch(0, dst)
assert np.array_equal(dst[o[3]:o[4]].cpu().numpy().view(np.uint32), oracle.run(seqs[3], lens[3], k, w, canonical=True))
g.close()
fq = b"".join(b"@r%d\n" % i + b"ACGTTGCATGCA" * (3 + i % 20) + b"\n+\n" + b"I" * (12 * (3 + i % 20)) + b"\n" for i in range(700))
rec = sm.fasta_pack_device(fq, max_records=1024)
assert len(rec) == 700
out = torch.zeros(int(rec.base[-1]) + 8, dtype=torch.int32, device="cuda")
offs = torch.zeros(701, dtype=torch.int64, device="cuda")
sm.run_packed_reads_device(b, rec, out, offs)
pos, ho, _ = sm.run_reads_host(b, [b"ACGTTGCATGCA" * (3 + i % 20) for i in range(300)])
assert ho[-1] == len(pos)
import ctypes as C
os.environ["MM_TAPER_SLOTS"] = "3"
arr = (C.c_uint64 * 3)(900_000, 0, 5_000_000)
seq_, w0_, nb_ = (C.c_uint32 * 4096)(), (C.c_uint32 * 4096)(), (C.c_uint32 * 4096)()
o7, nt = (C.c_uint64 * 7)(), C.c_uint64()
assert sm.lib().mm_debug_launch_plan(11, 1, 0, 3, arr, o7, seq_, w0_, nb_, 4096, C.byref(nt)) == 0 and nt.value > 3
o2 = (C.c_uint64 * 2)()
assert sm.lib().mm_debug_launch_lds(51, 1, 4, 10**9, o2) == 0 and o2[1] == 13312
# round 5: the host-to-host call's mechanisms (page-locked and pageable caller buffers, a capacity that is too small),
# the link probe, the skip-ambiguous run over a large window (prepass + landing area + chunked window bits) from host memory
n = 70_000_000
data = oracle.gen_packed(23, n + 64)
d = torch.from_numpy(data).cuda()
cap = int(n * 0.19)
dev_out = torch.zeros(cap, dtype=torch.int32, device="cuda")
c_dev = b.run_device(d, n, dev_out)
want = dev_out[:c_dev].cpu().numpy().view(np.uint32)
hp, _k1 = sm.pinned_array((len(data),), np.uint8)
hp[:] = data
ppos, _k2 = sm.pinned_array((cap + 3,), np.uint32)
u8p, u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)
ws = sm.default_workspace(0)
for out_m, in_m, chunks, pinned in (("engine", "engine", None, True), ("blit", "engine", "5", True), ("direct", "blit", None, True),
                                    ("blit", "blit", "64", False)):
    os.environ["MM_HOST_OUT"], os.environ["MM_HOST_IN"] = out_m, in_m
    if chunks: os.environ["MM_PIPE_CHUNKS"] = chunks
    else: os.environ.pop("MM_PIPE_CHUNKS", None)
    src = hp if pinned else data
    pos = ppos[1: cap + 1] if pinned else np.zeros(cap, dtype=np.uint32)
    cnt = C.c_uint64()
    sm._check(sm.lib().mm_run_host(b.plan().h, ws.h, src.ctypes.data_as(u8p), 0, n, pos.ctypes.data_as(u32p), None, cap, C.byref(cnt)))
    assert cnt.value == c_dev and np.array_equal(pos[:c_dev], want), (out_m, in_m)
    code = sm.lib().mm_run_host(b.plan().h, ws.h, src.ctypes.data_as(u8p), 0, n, pos.ctypes.data_as(u32p), None, 1000, C.byref(cnt))
    assert code == sm.ERR["CAPACITY"] and cnt.value == c_dev
for k_ in ("MM_HOST_OUT", "MM_HOST_IN", "MM_PIPE_CHUNKS"): os.environ.pop(k_, None)
rates = (C.c_double * 3)()
sm._check(sm.lib().mm_link_probe(ws.h, hp.ctypes.data_as(C.c_void_p), ppos.ctypes.data_as(C.c_void_p), 16 << 20, rates))
assert min(rates) > 0.5
a = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=900_001)].copy()
a[rng.integers(0, len(a), size=len(a) // 120)] = ord("N")
b51 = sm.canonical_minimizers(31, 51)
got = b51.run_skip_ambiguous_windows_once(sm.PackedNSeqVec.from_ascii(a.tobytes()))
packed, amb = oracle.pack_ascii_n(a.tobytes())
assert np.array_equal(np.asarray(got, dtype=np.uint32), oracle.run_skip_ambiguous(packed, amb, len(a), 31, 51))
# round 6: lane-table launches - packed reads above a lane's length, a batch of short contigs in one buffer, the fixed-stride
# entry with per-read lengths (host side: plan, workspace tables, the batch's uploaded starts and lengths)
lens6 = [int(x) for x in rng.integers(0, 9000, 200)] + [70_001, 0, 31]
st6 = np.zeros(len(lens6) + 1, dtype=np.int64); st6[1:] = np.cumsum(lens6)
d6 = sm.generate_device(int(st6[-1]), 41)
h6 = d6.cpu().numpy()
out6 = torch.zeros(int(st6[-1]) // 3 + 8, dtype=torch.int32, device="cuda")
off6 = torch.zeros(len(lens6) + 1, dtype=torch.int64, device="cuda")
c6 = C.c_uint64()
ds6 = torch.from_numpy(st6).cuda()
sm._check(sm.lib().mm_run_packed_reads_device(b.plan().h, ws.h, C.c_void_p(d6.data_ptr()), d6.numel(), 0, len(lens6), C.c_void_p(ds6.data_ptr()),
                                              int(st6[-1]), max(lens6), C.c_void_p(out6.data_ptr()), None, out6.numel(), C.c_void_p(off6.data_ptr()), C.byref(c6)))
assert ws.last_lane_table()
ho6 = off6.cpu().numpy()
for r_ in (0, 57, 200):
    s0 = int(st6[r_])
    assert np.array_equal(out6[ho6[r_]: ho6[r_ + 1]].cpu().numpy().view(np.uint32), oracle.run(h6[s0 // 4:], lens6[r_], k, w, canonical=True, base_offset=s0 % 4)), r_
seqs6 = [d6[int(s_) // 4:] for s_ in st6[:-1]]
bo6 = sm.run_batch_device(b, seqs6, lens6, out6, None, base_offsets=[int(s_) % 4 for s_ in st6[:-1]])
assert ws.last_lane_table() and bo6[-1] == int(c6.value)
print("host paths ok")
"""


@pytest.mark.gpu
def test_host_side_under_asan_ubsan():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    lib = os.path.join(ROOT, "simd-minimizers_amd", "libsimd_minimizers_amd_hostasan.so")
    if not os.path.exists(lib):
        pytest.skip("host-sanitized library not built (make -C simd-minimizers_amd/csrc hostasan)")
    # gcc's AddressSanitizer runtime (ROCm's clang runtime intercepts the HSA allocator for GPU sanitizing, which
    # this pool does not support; the interface the instrumented host objects call is the same)
    rt = _gcc_runtime("libasan.so")
    if not rt:
        pytest.skip("no AddressSanitizer runtime")
    # (libstdc++ preloaded as well: the runtime looks __cxa_throw up when it starts, and python itself does not
    # link the C++ library - without it the first C++ exception inside torch trips an internal check)
    stdcxx = _gcc_runtime("libstdc++.so.6") or _gcc_runtime("libstdc++.so")
    # (and torch's library directory on the search path: dlopen through the runtime's interceptor loses the RUNPATH
    # of the caller, so torch would not find its own lazily loaded libraries)
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    env = dict(os.environ, MM_LIB_PATH=lib, LD_PRELOAD=":".join(x for x in (rt, _gcc_runtime("libubsan.so"), stdcxx) if x),
               LD_LIBRARY_PATH=torch_lib + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""),
               ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0:abort_on_error=1:detect_odr_violation=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", _HOST_SCRIPT, ROOT], env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0 and "host paths ok" in r.stdout, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
