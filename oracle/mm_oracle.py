"""ctypes loader for the CPU oracle (oracle/mm_oracle.c).

TEST INFRASTRUCTURE ONLY: may be imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg, never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

MINIMIZERS, CLOSED_SYNCMERS, OPEN_SYNCMERS = 0, 1, 2
NAIVE, STREAMING = 0, 1


class Hasher(C.Structure):
    _fields_ = [("fw", C.c_uint32 * 4), ("rc", C.c_uint32 * 4), ("rot", C.c_uint32),
                ("canonical", C.c_uint32), ("fw_xor", C.c_uint32), ("rc_xor", C.c_uint32), ("kind", C.c_uint32)]


def build(native: bool = False, out_dir: str | None = None) -> str:
    """Compile the oracle with gcc (a few hundred ms). Returns the .so path."""
    if native:
        out_dir = out_dir or _HERE
        subprocess.run(["make", "-C", _HERE, "native", f"OUT={out_dir}"], check=True,
                       capture_output=True)
        return os.path.join(out_dir, "libmm_oracle_native.so")
    subprocess.run(["make", "-C", _HERE, "all"], check=True, capture_output=True)
    return os.path.join(_HERE, "libmm_oracle.so")


def _bind(lib):
    u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    hp = C.POINTER(Hasher)
    lib.mmo_default_hasher.argtypes = [hp, C.c_int]
    lib.mmo_default_hasher.restype = None
    lib.mmo_mul_hasher.argtypes = [hp, C.c_int]
    lib.mmo_mul_hasher.restype = None
    lib.mmo_antilex_hasher.argtypes = [hp, C.c_uint32, C.c_int]
    lib.mmo_antilex_hasher.restype = None
    lib.mmo_pack_ascii.argtypes = [u8p, C.c_uint64, u8p]
    lib.mmo_pack_ascii.restype = None
    lib.mmo_revcomp_packed.argtypes = [u8p, C.c_uint64, C.c_uint64, u8p]
    lib.mmo_revcomp_packed.restype = None
    lib.mmo_gen_packed.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, u8p]
    lib.mmo_gen_packed.restype = None
    for name in ("mmo_hash_kmers_naive", "mmo_hash_kmers_rolling"):
        f = getattr(lib, name)
        f.argtypes = [u8p, C.c_uint64, C.c_uint64, C.c_uint32, hp, u32p]
        f.restype = C.c_int64
    lib.mmo_window_positions.argtypes = [u8p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, hp,
                                         C.c_int, C.c_int, u32p]
    lib.mmo_window_positions.restype = C.c_int64
    lib.mmo_collect_and_dedup.argtypes = [u32p, C.c_uint64, u32p]
    lib.mmo_collect_and_dedup.restype = C.c_uint64
    lib.mmo_collect_and_dedup_with_index.argtypes = [u32p, C.c_uint64, u32p, u32p]
    lib.mmo_collect_and_dedup_with_index.restype = C.c_uint64
    lib.mmo_collect_syncmers.argtypes = [u32p, C.c_uint64, C.c_uint32, C.c_int, u32p]
    lib.mmo_collect_syncmers.restype = C.c_int64
    lib.mmo_run.argtypes = [u8p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, hp, C.c_int,
                            C.c_int, C.c_int, u32p, u32p, C.c_uint64]
    lib.mmo_run.restype = C.c_int64
    lib.mmo_run_fast.argtypes = [u8p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, hp, C.c_int, C.c_int,
                                 u32p, C.c_uint64]
    lib.mmo_run_fast.restype = C.c_int64
    lib.mmo_fast_lanes.argtypes = []
    lib.mmo_fast_lanes.restype = C.c_int
    lib.mmo_values_u64.argtypes = [u8p, C.c_uint64, C.c_uint32, C.c_int, u32p, C.c_uint64, u64p]
    lib.mmo_values_u64.restype = None
    lib.mmo_values_u128.argtypes = [u8p, C.c_uint64, C.c_uint32, C.c_int, u32p, C.c_uint64, u64p]
    lib.mmo_values_u128.restype = None
    lib.mmo_pack_ascii_n.argtypes = [u8p, C.c_uint64, u8p, u8p]
    lib.mmo_pack_ascii_n.restype = None
    lib.mmo_window_positions_skip_ambiguous.argtypes = [u8p, C.c_uint64, u8p, C.c_uint64, C.c_uint64,
                                                        C.c_uint32, C.c_uint32, hp, C.c_int, u32p]
    lib.mmo_window_positions_skip_ambiguous.restype = C.c_int64
    lib.mmo_collect_and_dedup_skip.argtypes = [u32p, C.c_uint64, C.c_int, C.c_int, u32p]
    lib.mmo_collect_and_dedup_skip.restype = C.c_uint64
    lib.mmo_run_skip_ambiguous.argtypes = [u8p, C.c_uint64, u8p, C.c_uint64, C.c_uint64, C.c_uint32,
                                           C.c_uint32, hp, C.c_int, C.c_int, C.c_int, u32p, C.c_uint64]
    lib.mmo_run_skip_ambiguous.restype = C.c_int64
    lib.mmo_checksum.argtypes = [u32p, C.c_uint64, u64p, u64p]
    lib.mmo_checksum.restype = None
    return lib


_LIB = None


def lib(path: str | None = None):
    global _LIB
    if path is not None:
        return _bind(C.CDLL(path))
    if _LIB is None and os.environ.get("MM_ORACLE_LIB"):
        # (tests/test_sanitizers.py: the same suite against the -fsanitize=address,undefined build)
        _LIB = _bind(C.CDLL(os.environ["MM_ORACLE_LIB"]))
    if _LIB is None:
        so = os.path.join(_HERE, "libmm_oracle.so")
        src_m = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("mm_oracle.c", "mm_oracle.h"))
        if not os.path.exists(so) or os.path.getmtime(so) < src_m:
            build()
        _LIB = _bind(C.CDLL(so))
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def default_hasher(canonical: bool) -> Hasher:
    h = Hasher()
    lib().mmo_default_hasher(C.byref(h), int(canonical))
    return h


def mul_hasher(canonical: bool) -> Hasher:
    """seq-hash MulHasher restated (PARITY UNPINNED, see mm_oracle.h)."""
    h = Hasher()
    lib().mmo_mul_hasher(C.byref(h), int(canonical))
    return h


def antilex_hasher(k: int, canonical: bool) -> Hasher:
    """seq-hash AntiLexHasher restated (PARITY UNPINNED, see mm_oracle.h)."""
    h = Hasher()
    lib().mmo_antilex_hasher(C.byref(h), k, int(canonical))
    return h


def pack_ascii(seq: bytes) -> np.ndarray:
    a = np.frombuffer(bytes(seq), dtype=np.uint8)
    out = np.zeros((len(a) + 3) // 4 + 16, dtype=np.uint8)
    if len(a):
        lib().mmo_pack_ascii(_p(a, C.c_uint8), len(a), _p(out, C.c_uint8))
    return out


def fasta_records(text: bytes) -> list[tuple[int, bytes, bytes]]:
    """FASTA reader restated (the reference's loader uses needletail::parse_fastx_file, a third-party crate that
    is not in the tree, and keeps ``r.seq()`` of every record, bench/src/lib.rs:51-82 - parity unpinned): a record
    starts with '>' at the start of a line; the header runs to the end of that line; the sequence is every
    following line up to the next header line with '\\n' and '\\r' removed; bytes before the first header are
    ignored.  Returns (byte offset of '>', header without '>' and line end, sequence) per record."""
    recs: list[list] = []
    i, n = 0, len(text)
    while i < n:
        j = text.find(b"\n", i)
        end = n if j < 0 else j
        line = text[i:end]
        if line[:1] == b">":
            recs.append([i, line[1:].rstrip(b"\r"), bytearray()])
        elif recs:
            recs[-1][2] += line.replace(b"\r", b"")
        i = end + 1
    return [(p, bytes(h), bytes(q)) for p, h, q in recs]


def fastq_records(text: bytes) -> list[tuple[int, bytes, bytes]]:
    """FASTQ reader restated (needletail::parse_fastx_file reads FASTQ through the same call, bench/src/lib.rs:51-82;
    the crate is not in the tree - parity unpinned): records of FOUR lines - '@' + name, the sequence, '+', the
    qualities; lines end with '\n' or '\r\n' ('\r' dropped), the last line may lack its '\n', blank lines after
    the last record are ignored.  No validation (needletail errors on a malformed record).  Returns (byte offset of
    '@', name without '@' and line end, sequence) per record."""
    recs = []
    lines = []  # (offset, content)
    i, n = 0, len(text)
    while i < n:
        j = text.find(b"\n", i)
        end = n if j < 0 else j
        lines.append((i, text[i:end]))
        i = end + 1
    for r in range(0, len(lines), 4):
        off, head = lines[r]
        # (the first non-'\r' byte of a line 4r starts a record; a blank line 4r starts none)
        stripped = head.replace(b"\r", b"")
        if not stripped:
            continue
        first = off + next(k for k, c in enumerate(head) if c != 13)
        seq = lines[r + 1][1].replace(b"\r", b"") if r + 1 < len(lines) else b""
        recs.append((first, stripped[1:], seq))
    return recs


def gen_packed(seed: int, n: int, first_base: int = 0) -> np.ndarray:
    out = np.zeros((n + 3) // 4 + 16, dtype=np.uint8)
    if n:
        lib().mmo_gen_packed(seed, first_base, n, _p(out, C.c_uint8))
    return out


def revcomp_packed(packed: np.ndarray, n: int, base_offset: int = 0) -> np.ndarray:
    out = np.zeros((n + 3) // 4 + 16, dtype=np.uint8)
    if n:
        lib().mmo_revcomp_packed(_p(packed, C.c_uint8), base_offset, n, _p(out, C.c_uint8))
    return out


def hash_kmers(packed, n, k, hasher, base_offset=0, rolling=False) -> np.ndarray:
    nk = max(0, n - k + 1)
    out = np.zeros(max(nk, 1), dtype=np.uint32)
    f = lib().mmo_hash_kmers_rolling if rolling else lib().mmo_hash_kmers_naive
    r = f(_p(packed, C.c_uint8), base_offset, n, k, C.byref(hasher), _p(out, C.c_uint32))
    assert r == nk, r
    return out[:nk]


def window_positions(packed, n, k, w, hasher, canonical, flavour=STREAMING, base_offset=0):
    nw = max(0, n - (k + w - 1) + 1)
    out = np.zeros(max(nw, 1), dtype=np.uint32)
    r = lib().mmo_window_positions(_p(packed, C.c_uint8), base_offset, n, k, w, C.byref(hasher),
                                   int(canonical), flavour, _p(out, C.c_uint32))
    if r < 0:
        raise ValueError(f"oracle error {r}")
    return out[:r]


def run(packed, n, k, w, hasher=None, canonical=False, mode=MINIMIZERS, flavour=STREAMING,
        base_offset=0, super_kmers=False):
    """Whole path. Returns positions (and super-k-mer indices if requested)."""
    if hasher is None:
        hasher = default_hasher(canonical)
    cap = max(1, n)
    pos = np.zeros(cap, dtype=np.uint32)
    sk = np.zeros(cap, dtype=np.uint32) if super_kmers else None
    r = lib().mmo_run(_p(packed, C.c_uint8), base_offset, n, k, w, C.byref(hasher), int(canonical),
                      mode, flavour, _p(pos, C.c_uint32),
                      _p(sk, C.c_uint32) if super_kmers else None, cap)
    if r < 0:
        raise ValueError(f"oracle error {r}")
    if super_kmers:
        return pos[:r].copy(), sk[:r].copy()
    return pos[:r].copy()


def run_fast(packed, n, k, w, canonical=False, threads=1, hasher=None, base_offset=0, lib_=None, cap=None):
    """One-pass (optionally threaded) port used for CPU timing; equals run(..., mode=MINIMIZERS)."""
    if hasher is None:
        hasher = default_hasher(canonical)
    cap = max(1, n) if cap is None else cap
    pos = np.zeros(cap, dtype=np.uint32)
    r = (lib_ or lib()).mmo_run_fast(_p(packed, C.c_uint8), base_offset, n, k, w, C.byref(hasher),
                                     int(canonical), threads, _p(pos, C.c_uint32), cap)
    if r < 0:
        raise ValueError(f"oracle error {r}")
    return pos[:r].copy()


def run_threads(packed, n, k, w, canonical=False, mode=MINIMIZERS, super_kmers=False, threads=None, hasher=None,
                base_offset=0, chunk_windows=1 << 22):
    """The whole path over a LONG sequence on all host cores (test infrastructure for the full-size element-by-element
    checks): minimizer positions without indices through the threaded AVX2 port (run_fast); every other flavour - syncmer
    modes, super-k-mer indices - through the streaming restatement (run) over window chunks on a thread pool (ctypes
    releases the GIL), joined with the reference's own rule for its lanes: a chunk's first position is dropped when it
    equals the previous chunk's last (src/collect.rs:265-271); syncmers have no such rule (src/syncmers.rs:166-169)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    threads = threads or os.cpu_count() or 1
    if hasher is None:
        hasher = default_hasher(canonical)
    l = k + w - 1
    if n < l:
        e = np.zeros(0, dtype=np.uint32)
        return (e, e.copy()) if super_kmers else e
    if mode == MINIMIZERS and not super_kmers:
        return run_fast(packed, n, k, w, canonical=canonical, threads=threads, hasher=hasher, base_offset=base_offset)
    nw = n - l + 1
    cuts = list(range(0, nw, chunk_windows)) + [nw]

    def piece(i):
        a, e = cuts[i], cuts[i + 1]
        r = run(packed, e - a + l - 1, k, w, hasher=hasher, canonical=canonical, mode=mode, base_offset=base_offset + a,
                super_kmers=super_kmers)
        if super_kmers:
            return r[0] + np.uint32(a), r[1] + np.uint32(a)
        return r + np.uint32(a)
    with ThreadPoolExecutor(max_workers=threads) as ex:
        parts = list(ex.map(piece, range(len(cuts) - 1)))
    pos, sks, last = [], [], None
    for pt in parts:
        p, q = (pt if super_kmers else (pt, None))
        if mode == MINIMIZERS and last is not None and len(p) and p[0] == last:
            p = p[1:]
            q = q[1:] if q is not None else None
        if len(p):
            last = p[-1]
        pos.append(p)
        if q is not None:
            sks.append(q)
    out = np.concatenate(pos) if pos else np.zeros(0, dtype=np.uint32)
    if super_kmers:
        return out, (np.concatenate(sks) if sks else np.zeros(0, dtype=np.uint32))
    return out


def values_u64(packed, length, positions, canonical, base_offset=0) -> np.ndarray:
    positions = np.ascontiguousarray(positions, dtype=np.uint32)
    out = np.zeros(max(1, len(positions)), dtype=np.uint64)
    lib().mmo_values_u64(_p(packed, C.c_uint8), base_offset, length, int(canonical),
                         _p(positions, C.c_uint32), len(positions), _p(out, C.c_uint64))
    return out[:len(positions)]


def values_u128(packed, length, positions, canonical, base_offset=0) -> np.ndarray:
    """Returns an (n, 2) uint64 array: low and high halves."""
    positions = np.ascontiguousarray(positions, dtype=np.uint32)
    out = np.zeros(max(1, 2 * len(positions)), dtype=np.uint64)
    lib().mmo_values_u128(_p(packed, C.c_uint8), base_offset, length, int(canonical),
                          _p(positions, C.c_uint32), len(positions), _p(out, C.c_uint64))
    return out[:2 * len(positions)].reshape(-1, 2)


SKIPPED = 0xFFFFFFFE


def pack_ascii_n(seq: bytes):
    """PackedNSeqVec::from_ascii: (packed 2-bit codes, ambiguity bits)."""
    a = np.frombuffer(seq, dtype=np.uint8)
    packed = np.zeros((len(a) + 3) // 4 + 16, dtype=np.uint8)
    amb = np.zeros((len(a) + 7) // 8 + 16, dtype=np.uint8)
    if len(a):
        lib().mmo_pack_ascii_n(_p(np.ascontiguousarray(a), C.c_uint8), len(a), _p(packed, C.c_uint8),
                               _p(amb, C.c_uint8))
    return packed, amb


def window_positions_skip_ambiguous(packed, amb, n, k, w, hasher=None, canonical=True, base_offset=0,
                                    amb_offset=0):
    if hasher is None:
        hasher = default_hasher(canonical)
    l = k + w - 1
    nw = max(0, n - l + 1)
    out = np.zeros(max(1, nw), dtype=np.uint32)
    r = lib().mmo_window_positions_skip_ambiguous(_p(packed, C.c_uint8), base_offset, _p(amb, C.c_uint8),
                                                  amb_offset, n, k, w, C.byref(hasher), int(canonical),
                                                  _p(out, C.c_uint32))
    if r < 0:
        raise ValueError(f"oracle error {r}")
    return out[:r].copy()


def collect_and_dedup_skip(stream, skip_max: bool, rule: int = 0):
    stream = np.ascontiguousarray(stream, dtype=np.uint32)
    out = np.zeros(max(1, len(stream)), dtype=np.uint32)
    m = lib().mmo_collect_and_dedup_skip(_p(stream, C.c_uint32), len(stream), int(skip_max), rule,
                                         _p(out, C.c_uint32))
    return out[:m].copy()


def run_skip_ambiguous(packed, amb, n, k, w, hasher=None, canonical=True, mode=MINIMIZERS, rule=0,
                       base_offset=0, amb_offset=0):
    """Builder::run_skip_ambiguous_windows (src/lib.rs:451-496)."""
    if hasher is None:
        hasher = default_hasher(canonical)
    cap = max(1, n)
    pos = np.zeros(cap, dtype=np.uint32)
    r = lib().mmo_run_skip_ambiguous(_p(packed, C.c_uint8), base_offset, _p(amb, C.c_uint8), amb_offset, n,
                                     k, w, C.byref(hasher), int(canonical), mode, rule,
                                     _p(pos, C.c_uint32), cap)
    if r < 0:
        raise ValueError(f"oracle error {r}")
    return pos[:r].copy()


def checksum(v) -> tuple[int, int]:
    v = np.ascontiguousarray(v, dtype=np.uint32)
    a, b = C.c_uint64(), C.c_uint64()
    lib().mmo_checksum(_p(v, C.c_uint32), len(v), C.byref(a), C.byref(b))
    return a.value, b.value
