"""simd_minimizers_amd — Python harness over the C ABI of the MI355X minimizer engine.

The product is the HIP library ``libsimd_minimizers_amd.so`` (``csrc/``, C ABI in
``include/simd_minimizers_amd.h``).  This module is the thin host-side mirror of the
reference's builder API (rust-seq/simd-minimizers ``src/lib.rs:225-654``): same constructor
names, same argument meaning, same error conditions (the reference's ``assert!``s surface as
``MinimizerError``).  It never computes on the CPU: without the HIP library or without a GPU
every ``run`` raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MM_LIB_PATH: tests/test_sanitizers.py loads the build whose HOST objects carry AddressSanitizer + UBSan)
LIB_PATH = os.environ.get("MM_LIB_PATH") or os.path.join(_HERE, "libsimd_minimizers_amd.so")

MM_MINIMIZERS, MM_CLOSED_SYNCMERS, MM_OPEN_SYNCMERS = 0, 1, 2
PATH_FUSED, PATH_GENERIC, PATH_SPLIT = 1, 2, 3
U64_MAX = (1 << 64) - 1

ERR = {
    "W_ZERO": -1, "W_TOO_LARGE": -2, "LEN_TOO_LARGE": -3, "EVEN_L": -4,
    "HASHER_NOT_CANONICAL": -5, "OPEN_EVEN_W": -6, "K_ZERO": -7, "CAPACITY": -8, "BAD_MODE": -9,
    "NULL": -10, "VALUE_LEN": -11, "FORMAT": -12, "NO_DEVICE": -20, "HIP": -21, "ALLOC": -22, "ORDER": -23,
}


class MinimizerError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[{code}] {msg}")
        self.code = code


class Hasher(C.Structure):
    """seq-hash NtHasher tables (see include/simd_minimizers_amd.h: mm_hasher_t)."""
    _fields_ = [("fw", C.c_uint32 * 4), ("rc", C.c_uint32 * 4), ("rot", C.c_uint32),
                ("canonical", C.c_uint32), ("fw_xor", C.c_uint32), ("rc_xor", C.c_uint32), ("kind", C.c_uint32)]

    @staticmethod
    def from_tables(fw, rc, rot=7, canonical=True) -> "Hasher":
        h = Hasher()
        for i in range(4):
            h.fw[i] = int(fw[i]) & 0xFFFFFFFF
            h.rc[i] = int(rc[i]) & 0xFFFFFFFF
        h.rot = rot
        h.canonical = 1 if canonical else 0
        h.fw_xor = h.rc_xor = h.kind = 0
        return h

    def is_canonical(self) -> bool:
        return bool(self.canonical)


_lib = None


def lib():
    """Load the HIP library; fail loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        try:
            # PyTorch-ROCm bundles its own libamdhip64.so.7; load it first so that the engine and
            # torch share ONE HIP runtime (same SONAME) whatever the import order.
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        u8p, u32p, u64p, vp = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.c_void_p
        L.mm_strerror.restype = C.c_char_p
        L.mm_strerror.argtypes = [C.c_int]
        L.mm_last_error.restype = C.c_char_p
        L.mm_device_count.restype = C.c_int
        L.mm_default_hasher.argtypes = [C.POINTER(Hasher), C.c_int]
        L.mm_mul_hasher.argtypes = [C.POINTER(Hasher), C.c_int]
        L.mm_antilex_hasher.argtypes = [C.POINTER(Hasher), C.c_uint32, C.c_int]
        L.mm_plan_create.argtypes = [C.POINTER(vp), C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                     C.POINTER(Hasher)]
        L.mm_plan_destroy.argtypes = [vp]
        L.mm_plan_destroy.restype = None
        L.mm_plan_value_len.argtypes = [vp]
        L.mm_plan_value_len.restype = C.c_uint32
        L.mm_workspace_create.argtypes = [C.POINTER(vp), C.c_int, vp]
        L.mm_workspace_destroy.argtypes = [vp]
        L.mm_workspace_destroy.restype = None
        L.mm_workspace_sync.argtypes = [vp]
        L.mm_workspace_check.argtypes = [vp]
        L.mm_device_group_create.argtypes = [C.POINTER(vp), C.POINTER(C.c_int), C.c_int]
        L.mm_device_group_destroy.argtypes = [vp]
        L.mm_device_group_destroy.restype = None
        L.mm_device_group_size.argtypes = [vp]
        L.mm_run_sharded_host.argtypes = [vp, vp, u8p, C.c_uint64, C.c_uint64, u32p, u32p, C.c_uint64, u64p]
        L.mm_run_batch_sharded_host.argtypes = [vp, vp, C.c_uint64, C.POINTER(u8p), u64p, u64p, u32p, u32p, C.c_uint64, u64p]
        if hasattr(L, "mm_run_sharded_device"):  # (absent from the round-3 library kept for A/B runs under tools/ab/)
            L.mm_device_group_upload.argtypes = [vp, u8p, C.c_uint64]
            if hasattr(L, "mm_device_group_upload_range"):
                L.mm_device_group_upload_range.argtypes = [vp, u8p, C.c_uint64, C.c_uint64, C.c_uint64]
            L.mm_device_group_adopt.argtypes = [vp, C.POINTER(vp), C.c_uint64]
            L.mm_run_sharded_device.argtypes = [vp, vp, C.c_uint64, C.c_uint64, C.c_int, u64p, u64p]
            L.mm_device_group_result.argtypes = [vp, C.c_int, C.POINTER(u32p), C.POINTER(u32p), u64p, u64p, u64p]
            L.mm_device_group_gather.argtypes = [vp, C.c_int, vp, vp, C.c_uint64, u64p]
            L.mm_device_group_upload_batch.argtypes = [vp, C.c_uint64, C.POINTER(u8p), u64p]
            L.mm_run_batch_sharded_device.argtypes = [vp, vp, u64p, u64p, C.c_int, u64p, u64p]
            L.mm_device_group_batch_result.argtypes = [vp, C.c_uint64, C.POINTER(C.c_int), C.POINTER(u32p), C.POINTER(u32p), u64p]
            L.mm_device_group_gather_batch.argtypes = [vp, C.c_int, vp, vp, C.c_uint64, u64p]
            L.mm_run_packed_reads_device_async.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint64, vp, C.c_uint64,
                                                           C.c_uint32, vp, vp, C.c_uint64, vp, vp]
            L.mm_run_packed_reads_device.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint64, vp, C.c_uint64,
                                                     C.c_uint32, vp, vp, C.c_uint64, vp, u64p]
            L.mm_run_packed_reads_host.argtypes = [vp, vp, u8p, C.c_uint64, u64p, C.c_uint32, u32p, u32p, C.c_uint64, u64p, u64p]
            L.mm_debug_launch_plan.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_uint64, u64p, u64p, u32p, u32p, u32p,
                                               C.c_uint64, u64p]
        L.mm_clock_probe_begin.argtypes = [vp, C.c_uint64]
        L.mm_clock_probe_end.argtypes = [vp, C.POINTER(C.c_double)]
        L.mm_link_probe.argtypes = [vp, vp, vp, C.c_uint64, C.POINTER(C.c_double)]
        L.mm_debug_launch_lds.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_uint64, u64p]
        L.mm_debug_lane_plan.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, u64p]
        L.mm_debug_last_lane_table.argtypes = [vp, u32p, C.c_uint64, u64p]
        L.mm_fused_overread_bytes.argtypes = []
        L.mm_fused_overread_bytes.restype = C.c_uint64
        L.mm_workspace_force_generic.argtypes = [vp, C.c_int]
        L.mm_workspace_set_blocks_per_lane.argtypes = [vp, C.c_uint32]
        L.mm_workspace_enable_timing.argtypes = [vp, C.c_int]
        L.mm_workspace_kernel_time.argtypes = [vp, C.POINTER(C.c_double), u64p, C.c_int]
        L.mm_workspace_last_path.argtypes = [vp]
        L.mm_workspace_last_lane_table.argtypes = [vp]
        if hasattr(L, "mm_prebuilt_window_sizes"):  # (absent from the round-3 library kept for A/B runs under tools/ab/)
            L.mm_prebuilt_window_sizes.argtypes = [C.c_int, C.c_int, u32p, C.c_int]
            L.mm_prebuilt_window_sizes.restype = C.c_int
        L.mm_run_device_async.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64,
                                          C.c_uint64, vp, vp, C.c_uint64, vp]
        L.mm_run_device.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64,
                                    C.c_uint64, vp, vp, C.c_uint64, u64p]
        L.mm_run_host.argtypes = [vp, vp, u8p, C.c_uint64, C.c_uint64, u32p, u32p, C.c_uint64, u64p]
        L.mm_run_host_ascii.argtypes = [vp, vp, u8p, C.c_uint64, u32p, u32p, C.c_uint64, u64p]
        L.mm_values_u64_device_async.argtypes = [vp, vp, C.c_uint64, C.c_uint64, C.c_uint64,
                                                 C.c_uint32, C.c_int, vp, C.c_uint64, vp]
        L.mm_values_u64_host.argtypes = [vp, u8p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, u32p,
                                         C.c_uint64, u64p]
        L.mm_values_u128_device_async.argtypes = [vp, vp, C.c_uint64, C.c_uint64, C.c_uint64,
                                                  C.c_uint32, C.c_int, vp, C.c_uint64, vp]
        L.mm_values_u128_host.argtypes = [vp, u8p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, u32p,
                                          C.c_uint64, u64p]
        L.mm_run_batch_device.argtypes = [vp, vp, C.c_uint64, C.POINTER(vp), u64p, u64p, u64p, vp, vp,
                                          C.c_uint64, u64p]
        reads_args = [vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, vp, vp,
                      C.c_uint64, vp]
        L.mm_run_reads_device_async.argtypes = reads_args + [vp]
        L.mm_run_reads_device.argtypes = reads_args + [u64p]
        reads_sk_args = [vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, vp, vp, vp,
                         C.c_uint64, vp]
        L.mm_run_reads_superkmers_device_async.argtypes = reads_sk_args + [vp]
        L.mm_run_reads_superkmers_device.argtypes = reads_sk_args + [u64p]
        skip_args = [vp, vp, vp, C.c_uint64, C.c_uint64, vp, C.c_uint64, C.c_uint64, C.c_uint64,
                     C.c_uint64, C.c_uint64, vp, C.c_uint64]
        L.mm_run_skip_ambiguous_device_async.argtypes = skip_args + [vp]
        L.mm_run_skip_ambiguous_device.argtypes = skip_args + [u64p]
        L.mm_run_skip_ambiguous_host.argtypes = [vp, vp, u8p, C.c_uint64, u8p, C.c_uint64, C.c_uint64, u32p,
                                                 C.c_uint64, u64p]
        L.mm_run_skip_ambiguous_host_ascii.argtypes = [vp, vp, u8p, C.c_uint64, u32p, C.c_uint64, u64p]
        reads_skip_args = [vp, vp, vp, C.c_uint64, C.c_uint64, vp, C.c_uint64, C.c_uint64, C.c_uint64,
                           C.c_uint32, C.c_uint32, vp, vp, C.c_uint64, vp]
        L.mm_run_reads_skip_ambiguous_device_async.argtypes = reads_skip_args + [vp]
        L.mm_run_reads_skip_ambiguous_device.argtypes = reads_skip_args + [u64p]
        L.mm_pack_ascii_n_device_async.argtypes = [vp, vp, C.c_uint64, vp, vp]
        L.mm_host_alloc.argtypes = [C.POINTER(vp), C.c_uint64]
        L.mm_host_free.argtypes = [vp]
        L.mm_host_free.restype = None
        L.mm_pack_ascii_device_async.argtypes = [vp, vp, C.c_uint64, vp]
        L.mm_generate_device_async.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, vp]
        fasta_args = [vp, vp, C.c_uint64, vp, C.c_uint64, vp, vp, C.c_uint64, vp]
        L.mm_fasta_pack_device_async.argtypes = fasta_args
        L.mm_fasta_pack_device.argtypes = fasta_args + [u64p]
        if hasattr(L, "mm_fastq_pack_device_async"):
            L.mm_fastq_pack_device_async.argtypes = fasta_args
        _lib = L
    return _lib


EXPORTED_SYMBOLS = [
    "mm_strerror", "mm_last_error", "mm_device_count", "mm_default_hasher", "mm_mul_hasher", "mm_antilex_hasher",
    "mm_plan_create",
    "mm_plan_destroy", "mm_plan_value_len", "mm_workspace_create", "mm_workspace_destroy",
    "mm_workspace_sync", "mm_workspace_check", "mm_workspace_force_generic", "mm_workspace_set_blocks_per_lane",
    "mm_workspace_enable_timing", "mm_workspace_kernel_time", "mm_workspace_last_path", "mm_workspace_last_lane_table", "mm_prebuilt_window_sizes",
    "mm_run_device_async", "mm_run_device", "mm_run_host", "mm_run_host_ascii",
    "mm_values_u64_device_async", "mm_values_u64_host", "mm_values_u128_device_async",
    "mm_values_u128_host", "mm_run_batch_device", "mm_run_reads_device_async", "mm_run_reads_device",
    "mm_run_reads_superkmers_device_async", "mm_run_reads_superkmers_device",
    "mm_run_skip_ambiguous_device_async", "mm_run_skip_ambiguous_device", "mm_run_skip_ambiguous_host",
    "mm_run_skip_ambiguous_host_ascii", "mm_run_reads_skip_ambiguous_device_async",
    "mm_run_reads_skip_ambiguous_device", "mm_pack_ascii_n_device_async", "mm_pack_ascii_device_async",
    "mm_host_alloc", "mm_host_free",
    "mm_generate_device_async", "mm_fasta_pack_device_async", "mm_fasta_pack_device", "mm_fastq_pack_device_async",
    "mm_clock_probe_begin", "mm_clock_probe_end", "mm_link_probe", "mm_fused_overread_bytes",
    "mm_device_group_create", "mm_device_group_destroy", "mm_device_group_size", "mm_run_sharded_host",
    "mm_run_batch_sharded_host",
    "mm_device_group_upload", "mm_device_group_upload_range", "mm_device_group_adopt", "mm_run_sharded_device", "mm_device_group_result",
    "mm_device_group_gather",
    "mm_device_group_upload_batch", "mm_run_batch_sharded_device", "mm_device_group_batch_result",
    "mm_device_group_gather_batch", "mm_debug_launch_plan", "mm_debug_launch_lds", "mm_debug_lane_plan",
    "mm_debug_last_lane_table",
    "mm_run_packed_reads_device_async", "mm_run_packed_reads_device", "mm_run_packed_reads_host",
]


def prebuilt_window_sizes(canonical: bool, reads: bool = False) -> list:
    """Window sizes with a prebuilt fused kernel (mm_prebuilt_window_sizes): what the tests sweep."""
    L = lib()
    n = L.mm_prebuilt_window_sizes(int(canonical), int(reads), None, 0)
    buf = (C.c_uint32 * max(1, n))()
    L.mm_prebuilt_window_sizes(int(canonical), int(reads), buf, n)
    return [int(x) for x in buf[:n]]


def _check(code: int):
    if code != 0:
        L = lib()
        msg = L.mm_strerror(code).decode()
        if code in (ERR["HIP"], ERR["ALLOC"], ERR["NO_DEVICE"], ERR["ORDER"]):
            msg += ": " + L.mm_last_error().decode()
        raise MinimizerError(code, msg)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


# ------------------------------------------------------------------ sequences


class PackedSeq:
    """packed-seq ``PackedSeq``: a borrowed view (bytes, base offset, length)."""

    def __init__(self, data: np.ndarray, offset: int, length: int):
        self.data, self.offset, self.length = data, offset, length

    def __len__(self):
        return self.length

    def slice(self, start: int, end: int) -> "PackedSeq":
        assert 0 <= start <= end <= self.length
        return PackedSeq(self.data, self.offset + start, end - start)

    def as_slice(self) -> "PackedSeq":
        return self

    def codes(self) -> np.ndarray:
        i = np.arange(self.offset, self.offset + self.length, dtype=np.int64)
        return ((self.data[i >> 2] >> (2 * (i & 3)).astype(np.uint8)) & 3).astype(np.uint8)

    def to_revcomp(self) -> "PackedSeqVec":
        return PackedSeqVec.from_codes((self.codes()[::-1] ^ 2).astype(np.uint8))


class PackedSeqVec(PackedSeq):
    """packed-seq ``PackedSeqVec``: owns 2-bit packed bases (A0 C1 T2 G3, 4 per byte)."""

    def __init__(self, data: np.ndarray, length: int):
        super().__init__(data, 0, length)

    @staticmethod
    def from_codes(codes: np.ndarray) -> "PackedSeqVec":
        n = len(codes)
        pad = np.zeros((n + 3) // 4 * 4, dtype=np.uint8)
        pad[:n] = codes
        q = pad.reshape(-1, 4)
        data = (q[:, 0] | (q[:, 1] << 2) | (q[:, 2] << 4) | (q[:, 3] << 6)).astype(np.uint8)
        return PackedSeqVec(np.concatenate([data, np.zeros(16, dtype=np.uint8)]), n)

    @staticmethod
    def from_ascii(seq: bytes) -> "PackedSeqVec":
        a = np.frombuffer(bytes(seq), dtype=np.uint8)
        return PackedSeqVec.from_codes(((a >> 1) & 3).astype(np.uint8))

    @staticmethod
    def random(n: int, seed: int = 0) -> "PackedSeqVec":
        rng = np.random.default_rng(seed)
        return PackedSeqVec.from_codes(rng.integers(0, 4, size=n, dtype=np.uint8))


class PackedNSeq:
    """packed-seq ``PackedNSeq``: a PackedSeq view plus one ambiguity bit per base (bit i%8 of byte
    i/8 of ``amb``, with its own bit offset)."""

    def __init__(self, seq: PackedSeq, amb: np.ndarray, amb_offset: int = 0):
        self.seq, self.amb, self.amb_offset = seq, amb, amb_offset

    def __len__(self):
        return len(self.seq)

    def as_slice(self) -> "PackedNSeq":
        return self

    def slice(self, start: int, end: int) -> "PackedNSeq":
        return PackedNSeq(self.seq.slice(start, end), self.amb, self.amb_offset + start)


class PackedNSeqVec(PackedNSeq):
    """packed-seq ``PackedNSeqVec`` (call site src/test.rs:436)."""

    @staticmethod
    def from_ascii(seq: bytes) -> "PackedNSeqVec":
        a = np.frombuffer(bytes(seq), dtype=np.uint8)
        up = a & 0xDF
        isn = ~((up == 65) | (up == 67) | (up == 71) | (up == 84))
        amb = np.concatenate([np.packbits(isn, bitorder="little"), np.zeros(16, dtype=np.uint8)])
        return PackedNSeqVec(PackedSeqVec.from_ascii(seq), amb, 0)


class AsciiSeq:
    """packed-seq ``AsciiSeq``: ACTG/actg characters, mapped with (c >> 1) & 3."""

    def __init__(self, seq: bytes):
        self.seq = bytes(seq)

    def __len__(self):
        return len(self.seq)

    def slice(self, start: int, end: int) -> "AsciiSeq":
        return AsciiSeq(self.seq[start:end])


# ------------------------------------------------------------------ workspace


class Workspace:
    """Device scratch + stream (the reference's thread-local CACHE, src/lib.rs:217-219)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        L = lib()
        if L.mm_device_count() <= 0:
            raise MinimizerError(ERR["NO_DEVICE"], "no HIP device (this engine has no CPU fallback)")
        h = C.c_void_p()
        _check(L.mm_workspace_create(C.byref(h), device, C.c_void_p(stream)))
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            lib().mm_workspace_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _check(lib().mm_workspace_sync(self.h))

    def check(self):
        """Completion status of the asynchronous runs since the last check (mm_workspace_check):
        raises MinimizerError(ORDER) if one of them has to be repeated."""
        _check(lib().mm_workspace_check(self.h))

    def clock_probe_begin(self, duration_us: int):
        """Start sampling the shader clock for ``duration_us`` on a stream of the workspace's own (diagnostics)."""
        _check(lib().mm_clock_probe_begin(self.h, int(duration_us)))

    def clock_probe_end(self) -> float:
        """Mean shader clock in GHz over the sampled span."""
        g = C.c_double()
        _check(lib().mm_clock_probe_end(self.h, C.byref(g)))
        return g.value

    def force_generic(self, on: bool):
        _check(lib().mm_workspace_force_generic(self.h, int(on)))

    def set_blocks_per_lane(self, nblk: int):
        _check(lib().mm_workspace_set_blocks_per_lane(self.h, nblk))

    def enable_timing(self, on: bool):
        _check(lib().mm_workspace_enable_timing(self.h, int(on)))

    def kernel_time(self, reset: bool = True) -> tuple[float, int]:
        ms, n = C.c_double(), C.c_uint64()
        _check(lib().mm_workspace_kernel_time(self.h, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    def last_path(self) -> int:
        return lib().mm_workspace_last_path(self.h)

    def last_lane_table(self) -> bool:
        """The last reads / batch run was ONE lane-table launch (``mm_workspace_last_lane_table``)."""
        return bool(lib().mm_workspace_last_lane_table(self.h))


class DeviceGroup:
    """``mm_device_group_t``: one workspace per listed device (a device may be listed more than once); the
    several-device calls fan one sequence (window ranges, exact seam) or a set of sequences (greedy placement)
    over them - the reference's rayon loop over contigs (bench/src/bin/paper.rs:442-459) behind the C ABI."""

    def __init__(self, devices):
        self.h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        _check(lib().mm_device_group_create(C.byref(self.h), arr, len(devices)))
        self._batch_n = None  # sequences of the resident batch (upload_batch), None: no batch uploaded yet

    def __len__(self):
        return lib().mm_device_group_size(self.h)

    def close(self):
        if self.h:
            lib().mm_device_group_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, builder: "Builder", packed: np.ndarray, n_bases: int, base_offset: int = 0, capacity=None, out=None):
        """One host PackedSeq over all entries (``mm_run_sharded_host``): (positions, super-k-mer indices or None).
        ``out``: a caller-owned uint32 array to receive the positions (e.g. page-locked, ``pinned_array``) instead of
        a fresh one per call."""
        plan = builder.plan()
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        l = builder.k + builder.w - 1
        cap = max(1, n_bases - l + 1) if capacity is None else capacity
        if out is not None:
            # (the library writes uint32 positions straight into it)
            if not (isinstance(out, np.ndarray) and out.dtype == np.uint32 and out.ndim == 1 and out.flags.c_contiguous
                    and out.flags.writeable):
                raise ValueError("out must be a writeable, C-contiguous, one-dimensional uint32 array")
            cap = min(cap, out.size) if capacity is not None else out.size
        pos = out if out is not None else np.empty(cap, dtype=np.uint32)
        sk = np.empty(cap, dtype=np.uint32) if builder._sk is not None else None
        cnt = C.c_uint64()
        code = lib().mm_run_sharded_host(plan.h, self.h, packed.ctypes.data_as(C.POINTER(C.c_uint8)), base_offset, n_bases,
                                         pos.ctypes.data_as(C.POINTER(C.c_uint32)),
                                         sk.ctypes.data_as(C.POINTER(C.c_uint32)) if sk is not None else None, cap,
                                         C.byref(cnt))
        if code == ERR["CAPACITY"]:
            raise MinimizerError(code, f"output capacity {cap} < {cnt.value}")
        _check(code)
        return pos[:cnt.value], (sk[:cnt.value] if sk is not None else None)

    # ---- device-resident shards (mm_device_group_upload / _adopt, mm_run_sharded_device, _result, _gather)
    def upload(self, packed: np.ndarray):
        """The packed sequence to every device of the group, once (kept until the next upload / adopt)."""
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        _check(lib().mm_device_group_upload(self.h, packed.ctypes.data_as(C.POINTER(C.c_uint8)), packed.size))

    def upload_range(self, packed: np.ndarray, n_bases: int, base_offset: int = 0):
        """``mm_device_group_upload_range``: every entry receives only the bytes its share of an N-way split of
        (base_offset, n_bases) reads (+ a halo): the sequence crosses the host link once in total."""
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        _check(lib().mm_device_group_upload_range(self.h, packed.ctypes.data_as(C.POINTER(C.c_uint8)), packed.size,
                                                  base_offset, n_bases))

    def adopt(self, device_tensors):
        """Device buffers the caller already holds (one per entry, on that entry's device, same bytes)."""
        self._adopted = list(device_tensors)  # keep them alive
        ptrs = (C.c_void_p * len(self._adopted))(*[t.data_ptr() for t in self._adopted])
        _check(lib().mm_device_group_adopt(self.h, ptrs, int(self._adopted[0].numel())))

    def run_device(self, builder: "Builder", n_bases: int, base_offset: int = 0):
        """``mm_run_sharded_device``: one asynchronous launch per entry over its window range of the resident
        sequence; the positions stay on the devices.  Returns the per-entry counts."""
        n = len(self)
        counts = (C.c_uint64 * n)()
        total = C.c_uint64()
        code = lib().mm_run_sharded_device(builder.plan().h, self.h, base_offset, n_bases,
                                           1 if builder._sk is not None else 0, counts, C.byref(total))
        if code == ERR["NULL"]:  # (no resident sequence, or one whose resident ranges do not cover this run: the library says which)
            raise MinimizerError(code, lib().mm_strerror(code).decode() + ": " + lib().mm_last_error().decode())
        _check(code)
        return [int(c) for c in counts]

    def result(self, entry: int):
        """(device address of the positions, of the super-k-mer indices or 0, count, win_begin, win_end) of an entry."""
        dp, ds = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)()
        cnt, wb, we = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(lib().mm_device_group_result(self.h, entry, C.byref(dp), C.byref(ds), C.byref(cnt), C.byref(wb), C.byref(we)))
        addr = lambda p: C.cast(p, C.c_void_p).value or 0
        return addr(dp), addr(ds), int(cnt.value), int(wb.value), int(we.value)

    def gather(self, root: int, d_dst_pos, d_dst_sk=None):
        """``mm_device_group_gather``: the shards, dense and in window order, into device tensors on the root entry's
        device (device-to-device copies; xGMI between the GPUs of a node).  Returns the number of positions."""
        total = C.c_uint64()
        code = lib().mm_device_group_gather(self.h, root, C.c_void_p(d_dst_pos.data_ptr()),
                                            C.c_void_p(d_dst_sk.data_ptr()) if d_dst_sk is not None else None,
                                            int(d_dst_pos.numel()), C.byref(total))
        if code == ERR["CAPACITY"]:
            raise MinimizerError(code, f"gather capacity {int(d_dst_pos.numel())} < {total.value}")
        _check(code)
        return int(total.value)

    # ---- device-resident batches (mm_device_group_upload_batch, mm_run_batch_sharded_device, _gather_batch)
    def upload_batch(self, seqs):
        """Independent host sequences placed greedily on the entries, each copied to its entry's device only."""
        keep = [np.ascontiguousarray(a, dtype=np.uint8) for a in seqs]
        n = len(keep)
        ptrs = (C.POINTER(C.c_uint8) * max(n, 1))(*[a.ctypes.data_as(C.POINTER(C.c_uint8)) for a in keep])
        nbytes = (C.c_uint64 * max(n, 1))(*[a.size for a in keep])
        _check(lib().mm_device_group_upload_batch(self.h, n, ptrs, nbytes))
        self._batch_n = n

    def _batch_count(self) -> int:
        if self._batch_n is None:
            raise MinimizerError(ERR["NULL"], "no resident batch: call upload_batch() first")
        return self._batch_n

    def run_batch_device(self, builder: "Builder", n_bases, base_offsets=None):
        """One batch launch per entry over its resident sequences; returns the per-sequence counts (input order)."""
        n = self._batch_count()
        n_bases = list(n_bases)
        base_offsets = list(base_offsets) if base_offsets is not None else [0] * n
        if len(n_bases) != n or len(base_offsets) != n:  # (a short list used to be zero-padded silently)
            raise MinimizerError(ERR["NULL"], f"run_batch_device: {n} sequences are resident (upload_batch), got "
                                              f"{len(n_bases)} lengths and {len(base_offsets)} base offsets")
        lens = (C.c_uint64 * max(n, 1))(*n_bases)
        offs = (C.c_uint64 * max(n, 1))(*base_offsets)
        counts = (C.c_uint64 * max(n, 1))()
        total = C.c_uint64()
        _check(lib().mm_run_batch_sharded_device(builder.plan().h, self.h, offs, lens, 1 if builder._sk is not None else 0,
                                                 counts, C.byref(total)))
        return [int(c) for c in counts[:n]]

    def gather_batch(self, root: int, d_dst_pos, d_dst_sk=None):
        """All sequences' positions, input order, into device tensors on the root entry's device; returns the offsets."""
        n = self._batch_count()
        offs = (C.c_uint64 * (n + 1))()
        code = lib().mm_device_group_gather_batch(self.h, root, C.c_void_p(d_dst_pos.data_ptr()),
                                                  C.c_void_p(d_dst_sk.data_ptr()) if d_dst_sk is not None else None,
                                                  int(d_dst_pos.numel()), offs)
        if code == ERR["CAPACITY"]:
            raise MinimizerError(code, f"gather capacity {int(d_dst_pos.numel())} < {offs[n]}")
        _check(code)
        return [int(o) for o in offs]

    def run_batch(self, builder: "Builder", seqs, n_bases, base_offsets=None, capacity=None):
        """Independent host sequences placed greedily on the entries (``mm_run_batch_sharded_host``): (positions,
        super-k-mer indices or None, n + 1 offsets); sequence-local positions in input order."""
        plan = builder.plan()
        n = len(seqs)
        keep = [np.ascontiguousarray(a, dtype=np.uint8) for a in seqs]
        ptrs = (C.POINTER(C.c_uint8) * max(n, 1))(*[a.ctypes.data_as(C.POINTER(C.c_uint8)) for a in keep])
        lens = (C.c_uint64 * max(n, 1))(*n_bases)
        offs = (C.c_uint64 * max(n, 1))(*(base_offsets or [0] * n))
        cap = max(1, sum(n_bases)) if capacity is None else capacity
        pos = np.empty(cap, dtype=np.uint32)
        sk = np.empty(cap, dtype=np.uint32) if builder._sk is not None else None
        out_offsets = (C.c_uint64 * (n + 1))()
        code = lib().mm_run_batch_sharded_host(plan.h, self.h, n, ptrs, offs, lens, pos.ctypes.data_as(C.POINTER(C.c_uint32)),
                                               sk.ctypes.data_as(C.POINTER(C.c_uint32)) if sk is not None else None, cap,
                                               out_offsets)
        if code == ERR["CAPACITY"]:
            raise MinimizerError(code, f"output capacity {cap} < {out_offsets[n]}")
        _check(code)
        o = [int(x) for x in out_offsets]
        return pos[:o[-1]], (sk[:o[-1]] if sk is not None else None), o


_default_ws: dict[int, Workspace] = {}


def default_workspace(device: int = 0) -> Workspace:
    """Per-device workspace bound to torch's current stream when torch is importable, so that
    tensor fills / copies issued through torch are ordered with the engine's kernels."""
    if device not in _default_ws:
        stream = None
        try:
            import torch
            if torch.cuda.is_available():
                stream = torch.cuda.current_stream(device).cuda_stream
        except ImportError:
            pass
        _default_ws[device] = Workspace(device, stream)
    return _default_ws[device]


# -------------------------------------------------------------------- builder


def NtHasher(k: int | None = None, canonical: bool = True) -> Hasher:
    """seq-hash ``NtHasher::<CANONICAL>::new(k)`` (k only rotates tables inside the plan)."""
    h = Hasher()
    _check(lib().mm_default_hasher(C.byref(h), int(canonical)))
    return h


def MulHasher(k: int | None = None, canonical: bool = True) -> Hasher:
    """seq-hash ``MulHasher::<CANONICAL>::new(k)`` (src/lib.rs:71-72).  PARITY UNPINNED: this engine's
    restatement of the published idea (see mm_mul_hasher in the C header); plug the real per-base values
    in through ``Hasher.from_tables`` where they are known."""
    h = Hasher()
    _check(lib().mm_mul_hasher(C.byref(h), int(canonical)))
    return h


def AntiLexHasher(k: int, canonical: bool = True) -> Hasher:
    """seq-hash ``AntiLexHasher::<CANONICAL>::new(k)`` (src/test.rs:83,109).  PARITY UNPINNED."""
    h = Hasher()
    _check(lib().mm_antilex_hasher(C.byref(h), k, int(canonical)))
    return h


class Plan:
    def __init__(self, k, w, canonical, mode, hasher):
        h = C.c_void_p()
        hp = C.byref(hasher) if hasher is not None else None
        _check(lib().mm_plan_create(C.byref(h), k, w, int(canonical), mode, hp))
        self.h = h

    def __del__(self):
        try:
            if self.h:
                lib().mm_plan_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def value_len(self) -> int:
        return lib().mm_plan_value_len(self.h)


class Output:
    """``Output`` of src/lib.rs:232-237: positions + lazily computed k-mer values."""

    def __init__(self, builder, seq, min_pos):
        self.len = builder.k if builder.mode == MM_MINIMIZERS else builder.k + builder.w - 1
        self.canonical = builder.canonical
        self.seq = seq
        self.min_pos = min_pos
        self._ws = builder._ws()

    def values_u64(self) -> np.ndarray:
        return self.pos_and_values_u64()[1]

    def values_u128(self) -> list:
        """``Output::values_u128`` (src/lib.rs:587-593): Python ints, len <= 64."""
        return self.pos_and_values_u128()[1]

    def pos_and_values_u128(self):
        """``Output::pos_and_values_u128`` (src/lib.rs:615-630): (positions, Python ints)."""
        pos = np.ascontiguousarray(self.min_pos, dtype=np.uint32)
        vals = np.zeros(2 * len(pos), dtype=np.uint64)
        seq = self.seq
        if isinstance(seq, AsciiSeq):
            seq = PackedSeqVec.from_ascii(seq.seq)
        if len(pos):
            _check(lib().mm_values_u128_host(self._ws.h, _p(seq.data, C.c_uint8), seq.offset, seq.length,
                                             self.len, int(self.canonical), _p(pos, C.c_uint32),
                                             len(pos), _p(vals, C.c_uint64)))
        return pos, [int(vals[2 * i]) | (int(vals[2 * i + 1]) << 64) for i in range(len(pos))]

    def pos_and_values_u64(self):
        """``Output::pos_and_values_u64`` (src/lib.rs:598-612)."""
        pos = np.ascontiguousarray(self.min_pos, dtype=np.uint32)
        vals = np.zeros(len(pos), dtype=np.uint64)
        seq = self.seq
        if isinstance(seq, AsciiSeq):
            seq = PackedSeqVec.from_ascii(seq.seq)
        if len(pos):
            _check(lib().mm_values_u64_host(self._ws.h, _p(seq.data, C.c_uint8), seq.offset, seq.length,
                                            self.len, int(self.canonical), _p(pos, C.c_uint32),
                                            len(pos), _p(vals, C.c_uint64)))
        return pos, vals


class Builder:
    """``Builder<CANONICAL, H, SkPos, SYNCMER>`` (src/lib.rs:225-230)."""

    def __init__(self, k, w, canonical, mode, hasher=None, sk_pos=None, workspace=None):
        self.k, self.w, self.canonical, self.mode = k, w, canonical, mode
        self._hasher, self._sk, self._workspace = hasher, sk_pos, workspace
        self._plan = None

    def hasher(self, hasher: Hasher) -> "Builder":  # src/lib.rs:327
        return Builder(self.k, self.w, self.canonical, self.mode, hasher, self._sk, self._workspace)

    def super_kmers(self, sk_pos: list) -> "Builder":  # src/lib.rs:341 (minimizers only)
        if self.mode != MM_MINIMIZERS:
            raise MinimizerError(ERR["BAD_MODE"], "super_kmers() is only defined for minimizers")
        return Builder(self.k, self.w, self.canonical, self.mode, self._hasher, sk_pos, self._workspace)

    def workspace(self, ws: Workspace) -> "Builder":
        return Builder(self.k, self.w, self.canonical, self.mode, self._hasher, self._sk, ws)

    def _ws(self) -> Workspace:
        return self._workspace or default_workspace()

    def plan(self) -> Plan:
        if self._plan is None:
            self._plan = Plan(self.k, self.w, self.canonical, self.mode, self._hasher)
        return self._plan

    # -- host sequences -------------------------------------------------
    def run(self, seq, min_pos: list) -> Output:
        """``Builder::run``: positions are APPENDED to ``min_pos`` (src/lib.rs:80-81); like the
        reference's SIMD collector a leading result equal to ``min_pos[-1]`` is dropped
        (src/collect.rs:265-271)."""
        pos, sk = self._run_arrays(seq)
        pos, sk = list(map(int, pos)), (list(map(int, sk)) if sk is not None else None)
        if self.mode == MM_MINIMIZERS:
            while pos and min_pos and pos[0] == min_pos[-1]:
                pos = pos[1:]
                sk = sk[1:] if sk is not None else None
        min_pos.extend(pos)
        if self._sk is not None:
            self._sk.extend(sk)
        return Output(self, seq, min_pos)

    def run_once(self, seq) -> list:
        out: list = []
        self.run(seq, out)
        return out

    def run_with_buf(self, seq, min_pos: list, cache: "Workspace") -> Output:
        """``Builder::run_with_buf`` (src/lib.rs:553-576): ``run`` with the scratch passed explicitly -
        the reference's ``Cache`` is this engine's ``Workspace`` (device buffers + stream)."""
        return self.workspace(cache).run(seq, min_pos)

    def run_scalar(self, seq, min_pos: list) -> Output:
        """``Builder::run_scalar`` (src/lib.rs:370-376, :517-543).  The reference's scalar collectors
        OVERWRITE ``min_pos`` from index 0 and truncate it to the result (src/collect.rs:15-37,39-76,
        src/syncmers.rs:19-48) - no append, no ``last()`` rule; a sequence without a window clears
        ``min_pos`` and (like src/collect.rs:45-48) leaves the super-k-mer vector untouched.  Served by
        the same HIP kernel as ``run``: the scalar flavour differs in the collector's contract only."""
        pos, sk = self._run_arrays(seq)
        min_pos[:] = list(map(int, pos))
        if self._sk is not None and len(seq) >= self.k + self.w - 1:
            self._sk[:] = list(map(int, sk))
        return Output(self, seq, min_pos)

    def run_scalar_once(self, seq) -> list:
        """``Builder::run_scalar_once`` (src/lib.rs:358-362, :511-515)."""
        out: list = []
        self.run_scalar(seq, out)
        return out

    def _run_arrays(self, seq):
        L = lib()
        ws = self._ws()
        plan = self.plan()
        n = len(seq)
        lwin = self.k + self.w - 1
        cap = max(1, n - lwin + 1) if n >= lwin else 1
        pos = np.zeros(cap, dtype=np.uint32)
        want_sk = self._sk is not None
        sk = np.zeros(cap, dtype=np.uint32) if want_sk else None
        cnt = C.c_uint64()
        skp = _p(sk, C.c_uint32) if want_sk else None
        if isinstance(seq, AsciiSeq):
            a = np.frombuffer(seq.seq, dtype=np.uint8)
            _check(L.mm_run_host_ascii(plan.h, ws.h, _p(a, C.c_uint8) if n else None, n,
                                       _p(pos, C.c_uint32), skp, cap, C.byref(cnt)))
        else:
            _check(L.mm_run_host(plan.h, ws.h, _p(seq.data, C.c_uint8), seq.offset, seq.length,
                                 _p(pos, C.c_uint32), skp, cap, C.byref(cnt)))
        m = cnt.value
        return pos[:m], (sk[:m] if want_sk else None)

    # -- PackedNSeq: skip windows with ambiguous bases (src/lib.rs:451-496) --
    def run_skip_ambiguous_windows(self, nseq, min_pos: list) -> Output:
        """``Builder::run_skip_ambiguous_windows`` (canonical builders only). ``nseq`` is a
        ``PackedNSeq`` or, as a convenience, an ``AsciiSeq`` (packed on the device)."""
        L = lib()
        ws = self._ws()
        plan = self.plan()
        n = len(nseq)
        lwin = self.k + self.w - 1
        cap = max(1, n - lwin + 1) if n >= lwin else 1
        pos = np.zeros(cap, dtype=np.uint32)
        cnt = C.c_uint64()
        if isinstance(nseq, AsciiSeq):
            a = np.frombuffer(nseq.seq, dtype=np.uint8)
            _check(L.mm_run_skip_ambiguous_host_ascii(plan.h, ws.h, _p(a, C.c_uint8) if n else None, n,
                                                      _p(pos, C.c_uint32), cap, C.byref(cnt)))
            seq = PackedSeqVec.from_ascii(nseq.seq)
        else:
            seq = nseq.seq
            _check(L.mm_run_skip_ambiguous_host(plan.h, ws.h, _p(seq.data, C.c_uint8), seq.offset,
                                                _p(nseq.amb, C.c_uint8), nseq.amb_offset, n,
                                                _p(pos, C.c_uint32), cap, C.byref(cnt)))
        out = list(map(int, pos[:cnt.value]))
        if self.mode == MM_MINIMIZERS:
            while out and min_pos and out[0] == min_pos[-1]:
                out = out[1:]
        min_pos.extend(out)
        return Output(self, seq, min_pos)

    def run_skip_ambiguous_windows_with_buf(self, nseq, min_pos: list, cache: "Workspace") -> Output:
        """``Builder::run_skip_ambiguous_windows_with_buf`` (src/lib.rs:465-496)."""
        return self.workspace(cache).run_skip_ambiguous_windows(nseq, min_pos)

    def run_skip_ambiguous_windows_once(self, nseq) -> list:
        out: list = []
        self.run_skip_ambiguous_windows(nseq, out)
        return out

    def run_skip_ambiguous_device(self, d_packed, d_amb, n_bases: int, out_pos, base_offset: int = 0,
                                  amb_offset: int = 0, win_begin: int = 0, win_end: int = U64_MAX,
                                  sync: bool = True, d_count=None):
        """Device-resident PackedNSeq (torch uint8 CUDA tensors for the codes and the ambiguity bits)."""
        L = lib()
        cap = out_pos.numel() if out_pos is not None else 0
        args = [self.plan().h, self._ws().h, C.c_void_p(d_packed.data_ptr()), d_packed.numel(), base_offset,
                C.c_void_p(d_amb.data_ptr()), d_amb.numel(), amb_offset, n_bases, win_begin, win_end,
                C.c_void_p(out_pos.data_ptr()) if out_pos is not None else None, cap]
        if sync:
            cnt = C.c_uint64()
            code = L.mm_run_skip_ambiguous_device(*args, C.byref(cnt))
            if code == ERR["CAPACITY"]:
                raise MinimizerError(code, f"output capacity {cap} < {cnt.value}")
            _check(code)
            return cnt.value
        _check(L.mm_run_skip_ambiguous_device_async(*args, C.c_void_p(d_count.data_ptr()) if d_count is not None else None))
        return None

    # -- device-resident sequences (torch uint8 CUDA tensors) -----------
    def run_device(self, d_packed, n_bases: int, out_pos, out_sk=None, base_offset: int = 0,
                   win_begin: int = 0, win_end: int = U64_MAX, sync: bool = True, d_count=None):
        """Run on a device-resident PackedSeq. ``d_packed``/``out_pos``/``out_sk``/``d_count`` are
        torch CUDA tensors (uint8 / int32-or-uint32 / int64). Returns the count if ``sync``."""
        L = lib()
        ws = self._ws()
        plan = self.plan()
        cap = out_pos.numel() if out_pos is not None else 0
        args = [plan.h, ws.h, C.c_void_p(d_packed.data_ptr()), d_packed.numel(), base_offset, n_bases,
                win_begin, win_end, C.c_void_p(out_pos.data_ptr()) if out_pos is not None else None,
                C.c_void_p(out_sk.data_ptr()) if out_sk is not None else None, cap]
        if sync:
            cnt = C.c_uint64()
            code = L.mm_run_device(*args, C.byref(cnt))
            if code == ERR["CAPACITY"]:
                raise MinimizerError(code, f"output capacity {cap} < {cnt.value}")
            _check(code)
            return cnt.value
        _check(L.mm_run_device_async(*args, C.c_void_p(d_count.data_ptr()) if d_count is not None else None))
        return None


def run_batch_device(builder: "Builder", d_seqs, n_bases, out_pos, out_sk=None, base_offsets=None):
    """Many device-resident sequences (list of torch uint8 CUDA tensors) with one plan; positions
    are sequence-local, written back to back into ``out_pos``. Returns the n_seqs+1 offsets."""
    L = lib()
    ws = builder._ws()
    plan = builder.plan()
    n = len(d_seqs)
    ptrs = (C.c_void_p * max(n, 1))(*[t.data_ptr() for t in d_seqs])
    nbytes = (C.c_uint64 * max(n, 1))(*[t.numel() for t in d_seqs])
    lens = (C.c_uint64 * max(n, 1))(*n_bases)
    offs = (C.c_uint64 * max(n, 1))(*(base_offsets or [0] * n))
    out_offsets = (C.c_uint64 * (n + 1))()
    code = L.mm_run_batch_device(plan.h, ws.h, n, ptrs, nbytes, offs, lens, C.c_void_p(out_pos.data_ptr()),
                                 C.c_void_p(out_sk.data_ptr()) if out_sk is not None else None,
                                 out_pos.numel(), out_offsets)
    _check(code)
    return list(out_offsets)


def run_reads_device(builder: "Builder", d_packed, n_reads, read_stride, read_len, out_pos, out_offsets,
                     read_lens=None, base_offset=0, d_count=None, sync=True, d_amb=None, amb_offset=0,
                     out_sk=None):
    """Batched short reads in one packed device buffer (read r at base ``base_offset + r*read_stride``);
    read-local positions go back to back into ``out_pos`` and ``out_offsets`` (int64/uint64 CUDA
    tensor, n_reads+1) delimits the reads; ``out_sk`` also receives the super-k-mer indices
    (read-local window of the first selection). Returns the total count when ``sync``."""
    L = lib()
    ws = builder._ws()
    plan = builder.plan()
    cap = out_pos.numel() if out_pos is not None else 0
    args = [plan.h, ws.h, C.c_void_p(d_packed.data_ptr()), d_packed.numel(), base_offset]
    if d_amb is not None:  # PackedNSeq reads: skip windows with ambiguous bases
        args += [C.c_void_p(d_amb.data_ptr()), d_amb.numel(), amb_offset]
    args += [n_reads, read_stride, read_len, C.c_void_p(read_lens.data_ptr()) if read_lens is not None else None,
             C.c_void_p(out_pos.data_ptr()) if out_pos is not None else None]
    if out_sk is not None:
        if d_amb is not None:
            raise MinimizerError(ERR["BAD_MODE"], "no super-k-mer flavour of the skip-ambiguous run")
        args += [C.c_void_p(out_sk.data_ptr())]
    args += [cap, C.c_void_p(out_offsets.data_ptr())]
    if out_sk is not None:
        fsync, fasync = L.mm_run_reads_superkmers_device, L.mm_run_reads_superkmers_device_async
    else:
        fsync, fasync = ((L.mm_run_reads_skip_ambiguous_device, L.mm_run_reads_skip_ambiguous_device_async)
                         if d_amb is not None else (L.mm_run_reads_device, L.mm_run_reads_device_async))
    if sync:
        cnt = C.c_uint64(0)
        code = fsync(*args, C.byref(cnt))
        if code == ERR["CAPACITY"]:
            raise MinimizerError(code, f"output capacity {cap} < {cnt.value}")
        _check(code)
        return cnt.value
    _check(fasync(*args, C.c_void_p(d_count.data_ptr()) if d_count is not None else None))
    return None


def pinned_array(shape, dtype):
    """numpy array backed by page-locked host memory (mm_host_alloc); keep the returned owner alive."""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = C.c_void_p()
    _check(lib().mm_host_alloc(C.byref(p), max(n, 1)))

    class _Owner:
        def __init__(self, ptr):
            self.ptr = ptr

        def __del__(self):
            try:
                lib().mm_host_free(self.ptr)
            except Exception:
                pass

    owner = _Owner(p)
    buf = (C.c_uint8 * max(n, 1)).from_address(p.value)
    arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    return arr, owner


def minimizers(k, w):  # src/lib.rs:240
    return Builder(k, w, False, MM_MINIMIZERS)


def canonical_minimizers(k, w):  # src/lib.rs:250
    return Builder(k, w, True, MM_MINIMIZERS)


def closed_syncmers(k, w):  # src/lib.rs:269
    return Builder(k, w, False, MM_CLOSED_SYNCMERS)


def canonical_closed_syncmers(k, w):  # src/lib.rs:282
    return Builder(k, w, True, MM_CLOSED_SYNCMERS)


def open_syncmers(k, w):  # src/lib.rs:301
    return Builder(k, w, False, MM_OPEN_SYNCMERS)


def canonical_open_syncmers(k, w):  # src/lib.rs:311
    return Builder(k, w, True, MM_OPEN_SYNCMERS)


# README.md:65 uses this older name for canonical closed syncmers
canonical_syncmers = canonical_closed_syncmers
syncmers = closed_syncmers


def minimizer_positions(seq, k, w):  # src/lib.rs:639
    return minimizers(k, w).run_once(seq)


def canonical_minimizer_positions(seq, k, w):  # src/lib.rs:652
    return canonical_minimizers(k, w).run_once(seq)


class FastaRecords:
    """Records of a FASTA text packed on the device (``fasta_pack_device``): ``packed`` = one 2-bit buffer holding
    all sequences back to back, ``base`` = the n + 1 base offsets delimiting them, ``text_pos`` = byte offset of
    every record's '>' in the text."""

    def __init__(self, packed, base, text_pos):
        self.packed, self.base, self.text_pos = packed, base, text_pos

    def __len__(self):
        return len(self.base) - 1

    def lengths(self):
        return [int(self.base[i + 1] - self.base[i]) for i in range(len(self))]

    def views(self):
        """(tensor, base_offset, n_bases) per record, as ``run_batch_device`` takes them."""
        out = []
        for i in range(len(self)):
            b, e = int(self.base[i]), int(self.base[i + 1])
            out.append((self.packed[b // 4:], b % 4, e - b))
        return out

    def header(self, text: bytes, i: int) -> bytes:
        """The header line of record i (without '>' and line end), sliced from the host copy of the text."""
        p = int(self.text_pos[i]) + 1
        q = text.find(b"\n", p)
        return text[p:len(text) if q < 0 else q].rstrip(b"\r")


def fasta_pack_device(text, max_records: int = 1 << 16, device: int = 0) -> FastaRecords:
    """needletail::parse_fastx_file + PackedSeqVec::from_ascii of every record (bench/src/lib.rs:51-82) on the
    device: ``text`` = the file's bytes (bytes / numpy uint8 / torch uint8 CUDA tensor), FASTA or - since round 4,
    told apart by the first non-blank byte like needletail does - FASTQ (four-line records)."""
    import torch

    dev = f"cuda:{device}"
    if isinstance(text, (bytes, bytearray)):
        text = np.frombuffer(bytes(text), dtype=np.uint8).copy()
    if isinstance(text, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(text)).to(dev) if text.size else torch.zeros(0, dtype=torch.uint8, device=dev)
    else:
        t = text
    n = int(t.numel())
    ws = default_workspace(device)
    packed = torch.empty((n // 4 + 8 + 3) // 4 * 4 + 64, dtype=torch.uint8, device=dev)
    rec_base = torch.zeros(max_records + 1, dtype=torch.int64, device=dev)
    rec_pos = torch.zeros(max(max_records, 1), dtype=torch.int64, device=dev)
    counts = torch.zeros(2, dtype=torch.int64, device=dev)
    torch.cuda.synchronize(device)
    out = (C.c_uint64 * 2)()
    code = lib().mm_fasta_pack_device(ws.h, C.c_void_p(t.data_ptr()) if n else None, n, C.c_void_p(packed.data_ptr()),
                                      packed.numel() // 4 * 4, C.c_void_p(rec_base.data_ptr()),
                                      C.c_void_p(rec_pos.data_ptr()), max_records, C.c_void_p(counts.data_ptr()), out)
    if code == ERR["CAPACITY"]:
        if out[1] > max_records:
            raise MinimizerError(code, f"{out[1]} records > max_records {max_records}")
        raise MinimizerError(code, f"{out[0]} bases do not fit the packed buffer of {packed.numel() // 4 * 4} bytes")
    _check(code)
    n_rec = int(out[1])
    return FastaRecords(packed, rec_base[: n_rec + 1].cpu().numpy().astype(np.uint64),
                        rec_pos[:n_rec].cpu().numpy().astype(np.uint64))


fastx_pack_device = fasta_pack_device  # (the reference's loader call reads both formats)


def run_reads_host(builder: "Builder", reads, super_kmers: bool = False):
    """Many short host sequences in ONE call (``mm_run_packed_reads_host``): ``reads`` = list of ``PackedSeq`` / ASCII
    ``bytes``; returns (positions, offsets, indices or None) - read r's read-local positions are
    ``positions[offsets[r]:offsets[r + 1]]``.  The replacement of a per-read loop over ``Builder.run``."""
    codes = []
    for s in reads:
        if isinstance(s, (bytes, bytearray)):
            a = np.frombuffer(bytes(s), dtype=np.uint8)
            codes.append((a >> 1) & 3)
        else:  # PackedSeq view
            a = np.asarray(s.data, dtype=np.uint8)
            idx = np.arange(s.offset, s.offset + s.len)
            codes.append((a[idx // 4] >> (2 * (idx % 4))) & 3)
    lens = [len(c) for c in codes]
    starts = np.zeros(len(lens) + 1, dtype=np.uint64)
    starts[1:] = np.cumsum(lens, dtype=np.uint64)
    total = int(starts[-1])
    flat = np.concatenate(codes).astype(np.uint8) if total else np.zeros(0, dtype=np.uint8)
    pad = np.zeros((-total) % 4, dtype=np.uint8)
    q = np.concatenate([flat, pad]).reshape(-1, 4)
    packed = (q[:, 0] | (q[:, 1] << 2) | (q[:, 2] << 4) | (q[:, 3] << 6)).astype(np.uint8) if total else np.zeros(1, dtype=np.uint8)
    cap = max(1, total)
    pos = np.empty(cap, dtype=np.uint32)
    sk = np.empty(cap, dtype=np.uint32) if super_kmers else None
    offs = np.zeros(len(lens) + 1, dtype=np.uint64)
    cnt = C.c_uint64()
    ws = builder._ws()
    _check(lib().mm_run_packed_reads_host(builder.plan().h, ws.h, packed.ctypes.data_as(C.POINTER(C.c_uint8)), len(lens),
                                          starts.ctypes.data_as(C.POINTER(C.c_uint64)), max(lens) if lens else 0,
                                          pos.ctypes.data_as(C.POINTER(C.c_uint32)),
                                          sk.ctypes.data_as(C.POINTER(C.c_uint32)) if sk is not None else None, cap,
                                          offs.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(cnt)))
    n = int(cnt.value)
    return pos[:n], [int(o) for o in offs], (sk[:n] if sk is not None else None)


def run_packed_reads_device(builder: "Builder", records: FastaRecords, out_pos, out_offsets, out_sk=None, max_read_len=None):
    """All records of a packed FASTQ / FASTA (``fasta_pack_device``) as READS: one launch of the reads-mode kernel over
    reads of any lengths packed back to back (``mm_run_packed_reads_device``); read-local positions, ``out_offsets``
    (int64 device tensor, n + 1) delimits the reads.  Returns the number of positions."""
    import torch
    n = len(records)
    starts = torch.from_numpy(np.ascontiguousarray(records.base, dtype=np.uint64).view(np.int64)).to(out_pos.device)
    lens = records.lengths()
    mx = max(lens) if lens else 0
    if max_read_len is None:
        max_read_len = mx
    cnt = C.c_uint64()
    ws = builder._ws()
    _check(lib().mm_run_packed_reads_device(builder.plan().h, ws.h, C.c_void_p(records.packed.data_ptr()),
                                            int(records.packed.numel()), 0, n, C.c_void_p(starts.data_ptr()),
                                            int(records.base[-1]) if n else 0, int(max_read_len),
                                            C.c_void_p(out_pos.data_ptr()),
                                            C.c_void_p(out_sk.data_ptr()) if out_sk is not None else None,
                                            int(out_pos.numel()), C.c_void_p(out_offsets.data_ptr()), C.byref(cnt)))
    return int(cnt.value)


def run_fasta_device(builder: "Builder", records: FastaRecords, out_pos, out_sk=None):
    """All records of a packed FASTA (``fasta_pack_device``) with one plan in one launch (``mm_run_batch_device``);
    record-local positions back to back in ``out_pos``; returns the n + 1 offsets."""
    v = records.views()
    return run_batch_device(builder, [t for t, _, _ in v], [m for _, _, m in v], out_pos, out_sk,
                            base_offsets=[o for _, o, _ in v])


def generate_device(n_bases: int, seed: int, device: int = 0, first_base: int = 0):
    """Synthetic PackedSeq (generator G) written directly into HBM.
    Returns a torch uint8 CUDA tensor of ceil(n/4)+64 bytes."""
    import torch

    ws = default_workspace(device)
    t = torch.zeros((n_bases + 3) // 4 + 64, dtype=torch.uint8, device=f"cuda:{device}")
    torch.cuda.synchronize(device)
    _check(lib().mm_generate_device_async(ws.h, seed, first_base, n_bases, C.c_void_p(t.data_ptr())))
    ws.sync()
    return t
