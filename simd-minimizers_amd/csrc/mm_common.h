// mm_common.h — shared device-side definitions of the gfx950 minimizer engine.
//
// Reference semantics (rust-seq/simd-minimizers v3.0.0, paths relative to /root/reference):
//   key of a k-mer      = (hash & 0xffff0000) | pos16            src/sliding_min.rs:104-127
//   right-most variant  = (~hash & 0xffff0000) | pos16 with MAX  src/sliding_min.rs:196-197
//   strand vote         = #(T|G) in the l-base window > l/2      src/canonical.rs:18-29
//   dedup               = drop ADJACENT equal positions          src/collect.rs:15-37
//   syncmer filter      = minpos in {i, i+w-1} / == i + w/2      src/syncmers.rs:33-37
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef MM_LB_SLEEP
#define MM_LB_SLEEP 120  // fused kernel: s_sleep argument (x 64 clocks, the maximum) between two polls of
                         // the nearest missing predecessor; shorter intervals measure slower
#endif
#ifndef MM_LB_FIRST
#define MM_LB_FIRST 1  // 64-status chunks of the first look-back hop (1 measured best; then 4, 16)
#endif

namespace mm {

constexpr int kBlockThreads = 256;
constexpr int kWave = 64;
constexpr int kWavesPerBlock = kBlockThreads / kWave;
#ifndef MM_FUSED_THREADS
#define MM_FUSED_THREADS 256
#endif
constexpr int kFusedThreads = MM_FUSED_THREADS;  // workgroup size of the fused kernel
constexpr int kFusedWaves = kFusedThreads / kWave;

// Rolling ntHash tables prepared on the host for one (hasher, k):
//   fw' = rotl(fw, rot) ^ t_in_out[out][in].x ;  rc' = rotr(rc, rot) ^ t_in_out[out][in].y
//   warm-up (no base leaves yet): the same with t_in[in]
struct HashTables {
    uint2 t_in_out[16];  // index = (out << 2) | in
    uint2 t_in[4];
    uint2 t_in2[16];     // two warm-up steps at once: index = (second << 2) | first
    uint32_t rot;
    uint32_t canonical;  // 1: h = fw + rc, 0: h = fw
    // state before the first base: the hasher's constant XOR terms (mm_hasher_t::fw_xor / rc_xor; 0 for
    // NtHasher).  The tables above carry rot(C) ^ C per step, so the rolled state IS hash ^ C throughout.
    uint32_t fw0, rc0;
};

// A PackedSeq seen as little-endian dwords: base i of the sequence sits at
// bits 2*((base0 + i) % 16) of dword (base0 + i) / 16.
struct SeqView {
    const uint32_t *d;
    uint32_t n_dwords;  // readable dwords (loads are clamped to [0, n_dwords))
    uint32_t base0;
    uint32_t n_bases;
};

// One sequence of a batch launch (device table, 32 bytes).
struct BatchSeq {
    const uint32_t *d;    // dword-aligned base of the packed bytes
    uint32_t n_dwords;    // readable dwords
    uint32_t base0;       // first base inside d
    uint32_t n_windows;   // windows of this sequence (> 0: empty sequences get no tile)
    uint32_t first_tile;  // id of the sequence's first tile
    uint32_t pad[2];
};

// One tile of a batch launch (device table, 16 bytes): the sequence it belongs to, its first window (sequence-local)
// and its lane length.  The host lays the tiles out (fused_batch_tiles, mm_fused.hip): whole tiles per sequence, the
// last round of the LAUNCH tapered like a single sequence's (FusedParams::taper_*).
struct BatchTile {
    uint32_t seq;
    uint32_t win0;
    uint32_t nblk;
    uint32_t pad;
};

// One lane of a lane-table launch (reads-mode kernels, round 6): the lane walks the windows [win0, win0 + count) of
// read (sequence) `read`; `start` = base of the buffer span at which window win0 starts.  A tile = 256 consecutive
// entries; a read longer than a lane takes consecutive lanes (the seam between two lanes of one read is the dedup
// rule against the lane's predecessor window, as between the lanes of a single sequence), a read's FIRST lane
// (win0 == 0) has no predecessor and stores the read's output offset.  Built on the device (mm_lanes.hip) from the
// reads' starts and lengths; entries behind the last lane of the table carry count = 0 and win0 != 0.
struct LaneSeg {
    uint32_t start;
    uint32_t win0;
    uint32_t count;
    uint32_t read;
};

struct OutParams {
    uint32_t *pos;
    uint32_t *sk;  // may be null
    unsigned long long cap;
    unsigned long long *total;   // device counter: running total of outputs
    unsigned long long *status;  // decoupled look-back words, one per block, zeroed per launch
    uint32_t *ticket;            // dynamic block id counter, zeroed per launch
    uint32_t *error;             // error[0]: flag of this launch (1 = a look-back spin ran out, see
                                 // lookback_exclusive; 2 = LDS layout violated; 4 = skip-ambiguous launch without its landing area; 5 = lane table too small for the reads (mm_lanes.hip); 0xbad..... = bad batch
                                 // table), cleared by the entry point that reads it; error[2]: the same, sticky
                                 // until mm_workspace_check() reads it (asynchronous callers); error[4..5]: 0 or the
                                 // device address of a page-locked HOST word that receives the code as well
                                 // (synchronous callers of the fused kernel read it without a copy, see flag_error)
    // fused kernel only (round 4: one stream operation per run instead of four)
    unsigned long long *count_out = nullptr;   // optional device word: receives the run's total like *total
    unsigned long long *total_host = nullptr;  // optional page-locked host word (device address): likewise
};

// Raise an error from a kernel: the per-launch word the synchronous entry points read, and the sticky
// word mm_workspace_check() reports to asynchronous callers.
__device__ __forceinline__ void flag_error(uint32_t *error, uint32_t code) {
    error[0] = code;
    error[2] = code;
    // (rare path: the address of the host's copy of error[0] travels in device memory so that no signature changes)
    uint32_t *host = *reinterpret_cast<uint32_t *const volatile *>(error + 4);
    if (host) __hip_atomic_store(host, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ uint32_t rotl32(uint32_t x, uint32_t r) {
    return __builtin_amdgcn_alignbit(x, x, (32u - r) & 31u);
}
__device__ __forceinline__ uint32_t rotr32(uint32_t x, uint32_t r) {
    return __builtin_amdgcn_alignbit(x, x, r & 31u);
}

__device__ __forceinline__ uint32_t load_dword_clamped(const SeqView &s, long long q) {
    long long hi = (long long)s.n_dwords - 1;
    q = q < 0 ? 0 : (q > hi ? hi : q);
    return s.d[q];
}
// 2-bit code of the base at absolute position p (in dword-array coordinates)
__device__ __forceinline__ uint32_t base_at(const SeqView &s, long long p) {
    uint32_t wd = load_dword_clamped(s, p >> 4);
    return (wd >> (2u * (uint32_t)(p & 15))) & 3u;
}

// ---------------------------------------------------------------- wave scan
// inclusive prefix sum over the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v) {
    const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        uint32_t t = __shfl_up(v, d, kWave);
        if (lane >= d) v += t;
    }
    return v;
}

// ------------------------------------------------- decoupled look-back scan
// status word: bits 63..62 = flag (1 aggregate, 2 inclusive prefix), low 62 bits = value.
// One naturally aligned 8-byte word carries flag and value together, so no fence is
// needed: both sides use relaxed agent-scope atomics (cross-XCD safe on gfx950).
constexpr unsigned long long kFlagAgg = 1ull << 62;
constexpr unsigned long long kFlagIncl = 2ull << 62;
constexpr unsigned long long kValMask = (1ull << 62) - 1;
// Fused kernel (round 4): bits 61..46 of a status word carry the EPOCH of the launch that wrote it and the value
// keeps 46 bits (7e13 outputs).  A word of another epoch reads as "not yet", so the words are not cleared between
// launches: the workspace hands every launch the next epoch and clears the buffer once per 65 535 launches (and
// when it is reallocated or lent to a kernel family that does not tag).  Epoch 0 with a cleared buffer is the
// untagged protocol of the other families.
constexpr int kEpochShift = 46;
constexpr unsigned long long kEpochValMask = (1ull << kEpochShift) - 1;
constexpr uint32_t kEpochMax = 0xffffu;
// flag of a status word as launch `etag` (= epoch << kEpochShift) sees it: 0 unless the word is its own
__device__ __forceinline__ uint32_t status_flag(unsigned long long s, unsigned long long etag) {
    const unsigned long long x = s ^ etag;
    return ((x >> kEpochShift) & kEpochMax) == 0ull ? (uint32_t)(s >> 62) : 0u;
}
constexpr uint32_t kMaxLookbackSpins = 1u << 22;

__device__ __forceinline__ unsigned long long ld_status(unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_status(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Called by ALL threads of wave 0 of the block (uniformly). `bid` is the dynamic
// block id, `block_total` this block's output count, `carry_in` the count produced
// before this launch (only used by block 0). Returns the exclusive prefix.
__device__ __forceinline__ unsigned long long lookback_exclusive(unsigned long long *status,
                                                                 uint32_t bid,
                                                                 unsigned long long block_total,
                                                                 unsigned long long carry_in,
                                                                 uint32_t *error) {
    const int lane = threadIdx.x & (kWave - 1);
    if (bid == 0) {
        if (lane == 0) st_status(&status[0], kFlagIncl | ((carry_in + block_total) & kValMask));
        return carry_in;
    }
    if (lane == 0) st_status(&status[bid], kFlagAgg | (block_total & kValMask));
    // Look back over 64, then 256, then 1024 predecessors per hop (all loads of a hop in flight
    // together): when many tiles finish at about the same time none of the nearest ones has its
    // inclusive prefix yet, and one 64-wide hop per memory round trip would serialise the scan.
    unsigned long long excl = 0;
    long long j = (long long)bid - 1;
    int chunks = MM_LB_FIRST;
    while (true) {
        unsigned long long s[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const long long idx = j - 64ll * c - lane;
            s[c] = (c < chunks && idx >= 0) ? ld_status(&status[idx]) : kFlagIncl;
        }
        bool done = false;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (c < chunks && !done) {
                const long long idx = j - 64ll * c - lane;
                // bounded: a predecessor that never shows up (dispatch-order violation) must not
                // hang the GPU; the launch is then reported as failed and redone in ticket mode
                for (uint32_t spins = 0; (s[c] >> 62) == 0; ++spins) {
                    if (spins > kMaxLookbackSpins) {
                        flag_error(error, 1u);
                        s[c] = kFlagIncl;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                    s[c] = ld_status(&status[idx]);
                }
                const unsigned long long incl_mask = __ballot((s[c] >> 62) == 2);
                const int first = incl_mask ? __builtin_ctzll(incl_mask) : kWave;
                unsigned long long v = (lane <= first) ? (s[c] & kValMask) : 0ull;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, kWave);  // wave reduce (64-bit)
                excl += v;
                if (incl_mask) done = true;
            }
        }
        if (done) break;
        j -= 64ll * chunks;
        chunks = chunks >= 4 ? 16 : chunks * 4;
    }
    if (lane == 0) st_status(&status[bid], kFlagIncl | ((excl + block_total) & kValMask));
    return excl;
}

}  // namespace mm
