// mm_text.h - what the two text packers that work on 32-bit masks share (mm_fastq.hip, mm_fasta2.hip): a workgroup of
// 256 threads reads a 16 KB chunk of text as two 8 KB pieces, 32 bytes per thread and piece (coalesced 16-byte loads),
// and turns every thread's 32 bytes into bit masks (bit i = byte i) by SWAR; sums over the chunk run in TEXT order
// (piece-major: piece p of thread t lies behind piece p of every thread < t and behind every earlier piece).
#pragma once
#include "mm_common.h"

namespace mm {
namespace {

constexpr uint32_t kFqThreads = 256;
constexpr uint32_t kFqBytesPerThread = 32;
constexpr uint32_t kFqPiece = kFqThreads * kFqBytesPerThread;   // 8 KB of text per piece of a workgroup
constexpr uint32_t kFqPieces = 2;                               // pieces per workgroup
constexpr uint32_t kFqChunk = kFqPiece * kFqPieces;             // 16 KB of text per workgroup (one entry of the sums)
constexpr int kFqWaves = (int)(kFqThreads / kWave);
static_assert(kFqPieces == 2 && kFqBytesPerThread == 32, "masks are 32 bits, the '\\r' test is written for two pieces");

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct Text32 {
    uint32_t d[8];  // the thread's 32 text bytes of a piece, little-endian dwords
};

// the thread's 32 text bytes (zeros past the end of the text), from a bounds-checked view of the piece that starts at
// byte c0 of the text
__device__ __forceinline__ Text32 load32(const uint8_t *text, uint64_t n, uint64_t c0, uint32_t t) {
    Text32 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.d[i] = 0u;
    if (c0 >= n) return r;
    const uint64_t left = n - c0;
    const uint32_t here = left < kFqPiece ? (uint32_t)left : kFqPiece;
    // (the text pointer may have any alignment: the view starts at the dword that holds byte c0)
    const uintptr_t a = reinterpret_cast<uintptr_t>(text + c0);
    const uint32_t sh = (uint32_t)(a & 3u);
    const __amdgpu_buffer_rsrc_t rs =
        // (whole dwords: the bounds check drops a dword that is only partly inside, and the text's last dword may be)
        __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<uint32_t *>(a - sh), 0, (int)((here + sh + 3u) & ~3u), 0x00020000);
    const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(rs, t * 32u, 0, 0);
    const u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(rs, t * 32u, 16, 0);
    uint32_t w[9] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w, 0u};
    if (sh == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) r.d[i] = w[i];
        return r;
    }
    w[8] = __builtin_amdgcn_raw_buffer_load_b32(rs, t * 32u, 32, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) r.d[i] = __builtin_amdgcn_alignbyte(w[i + 1], w[i], sh);
    return r;
}
__device__ __forceinline__ uint32_t byte_of(const Text32 &v, int i) { return (v.d[i >> 2] >> (8 * (i & 3))) & 0xffu; }

// 4-bit mask of the bytes of x that equal the byte replicated in pat (exact zero-byte test, then the four flag bits
// at 7 / 15 / 23 / 31 gathered by one multiply: the partial products do not overlap)
__device__ __forceinline__ uint32_t eq4(uint32_t x, uint32_t pat) {
    const uint32_t z = x ^ pat;
    const uint32_t t = (z & 0x7f7f7f7fu) + 0x7f7f7f7fu;
    const uint32_t m = ~(t | z | 0x7f7f7f7fu);
    return (m * 0x00204081u) >> 28;
}
__device__ __forceinline__ uint32_t eq32(const Text32 &v, uint32_t pat) {
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) m |= eq4(v.d[i], pat) << (4 * i);
    return m;
}
// does any byte equal the byte replicated in pat?  (three operations per dword; exact as a yes / no)
__device__ __forceinline__ uint32_t any_eq32(const Text32 &v, uint32_t pat) {
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t z = v.d[i] ^ pat;
        m |= (z - 0x01010101u) & ~z & 0x80808080u;
    }
    return m;
}
// inclusive prefix sum over the 64 lanes of a wave with DPP row shifts / broadcasts (also of packed 16-bit fields whose
// sums stay below 2^16)
__device__ __forceinline__ uint32_t fq_wave_scan(uint32_t v) {
    v += __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
    v += __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
    return v;
}
// inclusive prefix XOR of a 32-bit mask
__device__ __forceinline__ uint32_t pxor32(uint32_t x) {
    x ^= x << 1;
    x ^= x << 2;
    x ^= x << 4;
    x ^= x << 8;
    x ^= x << 16;
    return x;
}

// does the thread's first byte of this piece start a line?  The byte in front of it is the last byte of the lane before
// (one shuffle); lane 0 of a wave reads it from memory.
__device__ __forceinline__ bool starts_line_of(const uint8_t *text, uint64_t n, uint64_t b0, const Text32 &v) {
    const int lane = threadIdx.x & (kWave - 1);
    uint32_t prev = __shfl_up(v.d[7] >> 24, 1, kWave);
    if (lane == 0) prev = (b0 > 0 && b0 <= n) ? text[b0 - 1] : (uint32_t)'\n';
    return b0 == 0 || (b0 < n && prev == (uint32_t)'\n');
}

// Sums over the chunk in TEXT order (piece-major: piece p of thread t lies behind piece p of every thread < t and
// behind every earlier piece).  In: the thread's value per piece.  Out: the exclusive prefix per piece; returns the total.
__device__ __forceinline__ uint32_t chunk_exclusive(const uint32_t (&v)[kFqPieces], uint32_t (&excl)[kFqPieces],
                                                    uint32_t (*s_part)[kFqWaves]) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    uint32_t incl[kFqPieces];
#pragma unroll
    for (int p = 0; p < (int)kFqPieces; ++p) {
        incl[p] = fq_wave_scan(v[p]);
        if (lane == kWave - 1) s_part[p][wave] = incl[p];
    }
    __syncthreads();
    uint32_t run = 0;
#pragma unroll
    for (int p = 0; p < (int)kFqPieces; ++p) {
        uint32_t before = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < kFqWaves; ++w) {
            const uint32_t x = s_part[p][w];
            if (w < wave) before += x;
            tot += x;
        }
        excl[p] = run + before + incl[p] - v[p];
        run += tot;
    }
    __syncthreads();
    return run;
}

__device__ __forceinline__ unsigned long long block_exclusive64(unsigned long long v, unsigned long long *s_wave,
                                                                unsigned long long *total) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    unsigned long long x = v;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const unsigned long long y = __shfl_up(x, d, kWave);
        if (lane >= d) x += y;
    }
    if (lane == kWave - 1) s_wave[wave] = x;
    __syncthreads();
    unsigned long long before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kFqWaves; ++w) {
        const unsigned long long t = s_wave[w];
        if (w < wave) before += t;
        tot += t;
    }
    __syncthreads();
    *total = tot;
    return before + x - v;
}
}  // namespace
}  // namespace mm
