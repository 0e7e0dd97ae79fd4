// mm_aux.hip — the steps either side of the hot path: k-mer values of the sampled positions
// (Output::values_u64, src/lib.rs:584-612), ASCII -> PackedSeq packing
// (PackedSeqVec::from_ascii, call site src/lib.rs:110) and the synthetic input generator.
#include "mm_common.h"
#include "mm_launch.h"

namespace mm {

// packed-seq read_kmer: base j of the k-mer at bits 2j; read_revcomp_kmer: reversed, code ^ 2.
__global__ __launch_bounds__(kBlockThreads) void values_u64_kernel(SeqView seq, uint32_t len,
                                                                   int canonical,
                                                                   const uint32_t *__restrict__ pos,
                                                                   uint64_t n_pos,
                                                                   unsigned long long *__restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * kBlockThreads + threadIdx.x;
    if (i >= n_pos) return;
    long long p = (long long)seq.base0 + (long long)pos[i];
    long long q = p >> 4;
    uint32_t sh = 2u * (uint32_t)(p & 15);
    unsigned long long w0 = load_dword_clamped(seq, q);
    unsigned long long w1 = load_dword_clamped(seq, q + 1);
    unsigned long long w2 = load_dword_clamped(seq, q + 2);
    unsigned long long lo = w0 | (w1 << 32);
    unsigned long long v = sh ? (lo >> sh) | (w2 << (64u - sh)) : lo;
    const unsigned long long mask = len >= 32 ? ~0ull : ((1ull << (2u * len)) - 1ull);
    v &= mask;
    if (canonical) {
        unsigned long long r = __brevll(v);  // reverses bit order: pairs reversed and bit-swapped
        r = ((r & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((r & 0x5555555555555555ull) << 1);
        r >>= (64u - 2u * len);
        r ^= 0xAAAAAAAAAAAAAAAAull & mask;  // complement: code ^ 2
        v = r < v ? r : v;
    }
    out[i] = v;
}

int launch_values_u64(SeqView seq, uint32_t len, int canonical, const uint32_t *d_pos,
                      uint64_t n_pos, unsigned long long *d_values, hipStream_t stream) {
    if (n_pos == 0) return 0;
    uint32_t grid = (uint32_t)((n_pos + kBlockThreads - 1) / kBlockThreads);
    hipLaunchKernelGGL(values_u64_kernel, dim3(grid), dim3(kBlockThreads), 0, stream, seq, len,
                       canonical, d_pos, n_pos, d_values);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Output::values_u128 (src/lib.rs:587-629): up to 64 bases per value, stored as {lo, hi}.
__global__ __launch_bounds__(kBlockThreads) void values_u128_kernel(SeqView seq, uint32_t len,
                                                                    int canonical,
                                                                    const uint32_t *__restrict__ pos,
                                                                    uint64_t n_pos,
                                                                    unsigned long long *__restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * kBlockThreads + threadIdx.x;
    if (i >= n_pos) return;
    long long p = (long long)seq.base0 + (long long)pos[i];
    long long q = p >> 4;
    uint32_t sh = 2u * (uint32_t)(p & 15);
    unsigned long long w[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) w[t] = load_dword_clamped(seq, q + t);
    unsigned long long a = w[0] | (w[1] << 32), b = w[2] | (w[3] << 32);
    unsigned long long lo = sh ? (a >> sh) | (b << (64u - sh)) : a;
    unsigned long long hi = sh ? (b >> sh) | (w[4] << (64u - sh)) : b;
    const uint32_t bits = 2u * len;  // 2 .. 128
    if (bits <= 64) {
        hi = 0;
        if (bits < 64) lo &= (1ull << bits) - 1ull;
    } else if (bits < 128) {
        hi &= (1ull << (bits - 64u)) - 1ull;
    }
    if (canonical) {
        // reverse the 2-bit groups of the 128-bit value, align to bit 0, complement (code ^ 2)
        auto revpairs = [](unsigned long long x) {
            x = __brevll(x);
            return ((x & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((x & 0x5555555555555555ull) << 1);
        };
        unsigned long long rhi = revpairs(lo), rlo = revpairs(hi);  // 128-bit reversal
        const uint32_t s = 128u - bits;                              // shift right by s (0 .. 126)
        unsigned long long clo, chi;
        if (s == 0) { clo = rlo; chi = rhi; }
        else if (s < 64) { clo = (rlo >> s) | (rhi << (64u - s)); chi = rhi >> s; }
        else if (s == 64) { clo = rhi; chi = 0; }
        else { clo = rhi >> (s - 64u); chi = 0; }
        unsigned long long mlo = bits >= 64 ? ~0ull : (1ull << bits) - 1ull;
        unsigned long long mhi = bits <= 64 ? 0ull : (bits >= 128 ? ~0ull : (1ull << (bits - 64u)) - 1ull);
        clo ^= 0xAAAAAAAAAAAAAAAAull & mlo;
        chi ^= 0xAAAAAAAAAAAAAAAAull & mhi;
        if (chi < hi || (chi == hi && clo < lo)) { lo = clo; hi = chi; }
    }
    out[2 * i] = lo;
    out[2 * i + 1] = hi;
}

int launch_values_u128(SeqView seq, uint32_t len, int canonical, const uint32_t *d_pos,
                       uint64_t n_pos, unsigned long long *d_values, hipStream_t stream) {
    if (n_pos == 0) return 0;
    uint32_t grid = (uint32_t)((n_pos + kBlockThreads - 1) / kBlockThreads);
    hipLaunchKernelGGL(values_u128_kernel, dim3(grid), dim3(kBlockThreads), 0, stream, seq, len,
                       canonical, d_pos, n_pos, d_values);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// PackedSeqVec::from_ascii: code = (c >> 1) & 3 (A0 C1 T2 G3, case-insensitive), 4 bases per byte.
// 4 ASCII bytes -> 1 packed byte with SWAR: keep bits 1..2 of every byte, fold the four 2-bit
// fields together.
__device__ __forceinline__ uint32_t pack4(uint32_t x) {
    const uint32_t t = (x >> 1) & 0x03030303u;
    return (t | (t >> 6) | (t >> 12) | (t >> 18)) & 0xffu;
}

// fast path: one packed dword (16 bases) per thread from one aligned 16-byte load
__global__ __launch_bounds__(kBlockThreads) void pack_ascii16_kernel(const uint4 *__restrict__ ascii16,
                                                                     uint64_t n_groups,
                                                                     uint32_t *__restrict__ packed32) {
    uint64_t g = (uint64_t)blockIdx.x * kBlockThreads + threadIdx.x;
    if (g >= n_groups) return;
    const uint4 v = ascii16[g];
    packed32[g] = pack4(v.x) | (pack4(v.y) << 8) | (pack4(v.z) << 16) | (pack4(v.w) << 24);
}

// general path / tail: one output byte (4 bases) per thread
__global__ __launch_bounds__(kBlockThreads) void pack_ascii_kernel(const uint8_t *__restrict__ ascii,
                                                                   uint64_t first_byte, uint64_t n,
                                                                   uint8_t *__restrict__ packed) {
    uint64_t b = first_byte + (uint64_t)blockIdx.x * kBlockThreads + threadIdx.x;
    uint64_t nbytes = (n + 3) / 4;
    if (b >= nbytes) return;
    uint32_t v = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        uint64_t i = 4 * b + j;
        if (i < n) v |= (uint32_t)((ascii[i] >> 1) & 3u) << (2 * j);
    }
    packed[b] = (uint8_t)v;
}

int launch_pack_ascii(const uint8_t *d_ascii, uint64_t n, uint8_t *d_packed, hipStream_t stream) {
    uint64_t nbytes = (n + 3) / 4;
    if (nbytes == 0) return 0;
    uint64_t done_bytes = 0;
    const bool aligned = (reinterpret_cast<uintptr_t>(d_ascii) % 16 == 0) &&
                         (reinterpret_cast<uintptr_t>(d_packed) % 4 == 0);
    if (aligned && n >= 16) {
        const uint64_t groups = n / 16;  // whole 16-base groups
        uint32_t grid = (uint32_t)((groups + kBlockThreads - 1) / kBlockThreads);
        hipLaunchKernelGGL(pack_ascii16_kernel, dim3(grid), dim3(kBlockThreads), 0, stream,
                           reinterpret_cast<const uint4 *>(d_ascii), groups,
                           reinterpret_cast<uint32_t *>(d_packed));
        done_bytes = groups * 4;
    }
    if (done_bytes < nbytes) {
        uint32_t grid = (uint32_t)((nbytes - done_bytes + kBlockThreads - 1) / kBlockThreads);
        hipLaunchKernelGGL(pack_ascii_kernel, dim3(grid), dim3(kBlockThreads), 0, stream, d_ascii,
                           done_bytes, n, d_packed);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

__device__ __forceinline__ unsigned long long splitmix_final(unsigned long long z) {
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

// generator G of BASELINE.md §4: code[i] = mix(i + seed*g + g) >> 62
__global__ __launch_bounds__(kBlockThreads) void generate_kernel(unsigned long long seed,
                                                                 unsigned long long first_base,
                                                                 uint64_t n,
                                                                 uint8_t *__restrict__ packed) {
    uint64_t b = (uint64_t)blockIdx.x * kBlockThreads + threadIdx.x;
    uint64_t nbytes = (n + 3) / 4;
    if (b >= nbytes) return;
    const unsigned long long g = 0x9E3779B97F4A7C15ull;
    uint32_t v = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        uint64_t i = 4 * b + j;
        if (i < n) v |= (uint32_t)(splitmix_final(first_base + i + seed * g + g) >> 62) << (2 * j);
    }
    packed[b] = (uint8_t)v;
}

int launch_generate(uint64_t seed, uint64_t first_base, uint64_t n, uint8_t *d_packed,
                    hipStream_t stream) {
    uint64_t nbytes = (n + 3) / 4;
    if (nbytes == 0) return 0;
    uint32_t grid = (uint32_t)((nbytes + kBlockThreads - 1) / kBlockThreads);
    hipLaunchKernelGGL(generate_kernel, dim3(grid), dim3(kBlockThreads), 0, stream, seed, first_base,
                       n, d_packed);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace mm
