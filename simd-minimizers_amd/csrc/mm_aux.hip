// mm_aux.hip — the steps either side of the hot path: k-mer values of the sampled positions
// (Output::values_u64, src/lib.rs:584-612), ASCII -> PackedSeq packing
// (PackedSeqVec::from_ascii, call site src/lib.rs:110) and the synthetic input generator.
#include "mm_common.h"
#include "mm_launch.h"

namespace mm {

// packed-seq read_kmer: base j of the k-mer at bits 2j; read_revcomp_kmer: reversed, code ^ 2.
//
// Round 4: FOUR values per thread.  The round-3 kernel made one value per thread - one 4-byte position load, three
// dependent clamped dword loads, one 8-byte store: too few bytes in flight per wave to stream (3.2 TB/s of its 12 bytes
// per value where pack_ascii reaches 5.4 on the same chip).  Now a thread loads four consecutive positions with one
// 16-byte load, issues the four 16-byte sequence loads together (bounds-checked raw buffer loads: dwords past the end
// read as 0, and every bit a valid position needs lies inside the sequence, so no clamping arithmetic), funnel-shifts
// with v_alignbit, and stores the four values as two 16-byte stores - 32 contiguous bytes per lane.  Positions are in
// window order, so the 256 positions of a wave fall into a few hundred bytes of sequence: the sequence loads hit the
// same two or three lines.
constexpr int kValuesPerThread = 4;

__device__ __forceinline__ unsigned long long value_of(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t sh, uint32_t len,
                                                       int canonical, unsigned long long mask) {
    const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh), hi = __builtin_amdgcn_alignbit(w2, w1, sh);
    unsigned long long v = (((unsigned long long)hi << 32) | lo) & mask;
    if (canonical) {
        unsigned long long r = __brevll(v);  // reverses bit order: pairs reversed and bit-swapped
        r = ((r & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((r & 0x5555555555555555ull) << 1);
        r >>= (64u - 2u * len);
        r ^= 0xAAAAAAAAAAAAAAAAull & mask;  // complement: code ^ 2
        v = r < v ? r : v;
    }
    return v;
}

__global__ __launch_bounds__(kBlockThreads) void values_u64_kernel(SeqView seq, uint32_t len,
                                                                   int canonical,
                                                                   const uint32_t *__restrict__ pos,
                                                                   uint64_t n_pos,
                                                                   unsigned long long *__restrict__ out) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    constexpr uint64_t kPerBlock = (uint64_t)kBlockThreads * kValuesPerThread;
    const uint64_t i0 = (uint64_t)blockIdx.x * kPerBlock;  // first value of the block
    const uint64_t left = n_pos - i0;
    const uint32_t here = left < kPerBlock ? (uint32_t)left : (uint32_t)kPerBlock;
    // block-local bounds-checked views: positions in, values out (lanes past the end load 0 / store nothing)
    const __amdgpu_buffer_rsrc_t rpos = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(pos + i0), 0, (int)(here * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(out + i0, 0, (int)(here * 8u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rseq = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(seq.d), 0, seq.n_dwords >= 0x3fffffffu ? (int)0xfffffffcu : (int)(seq.n_dwords * 4u), 0x00020000);
    const uint32_t t = threadIdx.x;
    // (the position array may start at any 4-byte boundary: a 16-byte load needs no more)
    const u32x4 pp = __builtin_amdgcn_raw_buffer_load_b128(rpos, t * 16u, 0, 0);
    const uint32_t ps[4] = {pp.x, pp.y, pp.z, pp.w};
    const unsigned long long mask = len >= 32 ? ~0ull : ((1ull << (2u * len)) - 1ull);
    u32x4 w[kValuesPerThread];
    uint32_t sh[kValuesPerThread];
#pragma unroll
    for (int u = 0; u < kValuesPerThread; ++u) {
        const unsigned long long p = (unsigned long long)seq.base0 + ps[u];
        sh[u] = 2u * (uint32_t)(p & 15u);
        // (a sequence of 2^32 bases holds 2^28 dwords = 2^30 bytes: the byte offset fits 32 bits)
        w[u] = __builtin_amdgcn_raw_buffer_load_b128(rseq, (uint32_t)(p >> 4) * 4u, 0, 0);
    }
    unsigned long long v[kValuesPerThread];
#pragma unroll
    for (int u = 0; u < kValuesPerThread; ++u) v[u] = value_of(w[u].x, w[u].y, w[u].z, sh[u], len, canonical, mask);
#pragma unroll
    for (int u = 0; u < kValuesPerThread; u += 2) {
        u32x4 o;
        o.x = (uint32_t)v[u];
        o.y = (uint32_t)(v[u] >> 32);
        o.z = (uint32_t)v[u + 1];
        o.w = (uint32_t)(v[u + 1] >> 32);
        // (the last pair of the array may be half inside: the bounds check works per dword, so its inner half is stored)
        __builtin_amdgcn_raw_buffer_store_b128(o, rout, t * 32u + (uint32_t)u * 8u, 0, 0);
    }
}

int launch_values_u64(SeqView seq, uint32_t len, int canonical, const uint32_t *d_pos,
                      uint64_t n_pos, unsigned long long *d_values, hipStream_t stream) {
    if (n_pos == 0) return 0;
    const uint64_t per_block = (uint64_t)kBlockThreads * kValuesPerThread;
    uint32_t grid = (uint32_t)((n_pos + per_block - 1) / per_block);
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

// ------------------------------------------------------------------ PackedNSeq
// Window ambiguity bits for the skip-ambiguous path (src/minimizers.rs:203-212: the stream
// par_iter_kmer_ambiguity(l, ..) zipped with the positions): bit i of `out` = some base of
// window i, i.e. of bases [i, i + l), is ambiguous.  One output dword (32 windows) per thread:
//   window j of the dword is ambiguous iff an N sits in [32t + j, 32t + j + l):
//   l >= 32: the bases [32t+32, 32t+l) are common to all 32 windows; otherwise only an N at bit
//   b of the first dword (windows j <= b) or at bit r of the 32 bits from 32t+l on (windows
//   j > r) matters.   l < 32: OR of l shifted copies of a 64-bit view.
__device__ __forceinline__ uint32_t bit_view32(const uint32_t *__restrict__ a, uint32_t n_dwords,
                                               unsigned long long bit) {
    const unsigned long long idx = bit >> 5;
    const uint32_t sh = (uint32_t)bit & 31u;
    const uint32_t lo = idx < n_dwords ? a[idx] : 0u;
    const uint32_t hi = idx + 1 < n_dwords ? a[idx + 1] : 0u;
    return sh ? (lo >> sh) | (hi << (32u - sh)) : lo;
}

__global__ __launch_bounds__(kBlockThreads) void window_ambiguity_kernel(
    const uint32_t *__restrict__ amb, uint32_t amb_dwords, unsigned long long bit0, uint32_t l,
    uint64_t first_dword, uint64_t end_dword, uint32_t *__restrict__ out) {
    const uint64_t t = first_dword + (uint64_t)blockIdx.x * kBlockThreads + threadIdx.x;
    if (t >= end_dword) return;
    const unsigned long long b = bit0 + 32ull * t;
    uint32_t r;
    if (l < 32u) {
        const unsigned long long v =
            (unsigned long long)bit_view32(amb, amb_dwords, b) |
            ((unsigned long long)bit_view32(amb, amb_dwords, b + 32) << 32);
        // OR of v >> 0 .. v >> (l - 1) by doubling: five or six shifts instead of l
        unsigned long long acc = v;
        uint32_t cover = 1;
        while (2u * cover <= l) {
            acc |= acc >> cover;
            cover *= 2u;
        }
        if (cover < l) acc |= acc >> (l - cover);
        r = (uint32_t)acc;
    } else {
        uint32_t any = 0;
        for (uint32_t done = 32; done < l; done += 32) {
            uint32_t v = bit_view32(amb, amb_dwords, b + done);
            if (l - done < 32u) v &= (1u << (l - done)) - 1u;
            any |= v;
        }
        if (any) {
            r = 0xffffffffu;
        } else {
            const uint32_t a0 = bit_view32(amb, amb_dwords, b);
            const uint32_t a1 = bit_view32(amb, amb_dwords, b + l);
            const uint32_t left = a0 ? (0xffffffffu >> __clz(a0)) : 0u;          // windows j <= msb(a0)
            const uint32_t right = a1 ? ~((2u << (__ffs(a1) - 1)) - 1u) : 0u;    // windows j > lsb(a1)
            r = left | right;
        }
    }
    out[t] = r;
}

int launch_window_ambiguity(const uint32_t *d_amb, uint32_t amb_dwords, uint64_t bit0, uint32_t l,
                            uint64_t win_begin, uint64_t win_end, uint32_t *d_out, hipStream_t stream) {
    if (win_begin >= win_end) return 0;
    // one extra window in front: the dedup predecessor of win_begin
    const uint64_t first = (win_begin ? win_begin - 1 : 0) / 32, end = (win_end + 31) / 32;
    uint32_t grid = (uint32_t)((end - first + kBlockThreads - 1) / kBlockThreads);
    hipLaunchKernelGGL(window_ambiguity_kernel, dim3(grid), dim3(kBlockThreads), 0, stream, d_amb,
                       amb_dwords, (unsigned long long)bit0, l, first, end, d_out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// PackedNSeqVec::from_ascii (packed-seq; call site src/test.rs:436): lossy 2-bit codes plus one
// ambiguity bit per base (set for everything that is not ACGT / acgt).
// bit 7 of every byte of the result is set where the byte of x is NOT one of A C G T (any case)
__device__ __forceinline__ uint32_t non_acgt4(uint32_t x) {
    x &= 0xDFDFDFDFu;  // upper case
    uint32_t is = 0;
#pragma unroll
    for (uint32_t c : {0x41414141u, 0x43434343u, 0x47474747u, 0x54545454u}) {
        const uint32_t z = x ^ c;  // zero byte <=> match
        const uint32_t t = ((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z;
        is |= ~t;
    }
    return ~is & 0x80808080u;
}
__device__ __forceinline__ uint32_t movemask4(uint32_t m) {  // bit 7 of byte i -> bit i
    const uint32_t b = m >> 7;
    return (b | (b >> 7) | (b >> 14) | (b >> 21)) & 0xfu;
}

// fast path: 32 bases per thread from two aligned 16-byte loads
__global__ __launch_bounds__(kBlockThreads) void pack_ascii_n32_kernel(const uint4 *__restrict__ ascii16,
                                                                       uint64_t n_groups,
                                                                       uint2 *__restrict__ packed64,
                                                                       uint32_t *__restrict__ amb32) {
    uint64_t g = (uint64_t)blockIdx.x * kBlockThreads + threadIdx.x;
    if (g >= n_groups) return;
    const uint4 a = ascii16[2 * g], b = ascii16[2 * g + 1];
    uint2 p;
    p.x = pack4(a.x) | (pack4(a.y) << 8) | (pack4(a.z) << 16) | (pack4(a.w) << 24);
    p.y = pack4(b.x) | (pack4(b.y) << 8) | (pack4(b.z) << 16) | (pack4(b.w) << 24);
    packed64[g] = p;
    amb32[g] = movemask4(non_acgt4(a.x)) | (movemask4(non_acgt4(a.y)) << 4) |
               (movemask4(non_acgt4(a.z)) << 8) | (movemask4(non_acgt4(a.w)) << 12) |
               (movemask4(non_acgt4(b.x)) << 16) | (movemask4(non_acgt4(b.y)) << 20) |
               (movemask4(non_acgt4(b.z)) << 24) | (movemask4(non_acgt4(b.w)) << 28);
}

// general path / tail: 8 bases per thread -> two packed bytes and one ambiguity byte
__global__ __launch_bounds__(kBlockThreads) void pack_ascii_n_kernel(const uint8_t *__restrict__ ascii,
                                                                     uint64_t first_group, uint64_t n,
                                                                     uint8_t *__restrict__ packed,
                                                                     uint8_t *__restrict__ amb) {
    const uint64_t g = first_group + (uint64_t)blockIdx.x * kBlockThreads + threadIdx.x;
    if (8 * g >= n) return;
    uint32_t pk = 0, am = 0;
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) {
        const uint64_t i = 8 * g + j;
        if (i < n) {
            const uint32_t c = ascii[i];
            pk |= ((c >> 1) & 3u) << (2 * j);
            const uint32_t u = c & 0xDFu;
            am |= (uint32_t)!(u == 'A' || u == 'C' || u == 'G' || u == 'T') << j;
        }
    }
    packed[2 * g] = (uint8_t)pk;
    if (8 * g + 4 < n) packed[2 * g + 1] = (uint8_t)(pk >> 8);
    amb[g] = (uint8_t)am;
}

int launch_pack_ascii_n(const uint8_t *d_ascii, uint64_t n, uint8_t *d_packed, uint8_t *d_amb,
                        hipStream_t stream) {
    if (n == 0) return 0;
    uint64_t done_groups8 = 0;  // groups of 8 bases finished by the fast path
    const bool aligned = (reinterpret_cast<uintptr_t>(d_ascii) % 16 == 0) &&
                         (reinterpret_cast<uintptr_t>(d_packed) % 8 == 0) &&
                         (reinterpret_cast<uintptr_t>(d_amb) % 4 == 0);
    if (aligned && n >= 32) {
        const uint64_t groups = n / 32;
        uint32_t grid = (uint32_t)((groups + kBlockThreads - 1) / kBlockThreads);
        hipLaunchKernelGGL(pack_ascii_n32_kernel, dim3(grid), dim3(kBlockThreads), 0, stream,
                           reinterpret_cast<const uint4 *>(d_ascii), groups,
                           reinterpret_cast<uint2 *>(d_packed), reinterpret_cast<uint32_t *>(d_amb));
        done_groups8 = groups * 4;
    }
    const uint64_t groups8 = (n + 7) / 8;
    if (done_groups8 < groups8) {
        uint32_t grid = (uint32_t)((groups8 - done_groups8 + kBlockThreads - 1) / kBlockThreads);
        hipLaunchKernelGGL(pack_ascii_n_kernel, dim3(grid), dim3(kBlockThreads), 0, stream, d_ascii,
                           done_groups8, n, d_packed, d_amb);
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

// Shader clock under load (diagnostics, mm_clock_probe_*): a few single-wave workgroups that do nothing but sleep
// sample the shader cycle counter (s_memtime) against the 100 MHz real-time counter (s_memrealtime) for `ticks`
// real-time ticks while whatever else runs on the chip; out[2b] = shader cycles, out[2b + 1] = real-time ticks.
__global__ __launch_bounds__(kWave) void clock_probe_kernel(unsigned long long *out, unsigned long long ticks) {
    unsigned long long c0, r0, c1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0));
    do {
        __builtin_amdgcn_s_sleep(127);
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1));
    } while (r1 - r0 < ticks);
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = c1 - c0;
        out[2 * blockIdx.x + 1] = r1 - r0;
    }
}

int launch_clock_probe(unsigned long long *d_out, uint32_t workgroups, uint64_t ticks, hipStream_t stream) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(workgroups), dim3(kWave), 0, stream, d_out, (unsigned long long)ticks);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- copies between device memory and page-locked HOST memory done by a kernel instead of a copy engine (the host
// entry point's alternative mechanisms, mm_api.hip run_host_pipelined; measured against the engines by
// tools/ubench/link_duplex.hip).  Both keep to a few workgroups: the link moves ~55 GB/s, which 64 workgroups saturate,
// and the fused kernel runs beside them.
typedef uint32_t copy_v4u __attribute__((ext_vector_type(4)));

// dst[lo .. hi) = src[lo .. hi) in dwords, the range read from DEVICE memory (range[0], range[1]: the running totals
// of two consecutive chunk kernels), hi cut to cap: the device -> host leg of a chunk with no host round trip.
// 16-byte stores aligned to dst (full PCIe write payloads), the few dwords either side of them singly.
__global__ __launch_bounds__(kBlockThreads) void copy_range_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst,
                                                                   const unsigned long long *__restrict__ range,
                                                                   unsigned long long cap) {
    unsigned long long lo = range[0], hi = range[1];
    hi = hi < cap ? hi : cap;
    if (lo >= hi) return;
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long nthreads = (unsigned long long)gridDim.x * blockDim.x;
    // first dword index >= lo at which dst is 16-byte aligned
    const unsigned long long mis = (reinterpret_cast<uintptr_t>(dst + lo) >> 2) & 3ull;
    unsigned long long body = lo + ((4ull - mis) & 3ull);
    if (body > hi) body = hi;
    const unsigned long long n16 = (hi - body) >> 2, tail = body + (n16 << 2);
    if (tid < body - lo) __builtin_nontemporal_store(src[lo + tid], &dst[lo + tid]);
    if (tid < hi - tail) __builtin_nontemporal_store(src[tail + tid], &dst[tail + tid]);
    const uint32_t *s = src + body;
    copy_v4u *d16 = reinterpret_cast<copy_v4u *>(dst + body);
    for (unsigned long long i = tid; i < n16; i += nthreads) {
        copy_v4u v;  // (src is only dword aligned relative to dst: four dword loads, one 16-byte store)
        v.x = s[4 * i], v.y = s[4 * i + 1], v.z = s[4 * i + 2], v.w = s[4 * i + 3];
        __builtin_nontemporal_store(v, &d16[i]);
    }
}

int launch_copy_range(const uint32_t *d_src, uint32_t *dst_host_alias, const unsigned long long *d_range, uint64_t cap,
                      uint32_t workgroups, hipStream_t stream) {
    hipLaunchKernelGGL(copy_range_kernel, dim3(workgroups), dim3(kBlockThreads), 0, stream, d_src, dst_host_alias, d_range,
                       (unsigned long long)cap);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// dst[0 .. n16) = src[0 .. n16) in 16-byte units, both 16-byte aligned: the host -> device leg (src = the device's
// address of page-locked host memory)
__global__ __launch_bounds__(kBlockThreads) void copy16_kernel(const copy_v4u *__restrict__ src, copy_v4u *__restrict__ dst,
                                                               unsigned long long n16) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16;
         i += (unsigned long long)gridDim.x * blockDim.x)
        dst[i] = __builtin_nontemporal_load(&src[i]);
}

int launch_copy16(const void *src, void *dst, uint64_t n16, uint32_t workgroups, hipStream_t stream) {
    hipLaunchKernelGGL(copy16_kernel, dim3(workgroups), dim3(kBlockThreads), 0, stream,
                       reinterpret_cast<const copy_v4u *>(src), reinterpret_cast<copy_v4u *>(dst), (unsigned long long)n16);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace mm
