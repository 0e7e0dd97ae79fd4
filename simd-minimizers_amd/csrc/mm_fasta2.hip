// mm_fasta2.hip — FASTA text -> PackedSeq records on the device in TWO passes of mask arithmetic (round 4, late; the
// default since).  Same semantics and same outputs as mm_fasta.hip (needletail's reader restated, parity unpinned, see
// there): a record starts with '>' at the start of a line, its header runs to the end of that line, its sequence is
// every following line up to the next header line with '\n' / '\r' removed, bytes before the first header are ignored;
// all records back to back in one 2-bit buffer, record r = bases [rec_base[r], rec_base[r + 1]).
//
// What a byte is depends on two bits of state that come from the text in front of it: h (the line it lies in is a
// header line) and st (a record has started).  A piece of text acts on that state as a FUNCTION with a closed form
// (the one the one-pass kernel's look-back publishes, mm_fasta.hip): its sequence bytes in three classes - U in front of
// its first line start (they count iff st and not h), V between the first line start and the first record start (iff
// st), K behind the first record start (always) - its record starts, kind (0: holds no line start, 1: its last line
// start is no header, 2: it is) and whether it holds a record start.  Such functions compose associatively:
//   A then B:  K = K_A + K_B + [rec_A] (V_B + [not h_A] U_B)            (h_A = (kind_A == 2))
//              V = V_A + [not rec_A] (V_B + [kind_A != 0][not h_A] U_B)
//              U = U_A + [not rec_A][kind_A == 0] U_B
//              kind = kind_B ? kind_B : kind_A,  rec = rec_A | rec_B,  records add
// so the chunks need no look-back and no context pass:
//   K1  per 16 KB chunk: its function (one 64-bit word)
//   R1  one workgroup per 256 chunks: the composition of the chunks in front of every chunk within its group (a
//       workgroup scan with the operator above) and of the whole group
//   R2  one workgroup: the groups one after the other from the start of the text (h = st = 0): every group's state, first
//       output base and first record
//   K2  per chunk: state, first base and first record from its group's and its own prefix; packs.
// Per thread: 32 bytes per piece as 32-bit masks (mm_text.h); header bytes by ONE addition (a carry put on every header
// line start runs through the bytes that are not line starts); a thread's sequence bytes are one or two runs (a line end
// in the middle), shifted together; the chunk's output is assembled in LDS and leaves as whole dwords.  No tables
// with limits, no order of execution between workgroups, no fallback: the one-pass kernel (0.87 ms per GiB of 60-base
// lines) and the three-pass kernels (1.66 ms) stay as cross-checks (MM_FASTA_KERNEL=lines / three).
#include "mm_common.h"
#include "mm_launch.h"
#include "mm_text.h"

namespace mm {

namespace {

// ------------------------------------------------------------------------------------------------ the function
struct FaFn {
    uint32_t K, V, U, nrec, kind, rec;
};
__device__ __forceinline__ FaFn fa_identity() { return FaFn{0u, 0u, 0u, 0u, 0u, 0u}; }
__device__ __forceinline__ FaFn fa_compose(const FaFn &a, const FaFn &b) {  // a in front of b
    const bool ha = a.kind == 2u;
    FaFn r;
    r.K = a.K + b.K + (a.rec ? b.V + (ha ? 0u : b.U) : 0u);
    r.V = a.V + (a.rec ? 0u : b.V + ((a.kind != 0u && !ha) ? b.U : 0u));
    r.U = a.U + ((!a.rec && a.kind == 0u) ? b.U : 0u);
    r.nrec = a.nrec + b.nrec;
    r.kind = b.kind ? b.kind : a.kind;
    r.rec = a.rec | b.rec;
    return r;
}
// one chunk's function in one word (K1 -> R1): K, V, U below 2^15, records below 2^14
__device__ __forceinline__ unsigned long long fa_word(const FaFn &f) {
    return (unsigned long long)f.K | ((unsigned long long)f.V << 15) | ((unsigned long long)f.U << 30) |
           ((unsigned long long)f.nrec << 45) | ((unsigned long long)f.kind << 59) | ((unsigned long long)f.rec << 61);
}
__device__ __forceinline__ FaFn fa_unword(unsigned long long w) {
    return FaFn{(uint32_t)w & 0x7fffu, (uint32_t)(w >> 15) & 0x7fffu, (uint32_t)(w >> 30) & 0x7fffu, (uint32_t)(w >> 45) & 0x3fffu,
                (uint32_t)(w >> 59) & 3u, (uint32_t)(w >> 61) & 1u};
}
// up to 256 chunks' composition in two words (R1 -> R2, K2): K, V, U below 2^23, records below 2^22
__device__ __forceinline__ void fa_words2(const FaFn &f, unsigned long long *w0, unsigned long long *w1) {
    *w0 = (unsigned long long)f.K | ((unsigned long long)f.V << 32);
    *w1 = (unsigned long long)f.U | ((unsigned long long)f.nrec << 23) | ((unsigned long long)f.kind << 60) |
          ((unsigned long long)f.rec << 62);
}
__device__ __forceinline__ FaFn fa_unwords2(unsigned long long w0, unsigned long long w1) {
    return FaFn{(uint32_t)w0, (uint32_t)(w0 >> 32), (uint32_t)w1 & 0x7fffffu, (uint32_t)(w1 >> 23) & 0x3fffffu,
                (uint32_t)(w1 >> 60) & 3u, (uint32_t)(w1 >> 62) & 1u};
}

// ------------------------------------------------------------------------------------------------ a chunk's threads
// Header bytes of a piece: every line start carries its own answer (it is a record start or it is not), every other
// byte inherits the answer of the byte before it, the first byte inherits h0.  One addition does the segmented fill:
// a carry on every header line start runs through the bytes that are not line starts (bit 0 = the byte before the piece).
__device__ __forceinline__ uint32_t header_mask32(uint32_t rs, uint32_t ls, uint32_t h0) {
    const unsigned long long m = 0x1ffffffffull;
    const unsigned long long a = ((unsigned long long)rs << 1) | h0, u = ((unsigned long long)ls << 1) | 1ull;
    const unsigned long long z = (~u | a) & m, r = a + z;
    return (uint32_t)((((r ^ z) | a) & z) >> 1);
}

// '\n' mask of the thread's 32 bytes (as eq32) and, OR-ed into `special`, a flag for every byte below 0x40 (or in
// 0x80..0xbf) that is NOT a '\n': sequence lines hold none, so a wave without one holds neither a '\r' nor a '>' and
// skips their exact masks - three operations per dword where the two exact yes / no tests took six.
__device__ __forceinline__ uint32_t newline_mask32(const Text32 &v, uint32_t &special) {
    uint32_t mask = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t x = v.d[i], z = x ^ 0x0a0a0a0au;
        const uint32_t t = (z & 0x7f7f7f7fu) + 0x7f7f7f7fu;
        const uint32_t m = ~(t | z | 0x7f7f7f7fu);  // 0x80 where the byte is '\n'
        special |= ~x & 0x40404040u & ~(m >> 1);
        mask |= ((m * 0x00204081u) >> 28) << (4 * i);
    }
    return mask;
}

constexpr uint32_t kUnknown = 3u;  // "no line start / no record start in front of this inside the chunk"

// what both kernels start with: the chunk's text and every thread's class masks per piece
struct FaChunk {
    Text32 v[kFqPieces];
    uint32_t ub[kFqPieces], vb[kFqPieces], kb[kFqPieces];  // sequence-byte candidates of class U / V / K
    uint32_t rs[kFqPieces];                                // record starts
    uint32_t kind;                                         // of the whole chunk (uniform)
    uint32_t rec;
};
__device__ __forceinline__ void fa_read_chunk(const uint8_t *text, uint64_t n, uint64_t c0, uint32_t (*s_ctx)[kFqWaves], FaChunk &c) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
    for (int p = 0; p < (int)kFqPieces; ++p) c.v[p] = load32(text, n, c0 + (uint64_t)p * kFqPiece, threadIdx.x);
    const uint32_t left = n - c0 > 0x7fffffffull ? 0x7fffffffu : (uint32_t)(n - c0);
    uint32_t nl[kFqPieces], valid[kFqPieces], ls[kFqPieces], ctx_line[kFqPieces], ctx_rec[kFqPieces];
    // ('\r' and '>' are rare: their exact masks only in waves that hold a byte that could be one)
    uint32_t special = 0;
#pragma unroll
    for (int p = 0; p < (int)kFqPieces; ++p) nl[p] = newline_mask32(c.v[p], special);
    const bool has_cr = __ballot(special != 0u) != 0ull, has_gt = has_cr;
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;  // lanes in front of this one
#pragma unroll
    for (int p = 0; p < (int)kFqPieces; ++p) {
        const uint32_t rel = (uint32_t)p * kFqPiece + threadIdx.x * kFqBytesPerThread;
        const uint32_t nin = rel >= left ? 0u : (left - rel >= 32u ? 32u : left - rel);
        const uint32_t inside = nin >= 32u ? 0xffffffffu : ((1u << nin) - 1u);
        const bool sl = starts_line_of(text, n, c0 + rel, c.v[p]);
        nl[p] &= inside;
        const uint32_t cr = has_cr ? eq32(c.v[p], 0x0d0d0d0du) : 0u;
        const uint32_t gt = has_gt ? eq32(c.v[p], 0x3e3e3e3eu) : 0u;
        valid[p] = inside & ~nl[p] & ~cr;
        ls[p] = ((nl[p] << 1) | (sl ? 1u : 0u)) & inside;
        c.rs[p] = gt & ls[p];
        // the thread's own last line start: none / no header / header; does it hold a record start
        const uint32_t kind = ls[p] ? (((c.rs[p] >> (31 - __builtin_clz(ls[p]))) & 1u) ? 2u : 1u) : 0u;
        const unsigned long long Lb = __ballot(kind != 0u), Hb = __ballot(kind == 2u), Rb = __ballot(c.rs[p] != 0u);
        const unsigned long long lowL = Lb & below;
        ctx_line[p] = lowL ? (uint32_t)((Hb >> (63 - __builtin_clzll(lowL))) & 1ull) : kUnknown;
        ctx_rec[p] = (Rb & below) ? 1u : kUnknown;
        if (lane == 0) s_ctx[p][wave] = (Lb ? (((Hb >> (63 - __builtin_clzll(Lb))) & 1ull) ? 2u : 1u) : 0u) | (Rb ? 4u : 0u);
    }
    __syncthreads();
    // the wave's context from the wave-pieces in front of it (text order: piece-major); the chunk's own kind / rec
    uint32_t run_line = kUnknown, run_rec = kUnknown;
#pragma unroll
    for (int p = 0; p < (int)kFqPieces; ++p) {
#pragma unroll
        for (int w = 0; w < kFqWaves; ++w) {
            if (w == wave) {
                if (ctx_line[p] == kUnknown) ctx_line[p] = run_line;
                if (ctx_rec[p] == kUnknown) ctx_rec[p] = run_rec;
            }
            const uint32_t s = s_ctx[p][w];
            if (s & 3u) run_line = (s & 3u) == 2u ? 1u : 0u;
            if (s & 4u) run_rec = 1u;
        }
    }
    c.kind = run_line == kUnknown ? 0u : (run_line ? 2u : 1u);
    c.rec = run_rec == kUnknown ? 0u : 1u;
    __syncthreads();  // (s_ctx is reused by the callers' sums)
    // (header bytes exist only where a wave holds a '>' or continues a header line: a genome has a few dozen such waves)
    const bool any_hdr = has_gt || __ballot(ctx_line[0] == 1u || ctx_line[1] == 1u) != 0ull;
#pragma unroll
    for (int p = 0; p < (int)kFqPieces; ++p) {
        const uint32_t hdr = any_hdr ? header_mask32(c.rs[p], ls[p], ctx_line[p] == 1u ? 1u : 0u) : 0u;
        const uint32_t nonhdr = valid[p] & ~hdr;
        const uint32_t bls = ls[p] ? ((ls[p] & (0u - ls[p])) - 1u) : 0xffffffffu;        // bytes in front of the first line start
        const uint32_t brs = c.rs[p] ? ((c.rs[p] & (0u - c.rs[p])) - 1u) : 0xffffffffu;  // ... of the first record start
        const uint32_t ureg = ctx_line[p] == kUnknown ? bls : 0u;
        const uint32_t vreg = ctx_rec[p] == kUnknown ? brs : 0u;
        c.ub[p] = valid[p] & ureg;
        c.vb[p] = nonhdr & vreg & ~ureg;
        c.kb[p] = nonhdr & ~vreg;
    }
}

// K1: the chunk's function
__global__ __launch_bounds__(kFqThreads) void fasta2_count_kernel(const uint8_t *__restrict__ text, uint64_t n,
                                                                  unsigned long long *__restrict__ fn) {
    __shared__ uint32_t s_ctx[kFqPieces][kFqWaves];
    __shared__ unsigned long long s_red[kFqWaves];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    FaChunk c;
    fa_read_chunk(text, n, (uint64_t)blockIdx.x * kFqChunk, s_ctx, c);
    unsigned long long a = 0;  // K | V << 16 | U << 32 | records << 48 (a thread's fields stay below 65, the chunk's below 2^15)
#pragma unroll
    for (int p = 0; p < (int)kFqPieces; ++p)
        a += (unsigned long long)__popc(c.kb[p]) | ((unsigned long long)__popc(c.vb[p]) << 16) |
             ((unsigned long long)__popc(c.ub[p]) << 32) | ((unsigned long long)__popc(c.rs[p]) << 48);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d, kWave);
    if (lane == 0) s_red[wave] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
#pragma unroll
        for (int w = 0; w < kFqWaves; ++w) t += s_red[w];
        const FaFn f{(uint32_t)t & 0xffffu, (uint32_t)(t >> 16) & 0xffffu, (uint32_t)(t >> 32) & 0xffffu, (uint32_t)(t >> 48),
                     c.kind, c.rec};
        fn[blockIdx.x] = fa_word(f);
    }
}

// R1: one workgroup per group of 256 chunks: what lies in front of every chunk within its group, and the whole group
constexpr uint32_t kFaGroup = 256;
struct FaScratch {
    unsigned long long *fn;              // [chunks]      K1: the chunk's function
    unsigned long long *pre0, *pre1;     // [chunks]      R1: the composition of the chunks in front of it within its group
    unsigned long long *grp0, *grp1;     // [groups]      R1: the whole group
    unsigned long long *g_base, *g_rec;  // [groups + 1]  R2: first output base / first record of the group ([groups]: totals)
    uint32_t *g_state;                   // [groups]      R2: h | st << 1 in front of the group
};
__device__ __forceinline__ FaFn fa_shfl_up(const FaFn &f, int d) {
    return FaFn{(uint32_t)__shfl_up((int)f.K, d, kWave), (uint32_t)__shfl_up((int)f.V, d, kWave), (uint32_t)__shfl_up((int)f.U, d, kWave),
                (uint32_t)__shfl_up((int)f.nrec, d, kWave), (uint32_t)__shfl_up((int)f.kind, d, kWave),
                (uint32_t)__shfl_up((int)f.rec, d, kWave)};
}
__global__ __launch_bounds__(kFaGroup) void fasta2_groups_kernel(FaScratch sc, uint64_t chunks) {
    __shared__ FaFn s_wave[kFaGroup / kWave];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const uint64_t c = (uint64_t)blockIdx.x * kFaGroup + threadIdx.x;
    const FaFn mine = c < chunks ? fa_unword(sc.fn[c]) : fa_identity();
    FaFn incl = mine;  // inclusive composition over the wave's lanes, in order
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const FaFn o = fa_shfl_up(incl, d);
        if (lane >= d) incl = fa_compose(o, incl);
    }
    FaFn excl = fa_shfl_up(incl, 1);
    if (lane == 0) excl = fa_identity();
    if (lane == kWave - 1) s_wave[wave] = incl;
    __syncthreads();
    FaFn before = fa_identity(), total = fa_identity();
#pragma unroll
    for (int w = 0; w < (int)(kFaGroup / kWave); ++w) {
        if (w < wave) before = fa_compose(before, s_wave[w]);
        total = fa_compose(total, s_wave[w]);
    }
    if (c < chunks) fa_words2(fa_compose(before, excl), &sc.pre0[c], &sc.pre1[c]);
    if (threadIdx.x == 0) fa_words2(total, &sc.grp0[blockIdx.x], &sc.grp1[blockIdx.x]);
}
// R2: the groups one after the other (a 1 GiB text has 256 of them)
__global__ __launch_bounds__(kFaGroup) void fasta2_resolve_kernel(FaScratch sc, uint64_t groups) {
    __shared__ unsigned long long s0[kFaGroup], s1[kFaGroup];
    __shared__ unsigned long long s_base[kFaGroup], s_rec[kFaGroup];
    __shared__ uint32_t s_state[kFaGroup];
    __shared__ unsigned long long s_carry[3];
    if (threadIdx.x == 0) s_carry[0] = s_carry[1] = s_carry[2] = 0;
    __syncthreads();
    for (uint64_t g0 = 0; g0 < groups; g0 += kFaGroup) {
        const uint64_t g = g0 + threadIdx.x;
        s0[threadIdx.x] = g < groups ? sc.grp0[g] : 0ull;
        s1[threadIdx.x] = g < groups ? sc.grp1[g] : 0ull;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long base = s_carry[0], recs = s_carry[1];
            uint32_t h = (uint32_t)s_carry[2] & 1u, st = ((uint32_t)s_carry[2] >> 1) & 1u;
            const uint32_t m = groups - g0 < kFaGroup ? (uint32_t)(groups - g0) : kFaGroup;
            for (uint32_t i = 0; i < m; ++i) {
                s_base[i] = base;
                s_rec[i] = recs;
                s_state[i] = h | (st << 1);
                const FaFn f = fa_unwords2(s0[i], s1[i]);
                base += f.K + (st ? f.V + (h ? 0u : f.U) : 0u);
                recs += f.nrec;
                h = f.kind ? (f.kind == 2u ? 1u : 0u) : h;
                st |= f.rec;
            }
            s_carry[0] = base;
            s_carry[1] = recs;
            s_carry[2] = h | (st << 1);
        }
        __syncthreads();
        if (g < groups) {
            sc.g_base[g] = s_base[threadIdx.x];
            sc.g_rec[g] = s_rec[threadIdx.x];
            sc.g_state[g] = s_state[threadIdx.x];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sc.g_base[groups] = s_carry[0];
        sc.g_rec[groups] = s_carry[1];
    }
}

// K2: pack
__global__ __launch_bounds__(kFqThreads) void fasta2_pack_kernel(const uint8_t *__restrict__ text, uint64_t n, const FaScratch sc,
                                                                 uint32_t *__restrict__ out32, uint64_t out_dwords,
                                                                 unsigned long long *__restrict__ rec_base,
                                                                 unsigned long long *__restrict__ rec_pos, uint64_t max_records) {
    __shared__ uint32_t s_part[kFqPieces][kFqWaves];
    // the chunk's output, assembled in LDS (see fastq_pack_kernel): at most 16 384 bases = 1024 dwords + 1 + 2
    constexpr uint32_t kOutDwords = kFqChunk / 16u + 4u;
    __shared__ uint32_t s_out[kOutDwords];
    for (uint32_t i = threadIdx.x; i < kOutDwords; i += kFqThreads) s_out[i] = 0u;  // (ordered by fa_read_chunk's barriers)
    const uint64_t c0 = (uint64_t)blockIdx.x * kFqChunk;
    const uint64_t grp = blockIdx.x / kFaGroup;
    const uint32_t gs = sc.g_state[grp], gh = gs & 1u, gst = (gs >> 1) & 1u;
    const FaFn pre = fa_unwords2(sc.pre0[blockIdx.x], sc.pre1[blockIdx.x]);
    const uint32_t h_in = pre.kind ? (pre.kind == 2u ? 1u : 0u) : gh, st_in = gst | pre.rec;
    const unsigned long long seq0 = sc.g_base[grp] + pre.K + (gst ? pre.V + (gh ? 0u : pre.U) : 0u);
    const unsigned long long rec0 = sc.g_rec[grp] + pre.nrec;
    FaChunk c;
    fa_read_chunk(text, n, c0, s_part, c);
    uint32_t seq_mask[kFqPieces], both[kFqPieces], both_before[kFqPieces];
#pragma unroll
    for (int p = 0; p < (int)kFqPieces; ++p) {
        seq_mask[p] = c.kb[p] | (st_in ? c.vb[p] : 0u) | ((st_in && !h_in) ? c.ub[p] : 0u);
        both[p] = (uint32_t)__popc(seq_mask[p]) | ((uint32_t)__popc(c.rs[p]) << 16);  // (sums stay below 2^16)
    }
    const uint32_t chunk_seq = chunk_exclusive(both, both_before, s_part) & 0xffffu;  // sequence bytes of the chunk
    const uint64_t q0 = seq0 >> 4;  // the chunk's first output dword
#pragma unroll
    for (int p = 0; p < (int)kFqPieces; ++p) {
        const unsigned long long o0 = seq0 + (both_before[p] & 0xffffu);  // first output base of the thread's piece
        const unsigned long long r0 = rec0 + (both_before[p] >> 16);
        const uint32_t sm = seq_mask[p];
        if (sm) {
            // 2-bit codes of the 32 bytes, byte i at bits 2i (one multiply per dword gathers four codes)
            auto codes8 = [](uint32_t x) { return (((x >> 1) & 0x03030303u) * 0x01041040u) >> 24; };
            const Text32 &v = c.v[p];
            const uint32_t clo = codes8(v.d[0]) | (codes8(v.d[1]) << 8) | (codes8(v.d[2]) << 16) | (codes8(v.d[3]) << 24);
            const uint32_t chi = codes8(v.d[4]) | (codes8(v.d[5]) << 8) | (codes8(v.d[6]) << 16) | (codes8(v.d[7]) << 24);
            const unsigned long long codes = (unsigned long long)clo | ((unsigned long long)chi << 32);
            // the rule: ONE run of bytes (a piece of one sequence line) or TWO (a line end inside the 32 bytes)
            const uint32_t f1 = (uint32_t)__builtin_ctz(sm), t1 = sm >> f1, l1 = t1 == 0xffffffffu ? 32u : (uint32_t)__builtin_ctz(~t1);
            const uint32_t rest = l1 + f1 >= 32u ? 0u : (sm >> (f1 + l1)) << (f1 + l1);
            auto field = [&](uint32_t first, uint32_t len) {
                return (codes >> (2u * first)) & (len >= 32u ? ~0ull : ((1ull << (2u * len)) - 1ull));
            };
            unsigned long long bits = field(f1, l1);
            if (rest) {
                const uint32_t f2 = (uint32_t)__builtin_ctz(rest), t2 = rest >> f2, l2 = t2 == 0xffffffffu ? 32u : (uint32_t)__builtin_ctz(~t2);
                const uint32_t rest2 = l2 + f2 >= 32u ? 0u : (rest >> (f2 + l2)) << (f2 + l2);
                if (rest2 == 0u) {
                    bits |= field(f2, l2) << (2u * l1);
                } else {  // (lines shorter than the piece, '\r' inside a line: byte by byte)
                    bits = 0;
                    uint32_t k = 0;
#pragma unroll
                    for (int i = 0; i < (int)kFqBytesPerThread; ++i)
                        if ((sm >> i) & 1u) {
                            bits |= ((codes >> (2 * i)) & 3ull) << (2u * k);
                            ++k;
                        }
                }
            }
            const uint32_t q = (uint32_t)((o0 >> 4) - q0);
            const uint32_t sh = 2u * (uint32_t)(o0 & 15ull);
            const unsigned long long lo = bits << sh;                              // (64 bits shifted by at most 30:
            const uint32_t top = sh ? (uint32_t)(bits >> (64u - sh)) : 0u;         //  96 bits over three dwords)
            if ((uint32_t)lo) atomicOr(&s_out[q], (uint32_t)lo);
            if ((uint32_t)(lo >> 32)) atomicOr(&s_out[q + 1], (uint32_t)(lo >> 32));
            if (top) atomicOr(&s_out[q + 2], top);
        }
        // record table: a record starts where its '>' is; its bases start at the global index reached there
        uint32_t st = c.rs[p], k = 0;
        while (st) {
            const uint32_t i = (uint32_t)__builtin_ctz(st);
            st &= st - 1u;
            const unsigned long long r = r0 + k++;
            if (r < max_records) {
                rec_base[r] = o0 + (uint32_t)__popc(sm & ((1u << i) - 1u));
                if (rec_pos) rec_pos[r] = c0 + (uint64_t)p * kFqPiece + (uint64_t)threadIdx.x * kFqBytesPerThread + i;
            }
        }
    }
    __syncthreads();
    const uint32_t nd = chunk_seq ? (uint32_t)(((seq0 & 15ull) + chunk_seq + 15ull) >> 4) : 0u;  // dwords the chunk touches
    for (uint32_t i = threadIdx.x; i < nd; i += kFqThreads) {
        const uint32_t w = s_out[i];
        const uint64_t q = q0 + i;
        if (w == 0u || q >= out_dwords) continue;  // (the output was cleared)
        if (i == 0 || i + 1 == nd) atomicOr(&out32[q], w);
        else out32[q] = w;
    }
}

__global__ void fasta2_finish_kernel(const FaScratch sc, uint64_t groups, unsigned long long *rec_base, uint64_t max_records,
                                     unsigned long long *counts) {
    const unsigned long long bases = sc.g_base[groups], recs = sc.g_rec[groups];
    counts[0] = bases;
    counts[1] = recs;
    if (recs <= max_records) rec_base[recs] = bases;
}

uint64_t fa2_chunks(uint64_t n_bytes) { return (n_bytes + kFqChunk - 1) / kFqChunk; }
uint64_t fa2_groups(uint64_t chunks) { return (chunks + kFaGroup - 1) / kFaGroup; }

}  // namespace

uint64_t fasta2_scratch_bytes(uint64_t n_bytes) {
    const uint64_t chunks = fa2_chunks(n_bytes), groups = fa2_groups(chunks);
    return (3 * chunks + 5 * (groups + 1)) * sizeof(unsigned long long);
}

int launch_fasta_pack2(const uint8_t *d_text, uint64_t n_bytes, uint8_t *d_packed, uint64_t packed_capacity_bytes,
                       unsigned long long *d_rec_base, unsigned long long *d_rec_pos, uint64_t max_records,
                       unsigned long long *d_counts, void *scratch, hipStream_t stream) {
    const uint64_t chunks = fa2_chunks(n_bytes), groups = fa2_groups(chunks);
    if (chunks == 0 || chunks >= (1ull << 31)) return -1;
    unsigned long long *q = static_cast<unsigned long long *>(scratch);
    FaScratch sc;
    sc.fn = q, q += chunks;
    sc.pre0 = q, q += chunks;
    sc.pre1 = q, q += chunks;
    sc.grp0 = q, q += groups + 1;
    sc.grp1 = q, q += groups + 1;
    sc.g_base = q, q += groups + 1;
    sc.g_rec = q, q += groups + 1;
    sc.g_state = reinterpret_cast<uint32_t *>(q);
    const uint64_t out_dwords = packed_capacity_bytes / 4;
    // the packed bytes are OR-ed together where chunks meet: clear what the text can fill at most
    const uint64_t clear = packed_capacity_bytes < (n_bytes + 3) / 4 + 8 ? packed_capacity_bytes : (n_bytes + 3) / 4 + 8;
    if (clear && hipMemsetAsync(d_packed, 0, clear, stream) != hipSuccess) return -1;
    hipLaunchKernelGGL(fasta2_count_kernel, dim3((uint32_t)chunks), dim3(kFqThreads), 0, stream, d_text, n_bytes, sc.fn);
    hipLaunchKernelGGL(fasta2_groups_kernel, dim3((uint32_t)groups), dim3(kFaGroup), 0, stream, sc, chunks);
    hipLaunchKernelGGL(fasta2_resolve_kernel, dim3(1), dim3(kFaGroup), 0, stream, sc, groups);
    hipLaunchKernelGGL(fasta2_pack_kernel, dim3((uint32_t)chunks), dim3(kFqThreads), 0, stream, d_text, n_bytes, sc,
                       reinterpret_cast<uint32_t *>(d_packed), out_dwords, d_rec_base, d_rec_pos, max_records);
    hipLaunchKernelGGL(fasta2_finish_kernel, dim3(1), dim3(1), 0, stream, sc, groups, d_rec_base, max_records, d_counts);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace mm
