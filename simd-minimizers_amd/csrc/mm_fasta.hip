// mm_fasta.hip — FASTA text -> PackedSeq records on the device, the loader step in front of the hot path
// (the reference's bench harness does it on the CPU: needletail::parse_fastx_file + PackedSeqVec::from_ascii per
// record, bench/src/lib.rs:51-82).  Semantics restated here (needletail is not in the tree: parity unpinned, see
// DESIGN.md): a record starts with '>' at the start of a line; its header runs to the end of that line; its
// sequence is every following line up to the next header line with '\n' and '\r' removed; bytes before the first
// header are ignored; every sequence byte is packed as (c >> 1) & 3 like PackedSeqVec::from_ascii.
//
// All records are packed back to back into ONE 2-bit buffer (record r = bases [rec_base[r], rec_base[r+1]) of it,
// any base offset - what mm_run_batch_device takes), so the job is a stream compaction of the text.
//
// SINCE LATE ROUND 4 THE DEFAULT IS mm_fasta2.hip (two passes of mask arithmetic, no look-back, no limits); this file
// holds the packers of rounds 2-4, kept as its cross-checks (MM_FASTA_KERNEL=lines / three, MM_FASTA_ONEPASS=1 / 0).
// The default of rounds 3-4 is the ONE-PASS kernel fasta_lines_kernel further down ("the one-pass kernel over lines"):
// the text is read once, every 32 KB chunk is staged in LDS, cut into line segments with a SWAR separator scan and
// packed to its final place, and the chunks are chained by ONE decoupled look-back whose status word carries the line
// / header context, the bases and the records before the chunk (a look-back over functions, not only over sums).
// 1 GiB of 60-base lines: 0.87 ms (1.24 TB/s of text).  What follows first is the THREE-PASS family it replaced,
// kept as the fallback for texts the one-pass kernel gives up on (more than 2 048 line segments in a chunk, a
// look-back time-out: mm_fasta_pack_device repeats the call with it by itself) and as its cross-check
// (tests/test_gpu_fasta.py, tests/test_gpu_round4.py):
//   K1  per 32 KB chunk: position of its last '\n' and of its last record start (context-free: a '>' right after
//       a '\n' starts a record whatever came before)
//   S1  exclusive max-scan of both over the chunks (one workgroup) -> the line / record context at every chunk start
//   K2  per chunk, now with its context: number of sequence bytes and of record starts
//   S2  exclusive sum-scan of both
//   K3  per chunk: 2-bit codes of its sequence bytes to their final place (staged in LDS, whole dwords stored,
//       the two dwords a chunk shares with its neighbours OR-ed in), record table entries
// Three passes over the text (1.66 ms for the same 1 GiB).
#include "mm_common.h"
#include "mm_launch.h"
#include "mm_env.h"
#include <stdlib.h>

namespace mm {
namespace {

constexpr uint32_t kIterBytes = 16u * kBlockThreads;  // 4096: one 16-byte piece per thread
constexpr uint32_t kIters = 8;                        // pieces a workgroup walks one after the other (all loaded up front)
constexpr uint32_t kChunkBytes = kIterBytes * kIters;
constexpr int kScanThreads = 1024;

// 16 text bytes of one thread as bit masks (bit j = byte j): newline, carriage return, '>', inside the text
struct Piece {
    uint32_t nl, cr, gt, valid;
    uint32_t w[4];   // the bytes
    uint32_t prev_nl;  // the byte before the piece is '\n' (or the piece starts the text)
};

// bit j of the result = byte j of x equals c (bytes as 8-bit lanes of a dword)
__device__ __forceinline__ uint32_t eq_mask4(uint32_t x, uint32_t c) {
    const uint32_t t = x ^ (c * 0x01010101u);
    const uint32_t z = ~(((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t | 0x7f7f7f7fu);  // 0x80 where the byte is zero
    return ((z >> 7) | (z >> 14) | (z >> 21) | (z >> 28)) & 0xfu;
}

// the 16 bytes of a thread, loaded ahead of their use (a workgroup keeps all its iterations in flight)
struct Raw {
    uint32_t w[4];
    uint32_t valid;
};

__device__ __forceinline__ Raw load_raw(const uint8_t *__restrict__ text, uint64_t n, uint64_t o) {
    Raw r;
    r.w[0] = r.w[1] = r.w[2] = r.w[3] = 0;
    r.valid = 0;
    if (o >= n) return r;
    const uint64_t left = n - o;
    r.valid = left >= 16 ? 0xffffu : ((1u << (uint32_t)left) - 1u);
    if (left >= 16 && ((reinterpret_cast<uintptr_t>(text) + o) & 15u) == 0) {
        const uint4 v = *reinterpret_cast<const uint4 *>(text + o);
        r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z; r.w[3] = v.w;
    } else {
        const uint32_t m = left >= 16 ? 16u : (uint32_t)left;
        for (uint32_t j = 0; j < m; ++j) r.w[j >> 2] |= (uint32_t)text[o + j] << (8u * (j & 3u));
    }
    return r;
}

// masks of a piece; the byte in front of it comes from the neighbouring lane (the first lane of a wave reads it)
__device__ __forceinline__ Piece make_piece(const Raw &r, const uint8_t *__restrict__ text, uint64_t n, uint64_t o) {
    Piece p;
    p.w[0] = r.w[0]; p.w[1] = r.w[1]; p.w[2] = r.w[2]; p.w[3] = r.w[3];
    p.valid = r.valid;
    uint32_t prev = (uint32_t)__shfl_up((int)(r.w[3] >> 24), 1, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) prev = (o > 0 && o <= n) ? text[o - 1] : (uint32_t)'\n';
    p.prev_nl = (o < n && prev == (uint32_t)'\n') ? 1u : 0u;
    p.nl = p.cr = p.gt = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        p.nl |= eq_mask4(p.w[g], '\n') << (4 * g);
        p.cr |= eq_mask4(p.w[g], '\r') << (4 * g);
        p.gt |= eq_mask4(p.w[g], '>') << (4 * g);
    }
    p.nl &= p.valid;
    p.cr &= p.valid;
    p.gt &= p.valid;
    return p;
}

// line starts and record starts inside a piece
__device__ __forceinline__ void starts(const Piece &p, uint32_t &ls, uint32_t &rs) {
    ls = ((p.nl << 1) | p.prev_nl) & 0xffffu;
    rs = p.gt & ls;
}

// Header bytes of a piece: every line start carries its own answer (it is a record start or it is not), every
// other byte inherits the answer of the byte before it, the first byte inherits h0.  One addition does the
// segmented fill: put a carry on every header line start and let it run through the bytes that are not line
// starts (bit 0 is the virtual byte before the piece).
__device__ __forceinline__ uint32_t header_mask(uint32_t rs, uint32_t ls, uint32_t h0) {
    const uint32_t a = (rs << 1) | h0, u = (ls << 1) | 1u, m = 0x1ffffu;
    const uint32_t z = (~u | a) & m;
    const uint32_t r = a + z;
    return ((((r ^ z) | a) & z) >> 1) & 0xffffu;
}

// sequence bytes of a piece, given the context in front of it: ln / lr = position + 1 of the last '\n' / record
// start before the piece (0 = none; ln is the start of the line the piece begins in)
__device__ __forceinline__ uint32_t base_mask(const Piece &p, uint32_t ls, uint32_t rs, uint64_t ln, uint64_t lr) {
    const uint32_t hdr = header_mask(rs, ls, lr > ln ? 1u : 0u);
    // no sequence before the first record
    const uint32_t started = lr > 0 ? 0xffffu : (rs ? ~((rs & (0u - rs)) - 1u) & 0xffffu : 0u);
    return ~hdr & ~p.nl & ~p.cr & p.valid & started & 0xffffu;
}

__device__ __forceinline__ uint32_t top_bit_pos1(uint32_t m) { return m ? 32u - (uint32_t)__builtin_clz(m) : 0u; }

// Marks (newline / record start) are positions, so the latest mark before a thread is the largest one: the
// context of every thread of the workgroup comes from one ballot and one shuffle per wave (the nearest earlier
// lane that holds a mark) and a four-entry exchange between the waves.  a / b = the thread's own last newline /
// record start as position + 1 relative to the iteration (0 = none); returns the exclusive maxima in xa / xb and
// the workgroup's maxima in ta / tb.  s = 2 * kWavesPerBlock words of LDS.
template <int WAVES = kWavesPerBlock>
__device__ __forceinline__ void block_prev_marks(uint32_t a, uint32_t b, uint32_t *s, uint32_t &xa, uint32_t &xb,
                                                 uint32_t &ta, uint32_t &tb) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    auto wave_part = [&](uint32_t v, uint32_t &excl, uint32_t &last) {
        const unsigned long long m = __ballot(v != 0u);
        const unsigned long long lower = m & ((1ull << lane) - 1ull);
        const uint32_t from_lower = (uint32_t)__shfl((int)v, lower ? 63 - __builtin_clzll(lower) : 0, kWave);
        const uint32_t from_top = (uint32_t)__shfl((int)v, m ? 63 - __builtin_clzll(m) : 0, kWave);
        excl = lower ? from_lower : 0u;
        last = m ? from_top : 0u;
    };
    uint32_t la, lb;
    wave_part(a, xa, la);
    wave_part(b, xb, lb);
    if (lane == 0) {
        s[wave] = la;
        s[WAVES + wave] = lb;
    }
    __syncthreads();
    ta = tb = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
        const uint32_t va = s[w], vb = s[WAVES + w];
        if (w < wave) {
            xa = va > xa ? va : xa;
            xb = vb > xb ? vb : xb;
        }
        ta = va > ta ? va : ta;
        tb = vb > tb ? vb : tb;
    }
    __syncthreads();
}

// exclusive sum over the workgroup of one small count per thread (two 16-bit counts in one word)
__device__ __forceinline__ uint32_t block_sum_excl(uint32_t v, uint32_t *s, uint32_t &total) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d, kWave);
        if (lane >= d) incl += o;
    }
    if (lane == kWave - 1) s[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) {
        const uint32_t t = s[w];
        if (w < wave) base += t;
        tot += t;
    }
    __syncthreads();
    total = tot;
    return base + incl - v;
}

// inclusive prefix sum over the lanes of a wave
__device__ __forceinline__ uint32_t wave_scan_sum(uint32_t v) {
    const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)v, d, kWave);
        if (lane >= d) v += o;
    }
    return v;
}

struct MaxOp {
    __device__ unsigned long long operator()(unsigned long long a, unsigned long long b) const { return a > b ? a : b; }
};
struct AddOp {
    __device__ unsigned long long operator()(unsigned long long a, unsigned long long b) const { return a + b; }
};

// K1: last newline and last record start of every chunk (position + 1, 0 = none)
__global__ __launch_bounds__(kBlockThreads) void fasta_marks_kernel(const uint8_t *__restrict__ text, uint64_t n,
                                                                    unsigned long long *__restrict__ last_nl,
                                                                    unsigned long long *__restrict__ last_rec) {
    __shared__ unsigned long long s[2 * kWavesPerBlock];
    const uint64_t c0 = (uint64_t)blockIdx.x * kChunkBytes;
    unsigned long long mnl = 0, mrec = 0;
    Raw raw[kIters];
#pragma unroll
    for (uint32_t it = 0; it < kIters; ++it) raw[it] = load_raw(text, n, c0 + (uint64_t)it * kIterBytes + 16ull * threadIdx.x);
#pragma unroll
    for (uint32_t it = 0; it < kIters; ++it) {
        const uint64_t o = c0 + (uint64_t)it * kIterBytes + 16ull * threadIdx.x;
        const Piece p = make_piece(raw[it], text, n, o);
        uint32_t ls, rs;
        starts(p, ls, rs);
        if (p.nl) mnl = o + top_bit_pos1(p.nl);
        if (rs) mrec = o + top_bit_pos1(rs);
    }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long a = __shfl_xor(mnl, d, kWave), b = __shfl_xor(mrec, d, kWave);
        mnl = a > mnl ? a : mnl;
        mrec = b > mrec ? b : mrec;
    }
    if (lane == 0) {
        s[wave] = mnl;
        s[kWavesPerBlock + wave] = mrec;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long a = 0, b = 0;
        for (int w = 0; w < kWavesPerBlock; ++w) {
            a = s[w] > a ? s[w] : a;
            b = s[kWavesPerBlock + w] > b ? s[kWavesPerBlock + w] : b;
        }
        last_nl[blockIdx.x] = a;
        last_rec[blockIdx.x] = b;
    }
}

// S1 / S2: exclusive scans of TWO arrays of `n` values by ONE workgroup (n <= 2^17 chunks for a 4 GB text);
// out[n] = total.  Every wave owns a contiguous slab: it reduces it (coalesced groups of 64, loads independent),
// the sixteen slab totals are combined through LDS, then the wave scans its slab group by group.
template <class Op>
__global__ __launch_bounds__(kScanThreads) void fasta_scan2_kernel(const unsigned long long *__restrict__ in_a,
                                                                   const unsigned long long *__restrict__ in_b,
                                                                   uint64_t n, unsigned long long identity,
                                                                   unsigned long long *__restrict__ out_a,
                                                                   unsigned long long *__restrict__ out_b) {
    constexpr int kWaves = kScanThreads / kWave;
    __shared__ unsigned long long s[2][kWaves];
    Op op;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const uint64_t groups = (n + kWave - 1) / kWave, per_wave = (groups + kWaves - 1) / kWaves;
    const uint64_t g0 = (uint64_t)wave * per_wave, g1 = g0 + per_wave < groups ? g0 + per_wave : groups;
    unsigned long long ra = identity, rb = identity;
    constexpr int kAhead = 8;  // groups loaded together: one wave per SIMD has nothing else to hide the latency
    for (uint64_t g = g0; g < g1; g += kAhead) {
        unsigned long long va[kAhead], vb[kAhead];
#pragma unroll
        for (int e = 0; e < kAhead; ++e) {
            const uint64_t i = (g + e) * kWave + lane;
            const bool in = g + e < g1 && i < n;
            va[e] = in ? in_a[i] : identity;
            vb[e] = in ? in_b[i] : identity;
        }
#pragma unroll
        for (int e = 0; e < kAhead; ++e) {
            ra = op(ra, va[e]);
            rb = op(rb, vb[e]);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        ra = op(ra, __shfl_xor(ra, d, kWave));
        rb = op(rb, __shfl_xor(rb, d, kWave));
    }
    if (lane == 0) {
        s[0][wave] = ra;
        s[1][wave] = rb;
    }
    __syncthreads();
    unsigned long long ca = identity, cb = identity, ta = identity, tb = identity;
    for (int w = 0; w < kWaves; ++w) {
        if (w < wave) {
            ca = op(ca, s[0][w]);
            cb = op(cb, s[1][w]);
        }
        ta = op(ta, s[0][w]);
        tb = op(tb, s[1][w]);
    }
    for (uint64_t gg = g0; gg < g1; gg += kAhead) {
        unsigned long long wa[kAhead], wb[kAhead];
#pragma unroll
        for (int e = 0; e < kAhead; ++e) {
            const uint64_t i = (gg + e) * kWave + lane;
            const bool in = gg + e < g1 && i < n;
            wa[e] = in ? in_a[i] : identity;
            wb[e] = in ? in_b[i] : identity;
        }
#pragma unroll
        for (int e = 0; e < kAhead; ++e) {
            const uint64_t i = (gg + e) * kWave + lane;
            unsigned long long ia = wa[e], ib = wb[e];
#pragma unroll
            for (int d = 1; d < kWave; d <<= 1) {
                const unsigned long long oa = __shfl_up(ia, d, kWave), ob = __shfl_up(ib, d, kWave);
                if (lane >= d) {
                    ia = op(oa, ia);
                    ib = op(ob, ib);
                }
            }
            unsigned long long ea = __shfl_up(ia, 1, kWave), eb = __shfl_up(ib, 1, kWave);
            if (lane == 0) ea = eb = identity;
            if (gg + e < g1 && i < n) {
                out_a[i] = op(ca, ea);
                out_b[i] = op(cb, eb);
            }
            ca = op(ca, __shfl(ia, kWave - 1, kWave));
            cb = op(cb, __shfl(ib, kWave - 1, kWave));
        }
    }
    if (threadIdx.x == 0) {
        out_a[n] = ta;
        out_b[n] = tb;
    }
}

// K2 (PACK = false): sequence bytes and record starts per chunk.
// K3 (PACK = true): the same walk with the chunk's output offsets: packs and fills the record table.
template <bool PACK>
__global__ __launch_bounds__(kBlockThreads) void fasta_walk_kernel(
    const uint8_t *__restrict__ text, uint64_t n, const unsigned long long *__restrict__ ctx_nl,
    const unsigned long long *__restrict__ ctx_rec, unsigned long long *__restrict__ cnt_bases,
    unsigned long long *__restrict__ cnt_recs, const unsigned long long *__restrict__ off_bases,
    const unsigned long long *__restrict__ off_recs, uint32_t *__restrict__ out32, uint64_t out_dwords,
    unsigned long long *__restrict__ rec_base, unsigned long long *__restrict__ rec_pos, uint64_t max_records,
    unsigned long long *__restrict__ counts, uint32_t n_chunks) {
    __shared__ uint32_t s[2 * kWavesPerBlock];
    __shared__ uint32_t s_stage[2][kIterBytes / 16 + 2];
    const uint32_t tid = threadIdx.x;
    const uint64_t c0 = (uint64_t)blockIdx.x * kChunkBytes;
    unsigned long long ln_run = ctx_nl[blockIdx.x], lr_run = ctx_rec[blockIdx.x];
    unsigned long long bases_run = PACK ? off_bases[blockIdx.x] : 0ull, recs_run = PACK ? off_recs[blockIdx.x] : 0ull;
    if (PACK) {
        for (uint32_t i = tid; i < kIterBytes / 16 + 2; i += kBlockThreads) s_stage[0][i] = s_stage[1][i] = 0;
        // (ordered before the first use by the barriers inside block_prev_marks)
    }
    uint32_t my_bases = 0, my_recs = 0;
    Raw raw[kIters];
#pragma unroll
    for (uint32_t it = 0; it < kIters; ++it) raw[it] = load_raw(text, n, c0 + (uint64_t)it * kIterBytes + 16ull * tid);
#pragma unroll
    for (uint32_t it = 0; it < kIters; ++it) {
        const uint64_t i0 = c0 + (uint64_t)it * kIterBytes;  // first byte of the iteration
        if (i0 >= n) break;                                   // (uniform)
        const uint64_t o = i0 + 16ull * tid;
        const Piece p = make_piece(raw[it], text, n, o);
        uint32_t ls, rs;
        starts(p, ls, rs);
        // marks as position + 1 relative to the iteration
        const uint32_t tnl = p.nl ? 16u * tid + top_bit_pos1(p.nl) : 0u;
        const uint32_t trec = rs ? 16u * tid + top_bit_pos1(rs) : 0u;
        uint32_t xnl, xrec, tot_nl, tot_rec;
        block_prev_marks(tnl, trec, s, xnl, xrec, tot_nl, tot_rec);
        const unsigned long long ln = xnl ? i0 + xnl : ln_run, lr = xrec ? i0 + xrec : lr_run;
        const uint32_t bm = base_mask(p, ls, rs, ln, lr);
        const uint32_t nb = (uint32_t)__builtin_popcount(bm), nr = (uint32_t)__builtin_popcount(rs);
        if (!PACK) {  // only the chunk's totals are wanted: summed once, after the last iteration
            my_bases += nb;
            my_recs += nr;
            ln_run = tot_nl ? i0 + tot_nl : ln_run;
            lr_run = tot_rec ? i0 + tot_rec : lr_run;
            continue;
        }
        uint32_t tot;
        const uint32_t ex = block_sum_excl((nr << 16) | nb, s, tot);
        const uint32_t xb = ex & 0xffffu, xr = ex >> 16, tot_b = tot & 0xffffu, tot_r = tot >> 16;
        if (PACK) {
            const unsigned long long g0 = bases_run + xb;       // global index of this thread's first base
            // record table: a record starts where its '>' is; its bases start at the global index reached there
            for (uint32_t m = rs, k = 0; m; m &= m - 1u, ++k) {
                const uint32_t j = (uint32_t)__builtin_ctz(m);
                const unsigned long long r = recs_run + xr + k;
                if (r < max_records) {
                    rec_base[r] = g0 + (uint32_t)__builtin_popcount(bm & ((1u << j) - 1u));
                    if (rec_pos) rec_pos[r] = o + j;
                }
            }
            // 2-bit codes of the thread's bases, in order: the codes of all 16 bytes (SWAR, as pack_ascii does),
            // then the bytes that are not bases are squeezed out from the top down - a sequence line has one or
            // two of them per piece (its line end), so the loop runs once or twice for most waves
            uint32_t v = 0;
            if (bm) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint32_t t = (p.w[g] >> 1) & 0x03030303u;
                    v |= ((t | (t >> 6) | (t >> 12) | (t >> 18)) & 0xffu) << (8 * g);
                }
                for (uint32_t holes = ~bm & 0xffffu; holes;) {
                    const uint32_t j = 31u - (uint32_t)__builtin_clz(holes);
                    holes &= ~(1u << j);
                    const uint32_t low = (1u << (2u * j)) - 1u;
                    v = (v & low) | ((v >> 2) & ~low);
                }
            }
            // stage the iteration's bases in LDS at their place relative to its first output dword (two staging
            // buffers take turns: the one not in use was cleared while the other was flushed)
            uint32_t *stage = s_stage[it & 1u];
            const unsigned long long G0 = bases_run, G0a = G0 & ~15ull;
            const uint32_t n_stage = (uint32_t)((G0 - G0a + tot_b + 15ull) / 16ull);
            if (nb) {
                const uint32_t rel = (uint32_t)(g0 - G0a), d = rel >> 4, bsh = 2u * (rel & 15u);
                atomicOr(&stage[d], v << bsh);
                if (bsh && bsh + 2u * nb > 32u) atomicOr(&stage[d + 1], v >> (32u - bsh));
            }
            __syncthreads();
            for (uint32_t i = tid; i < n_stage; i += kBlockThreads) {
                const unsigned long long dw = G0a / 16ull + i;
                const uint32_t val = stage[i];
                stage[i] = 0;
                if (dw < out_dwords) {
                    // the first and the last dword may be shared with the neighbouring iteration / chunk
                    if (i == 0 || i + 1 == n_stage) {
                        if (val) atomicOr(&out32[dw], val);
                    } else {
                        out32[dw] = val;
                    }
                }
            }
            // (the next iteration stages into the other buffer; this one is cleared again and will be reused
            // only after two more barriers)
        }
        ln_run = tot_nl ? i0 + tot_nl : ln_run;
        lr_run = tot_rec ? i0 + tot_rec : lr_run;
        bases_run += tot_b;
        recs_run += tot_r;
    }
    if (!PACK) {
        uint32_t tot;
        block_sum_excl((my_recs << 16) | my_bases, s, tot);  // (a chunk holds at most 32768 bases: they fit the low half)
        if (tid == 0) {
            cnt_bases[blockIdx.x] = tot & 0xffffu;
            cnt_recs[blockIdx.x] = tot >> 16;
        }
    } else if (tid == 0 && blockIdx.x == n_chunks - 1) {
        counts[0] = bases_run;
        counts[1] = recs_run;
        if (recs_run <= max_records) rec_base[recs_run] = bases_run;
    }
}

// ---------------------------------------------------------------------------------------------- one pass (round 3)
// The same compaction with the text read ONCE: a chunk keeps its 32 KB in registers and resolves what it needs
// from its predecessors through decoupled look-backs (status words as in mm_common.h, one word per chunk and
// quantity, relaxed agent-scope accesses):
//   A  the line / record context at the chunk's start, which is only two bits - "inside a header line" and "a
//      record has started" - and context-free to publish: a chunk that holds a newline or a record start decides
//      the first bit for its successors by itself (kind 1 / 2), one that holds neither is transparent (kind 0);
//   B, R  the number of sequence bytes and of record starts before the chunk (sums), published once the context
//      is known; wave 0 looks back over B while wave 1 looks back over R.
// Chunk ids are blockIdx.x (in-order dispatch, as the fused kernel assumes); every spin is bounded and a time-out
// raises the workspace's error word, upon which the host repeats the call with the three-pass kernels above.
constexpr unsigned long long kCtxHeader = 2ull, kCtxPlain = 1ull, kCtxRec = 4ull;  // payload bits of an A word

// exclusive context of chunk `bid`: bit 0 = inside a header line, bit 1 = a record has started
__device__ __forceinline__ uint32_t lookback_context(unsigned long long *status, uint32_t bid, uint32_t kind, uint32_t has_rec,
                                                     uint32_t *error) {
    const int lane = threadIdx.x & (kWave - 1);
    uint32_t h = 0, started = 0;  // before the text: no header line, no record
    bool have_h = false;
    if (bid != 0) {
        if (lane == 0) st_status(&status[bid], kFlagAgg | (kind & 3u) | (has_rec ? kCtxRec : 0ull));
        long long j = (long long)bid - 1;
        while (true) {
            const long long idx = j - lane;
            unsigned long long s = idx >= 0 ? ld_status(&status[idx]) : (kFlagIncl | kCtxPlain);
            for (uint32_t spins = 0; (s >> 62) == 0; ++spins) {
                if (spins > kMaxLookbackSpins) {
                    flag_error(error, 1u);
                    s = kFlagIncl | kCtxPlain;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
                s = ld_status(&status[idx]);
            }
            const unsigned long long incl_mask = __ballot((s >> 62) == 2);
            const int first = incl_mask ? __builtin_ctzll(incl_mask) : kWave;  // lanes 0 .. first count (nearest first)
            const bool mine = lane <= first;
            const unsigned long long kmask = __ballot(mine && (s & 3ull) != 0ull);
            const unsigned long long rmask = __ballot(mine && (s & kCtxRec) != 0ull);
            if (!have_h && kmask) {
                const int src = __builtin_ctzll(kmask);
                h = ((uint32_t)__shfl((int)(uint32_t)(s & 3ull), src, kWave) == (uint32_t)kCtxHeader) ? 1u : 0u;
                have_h = true;
            }
            if (rmask) started = 1;
            if (incl_mask) break;
            j -= kWave;
        }
    }
    // inclusive state of this chunk (an inclusive word always carries a decided kind)
    const uint32_t h_out = kind ? (kind == (uint32_t)kCtxHeader ? 1u : 0u) : h;
    if (lane == 0)
        st_status(&status[bid], kFlagIncl | (h_out ? kCtxHeader : kCtxPlain) | ((started | has_rec) ? kCtxRec : 0ull));
    return h | (started << 1);
}

// One-pass kernel over LINES (round 3, third version; the first two kept the three-pass kernels' per-byte masks behind
// the look-backs and ran no faster than the three passes: 2.07 and 1.77 ms for 1 GiB against 1.68 ms).
//
// FASTA text is long runs of sequence bytes between a few separators, so the work is split by what it is proportional
// to.  A chunk is kLnChunk = 32 KB of text (eight waves; MM_FASTA_LN_WAVES=4: 16 KB), staged in LDS once:
//   A1 (per text byte, SWAR, ~0.13 VALU cycles / byte)  candidate separators = bytes below 0x0E ('\n', '\r' and the odd
//       control character), four adds and logic operations per dword; their positions go to a sorted list in LDS
//       (a wave owns 4 KB: 64 lanes x 16 bytes x 4 rows; one packed scan of the per-lane counts orders them);
//   A2 (per SEGMENT = the bytes between two candidates, one thread each)  where it starts and ends, whether it starts
//       a line ('\n' before it), whether it starts a record ('>' at a line start); the chunk's kind goes to the
//       context look-back; header state by "latest line start == latest record start"; lengths, output offsets and
//       the compact table of segments that hold sequence through one packed workgroup scan; the totals to the two
//       sum look-backs;
//   B  (per OUTPUT dword, ~0.2 VALU cycles / byte)  lane q packs output dword q of the chunk: the segment that holds
//       its first base comes from a mark every segment leaves at the first dword boundary it covers (nearest mark
//       below: one ballot), then 16 text bytes from LDS at any byte offset, 2-bit codes by one multiply per dword,
//       and on to the next segment while bases are missing (a line end costs a second round, nothing else).  Stores
//       are whole dwords in order; only the two dwords a chunk shares with its neighbours are OR-ed in.
// Bytes that are neither '\n' nor '\r' but below 0x0E stay sequence bytes (they end a segment of width zero).  A chunk
// with more than kLnMaxSeg - 1 candidates (lines shorter than 16 bytes on average) or more than kLnMaxRec record starts
// raises error 3 and the host takes
// the three-pass kernels for that text.
#ifndef MM_FASTA_LN_WAVES
#define MM_FASTA_LN_WAVES 8  // (4: 16 KB chunks, six workgroups per CU; 8: 32 KB chunks, three - half as many links in the look-back chain)
#endif
constexpr int kLnWaves = MM_FASTA_LN_WAVES;                        // waves of a workgroup
constexpr int kLnThreads = kLnWaves * kWave;
constexpr uint32_t kLnWaveBytes = 4096;                            // 4 KB of text per wave
constexpr uint32_t kLnChunk = kLnWaveBytes * kLnWaves;             // text bytes per chunk = workgroup
constexpr uint32_t kLnRows = kLnWaveBytes / (16u * kWave);         // 4 rows of 1 KB per wave
#ifndef MM_FASTA_MAXSEG
#define MM_FASTA_MAXSEG (256 * MM_FASTA_LN_WAVES)
#endif
constexpr uint32_t kLnMaxSeg = MM_FASTA_MAXSEG;                               // segments of a chunk (candidates + 1)
constexpr uint32_t kLnMaxRec = 128u * kLnWaves;                               // record starts of a chunk
constexpr uint32_t kLnPad = 16;                                    // bytes in front of the chunk's text in LDS
constexpr uint32_t kLnMaxQ = kLnChunk / 16u + 2u;                  // output dwords a chunk can touch
static_assert(kLnRows == 4, "the packed scans below hold four rows");

struct LnShared {
    uint32_t text[(kLnPad + kLnChunk + 32u) / 4u];
    uint32_t tab[kLnMaxSeg + 1];     // segments that hold sequence: output offset in the chunk | text start << 16
    uint32_t recs[kLnMaxRec];        // record starts: output offset | text position << 16
    uint16_t list[kLnMaxSeg];        // candidate positions, ascending
    uint16_t marks[kLnMaxQ];         // per output dword: 1 + the table entry that holds its first base (0 = none)
    uint32_t s[2 * kLnWaves];
    unsigned long long s64[kLnWaves];
    uint32_t cnt[kLnWaves];
    uint32_t ctx;
    uint32_t u_end, v_end;           // sequence bytes | table entries << 16 before the first line start / record start
    uint32_t edge[kLnWaves];   // the last output dword every wave holds (for its neighbour's funnel shift)
    unsigned long long off[2];
};

// ---- the look-back of the one-pass kernel: ONE status word per chunk.
// What a chunk adds to the running sums depends on the state it is entered in - inside a header line or not, a record
// started or not - and that state comes from the chunks before it.  So that a chunk can publish BEFORE it knows its
// state (the point of a decoupled look-back), its aggregate is the function itself: its sequence bytes in three
// classes - U before its first line start (count if a record has started and the text is not inside a header), V
// between the first line start and the first record start (count if a record has started), K behind the first record
// start (always count) - its record starts, and what it does to the state (kind: 0 nothing, 1 ends outside a header
// line, 2 ends inside one; whether it holds a record start).  The chunk that looks back evaluates the functions of
// its predecessors from the nearest inclusive word forwards; the states along the way come from three ballots.
//   aggregate: K | V | U (kLnFB = 15 or 16 bits each) | records (11 bits) | kind (2) | has record (1) | flag 62..63 (= 1)
//   inclusive: sequence bytes 0..31 | records 32..59 | inside header 60 | record started 61 | flag (= 2)
// (texts are below 4 GB, a chunk holds fewer than 1024 segments: 28 bits hold the records.)
struct LnPrefix {
    uint32_t h, started;
    unsigned long long bases, recs;
};
constexpr int kLnFB = kLnChunk <= 16384u ? 15 : 16;  // bits of a class count in the aggregate word
static_assert(kLnChunk <= 32768u && kLnMaxSeg <= 2048u, "the aggregate word holds 16-bit class counts and 11-bit record counts");
constexpr int kLnGroups = 16;  // 64-chunk groups kept in registers while looking for the nearest inclusive word
#ifndef MM_FASTA_LB_BATCH
#define MM_FASTA_LB_BATCH 4
#endif
constexpr int kLnBatch = MM_FASTA_LB_BATCH;  // groups loaded together

// one thread, as soon as the chunk's classes are counted (chunk 0 publishes its inclusive word straight away)
__device__ __forceinline__ void ln_publish(unsigned long long *status, uint32_t bid, uint32_t K, uint32_t V, uint32_t U,
                                           uint32_t nr, uint32_t kind, uint32_t has_rec) {
    st_status(&status[bid], kFlagAgg | K | ((unsigned long long)V << kLnFB) | ((unsigned long long)U << (2 * kLnFB)) |
                                ((unsigned long long)nr << (3 * kLnFB)) | ((unsigned long long)kind << (3 * kLnFB + 11)) |
                                ((unsigned long long)has_rec << (3 * kLnFB + 13)));
}

__device__ __forceinline__ LnPrefix lookback_lines(unsigned long long *status, uint32_t bid, uint32_t K, uint32_t V, uint32_t U,
                                                   uint32_t nr, uint32_t kind, uint32_t has_rec, uint32_t *error,
                                                   const bool no_wait = false) {
    const int lane = threadIdx.x & (kWave - 1);
    LnPrefix r;
    r.h = r.started = 0;
    r.bases = r.recs = 0;
    if (bid != 0) {
        unsigned long long s[kLnGroups];
#pragma unroll
        for (int c = 0; c < kLnGroups; ++c) s[c] = kFlagIncl;
        int gF = -1, F = 0;
        for (uint32_t tries = 0; gF < 0; ++tries) {
            if (tries > 64u) {  // (every try spins up to kMaxLookbackSpins per word)
                flag_error(error, 1u);
                return r;
            }
            // kLnBatch groups' loads in flight together (one memory round trip for 64 * kLnBatch predecessors: the
            // frontier of finished chunks has to advance ~100 chunks per microsecond to keep up with the packing),
            // examined nearest first; a word that was still empty is polled when its group's turn comes
#pragma unroll
            for (int c0 = 0; c0 < kLnGroups; c0 += kLnBatch) {
                if (gF >= 0) continue;
#pragma unroll
                for (int c = c0; c < c0 + kLnBatch; ++c) {
                    const long long idx = (long long)bid - 1 - 64ll * c - lane;
                    s[c] = idx >= 0 ? ld_status(&status[idx]) : kFlagIncl;  // before the text: all zero
                }
#pragma unroll
                for (int c = c0; c < c0 + kLnBatch; ++c) {
                    if (gF >= 0) continue;
                    const long long idx = (long long)bid - 1 - 64ll * c - lane;
                    unsigned long long w = s[c];
                    if (no_wait && (w >> 62) != 2) w = kFlagIncl | (1ull << 61);  // (timing experiment: one round trip, no polling)
                    for (uint32_t spins = 0; (w >> 62) == 0; ++spins) {
                        if (spins > kMaxLookbackSpins) {
                            flag_error(error, 1u);
                            w = kFlagIncl;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                        w = ld_status(&status[idx]);
                    }
                    s[c] = w;
                    const unsigned long long incl_mask = __ballot((w >> 62) == 2);
                    if (incl_mask) {
                        gF = c;
                        F = __builtin_ctzll(incl_mask);
                    }
                }
            }
            if (gF < 0) __builtin_amdgcn_s_sleep(32);  // more than 1024 chunks ahead of every finished one: look again
        }
        // from the inclusive word forwards (far to near: group gF .. 0, within a group lane 63 .. 0)
        uint32_t h = 0, st = 0;
        unsigned long long sums = 0, mine = 0;  // sequence bytes | records << 32: of the inclusive word / this lane's share
#pragma unroll
        for (int c = kLnGroups - 1; c >= 0; --c) {
            if (c > gF) continue;
            const unsigned long long w = s[c];
            unsigned long long part = ~0ull;
            if (c == gF) {
                const unsigned long long wi = __shfl(w, F, kWave);
                h = (uint32_t)(wi >> 60) & 1u;
                st = (uint32_t)(wi >> 61) & 1u;
                sums = (wi & 0xffffffffull) | (((wi >> 32) & 0xfffffffull) << 32);
                part = F ? ((1ull << F) - 1ull) : 0ull;
            }
            const uint32_t wk = (uint32_t)(w >> (3 * kLnFB + 11)) & 3u;
            const unsigned long long D = __ballot(wk != 0u) & part, Hd = __ballot(wk == 2u) & part,
                                     Rc = __ballot(((w >> (3 * kLnFB + 13)) & 1ull) != 0ull) & part;
            const unsigned long long above = lane == 63 ? 0ull : ~((2ull << lane) - 1ull);
            const unsigned long long da = D & above;
            const uint32_t h_in = da ? (uint32_t)(Hd >> __builtin_ctzll(da)) & 1u : h;
            const uint32_t st_in = (st || (Rc & above)) ? 1u : 0u;
            if ((part >> lane) & 1ull) {
                constexpr uint32_t kFM = (1u << kLnFB) - 1u;
                const uint32_t k = (uint32_t)w & kFM, v = (uint32_t)(w >> kLnFB) & kFM, u = (uint32_t)(w >> (2 * kLnFB)) & kFM;
                mine += (unsigned long long)(k + (st_in ? v + (h_in ? 0u : u) : 0u)) | (((w >> (3 * kLnFB)) & 0x7ffull) << 32);
            }
            if (D) h = (uint32_t)(Hd >> __builtin_ctzll(D)) & 1u;
            if (Rc) st = 1u;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d, kWave);  // (one reduction for all groups)
        sums += mine;
        r.h = h;
        r.started = st;
        r.bases = sums & 0xffffffffull;
        r.recs = sums >> 32;
    }
    const uint32_t mine = K + (r.started ? V + (r.h ? 0u : U) : 0u);
    const uint32_t h_out = kind ? (kind == 2u ? 1u : 0u) : r.h, st_out = r.started | has_rec;
    const unsigned long long rb = r.bases + mine, rr = r.recs + nr;
    if (rb >= (1ull << 32) || rr >= (1ull << 28)) flag_error(error, 3u);  // (cannot happen below 4 GB of text)
    if (lane == 0)
        st_status(&status[bid], kFlagIncl | (rb & 0xffffffffull) | ((rr & 0xfffffffull) << 32) |
                                    ((unsigned long long)h_out << 60) | ((unsigned long long)st_out << 61));
    return r;
}

// exclusive sum over the workgroup of three packed counts (16-bit fields of a 64-bit word)
__device__ __forceinline__ unsigned long long block_sum_excl64(unsigned long long v, unsigned long long *s,
                                                               unsigned long long &total) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    unsigned long long incl = v;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const unsigned long long o = __shfl_up(incl, d, kWave);
        if (lane >= d) incl += o;
    }
    if (lane == kWave - 1) s[wave] = incl;
    __syncthreads();
    unsigned long long base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kLnWaves; ++w) {
        const unsigned long long t = s[w];
        if (w < wave) base += t;
        tot += t;
    }
    __syncthreads();
    total = tot;
    return base + incl - v;
}

// 16 text bytes (four dwords, first byte lowest) -> their sixteen 2-bit codes, first base lowest: (c >> 1) & 3 per
// byte, then one multiply gathers the four fields of a dword in its top byte (the partial products do not overlap)
__device__ __forceinline__ uint32_t pack16(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3) {
    const uint32_t m0 = ((x0 >> 1) & 0x03030303u) * 0x01041040u, m1 = ((x1 >> 1) & 0x03030303u) * 0x01041040u;
    const uint32_t m2 = ((x2 >> 1) & 0x03030303u) * 0x01041040u, m3 = ((x3 >> 1) & 0x03030303u) * 0x01041040u;
    return __builtin_amdgcn_perm(m1, m0, 0x0c0c0703u) | __builtin_amdgcn_perm(m3, m2, 0x07030c0cu);
}

// (Tried, round 3: the same look-back over TWO levels - super-chunks of 16 chunks whose last chunk composes their
// functions into one level-2 word, every chunk looking back over 64 super-chunks and its own super-chunk's chunks in
// one round trip - so that the front of finished chunks is not held to one 256-chunk window per status round trip.
// 1.52 ms against 0.89 ms: the level-2 words of the super-chunks in flight are published a packing phase and a round
// trip after their chunks' own words, and every chunk of the following super-chunks waits for them.  What bounds the
// one-level form is DESIGN.md 4.3a.)
#ifndef MM_FASTA_WAVES
#define MM_FASTA_WAVES 6  // 85 VGPRs and 26.8 KB of LDS: six workgroups per CU (the kernel is bound by latencies)
#endif
__global__ __launch_bounds__(kLnThreads, MM_FASTA_WAVES) void fasta_lines_kernel(
    const uint8_t *__restrict__ text, uint64_t n, unsigned long long *__restrict__ st_ctx,
    uint32_t *__restrict__ out32,
    uint64_t out_dwords, unsigned long long *__restrict__ rec_base, unsigned long long *__restrict__ rec_pos,
    uint64_t max_records, unsigned long long *__restrict__ counts, uint32_t n_chunks, uint32_t *error, uint32_t debug) {
    // debug (MM_FASTA_DEBUG, timing experiments with wrong results): 1 no look-backs, 2 stop before B, 4 stop after A1,
    // 8 stop before A2's scans, 16 the look-back takes whatever its first loads return
    __shared__ __attribute__((aligned(16))) LnShared sh;
    const uint32_t tid = threadIdx.x, bid = blockIdx.x;
    const int lane = tid & (kWave - 1), wave = tid / kWave;
    const uint64_t c0 = (uint64_t)bid * kLnChunk;
    const uint32_t L = n - c0 < kLnChunk ? (uint32_t)(n - c0) : kLnChunk;  // text bytes of this chunk
    uint8_t *const tx = reinterpret_cast<uint8_t *>(sh.text) + kLnPad;     // tx[p] = byte p of the chunk
    const uint32_t w0 = (uint32_t)wave * kLnWaveBytes;
    // A chunk that cannot go on (its text does not fit the tables) raises error 3 - the host repeats the text with the
    // three-pass kernels - and still plays its part in the look-back, with nothing to add: its successors must not wait.
    auto give_up = [&]() {
        if (tid == 0) flag_error(error, 3u);
        if (tid == 0) st_status(&st_ctx[bid], kFlagIncl | (1ull << 61));
    };

    // ---- A1: the text into LDS, candidate separators into the list
    Raw raw[kLnRows];
#pragma unroll
    for (uint32_t r = 0; r < kLnRows; ++r) raw[r] = load_raw(text, n, c0 + w0 + r * (16u * kWave) + 16ull * lane);
    if (tid == 0) tx[-1] = c0 ? text[c0 - 1] : (uint8_t)'\n';
    if (tid < 8) sh.text[(kLnPad + kLnChunk) / 4u + tid] = 0x41414141u;
    for (uint32_t i = tid; i < kLnMaxQ; i += kLnThreads) sh.marks[i] = 0;
    uint32_t cm[kLnRows];  // bit j = byte j of the lane's piece is a candidate
#pragma unroll
    for (uint32_t r = 0; r < kLnRows; ++r) {
        const uint32_t rel = w0 + r * (16u * kWave) + 16u * (uint32_t)lane;
        uint32_t m = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint32_t x = raw[r].w[g];
            const uint32_t vb = (raw[r].valid >> (4 * g)) & 0xfu;
            if (vb != 0xfu) {  // behind the end of the text: bytes that are no candidates
                const uint32_t keep = (vb & 1u ? 0xffu : 0u) | (vb & 2u ? 0xff00u : 0u) | (vb & 4u ? 0xff0000u : 0u) |
                                      (vb & 8u ? 0xff000000u : 0u);
                x = (x & keep) | (0x41414141u & ~keep);
                raw[r].w[g] = x;
            }
            // bit 7 of a byte of y is set unless the byte is below 0x0E
            const uint32_t y = ((x & 0x7f7f7f7fu) + 0x72727272u) | x;
            const uint32_t z = (~y & 0x80808080u) >> 7;
            m |= (((z * 0x00204081u) >> 21) & 0xfu) << (4 * g);
        }
        cm[r] = m;
        *reinterpret_cast<uint4 *>(tx + rel) = make_uint4(raw[r].w[0], raw[r].w[1], raw[r].w[2], raw[r].w[3]);
    }
    // candidates before each lane's piece, row by row (two packed scans), and before the wave
    const uint32_t p01 = (uint32_t)__builtin_popcount(cm[0]) | ((uint32_t)__builtin_popcount(cm[1]) << 16);
    const uint32_t p23 = (uint32_t)__builtin_popcount(cm[2]) | ((uint32_t)__builtin_popcount(cm[3]) << 16);
    const uint32_t i01 = wave_scan_sum(p01), i23 = wave_scan_sum(p23);
    const uint32_t t01 = (uint32_t)__builtin_amdgcn_readlane((int)i01, kWave - 1);
    const uint32_t t23 = (uint32_t)__builtin_amdgcn_readlane((int)i23, kWave - 1);
    if (lane == 0) sh.cnt[wave] = (t01 & 0xffffu) + (t01 >> 16) + (t23 & 0xffffu) + (t23 >> 16);
    __syncthreads();
    uint32_t E = 0, before = 0;
#pragma unroll
    for (int w = 0; w < kLnWaves; ++w) {
        if (w < wave) before += sh.cnt[w];
        E += sh.cnt[w];
    }
    if (E > kLnMaxSeg - 1u) {
        // too many short lines for the tables: the host repeats the text with the three-pass kernels; the successors
        // must not wait for this chunk
        give_up();
        return;
    }
    {
        const uint32_t e01 = i01 - p01, e23 = i23 - p23;
        const uint32_t row_base[kLnRows] = {before, before + (t01 & 0xffffu), before + (t01 & 0xffffu) + (t01 >> 16),
                                            before + (t01 & 0xffffu) + (t01 >> 16) + (t23 & 0xffffu)};
        const uint32_t lane_ex[kLnRows] = {e01 & 0xffffu, e01 >> 16, e23 & 0xffffu, e23 >> 16};
#pragma unroll
        for (uint32_t r = 0; r < kLnRows; ++r) {
            const uint32_t rel = w0 + r * (16u * kWave) + 16u * (uint32_t)lane;
            uint32_t idx = row_base[r] + lane_ex[r], m = cm[r];
            while (__ballot(m != 0u)) {
                if (m) {
                    sh.list[idx] = (uint16_t)(rel + (uint32_t)__builtin_ctz(m));
                    ++idx;
                    m &= m - 1u;
                }
            }
        }
    }
    __syncthreads();
    if (debug & 4u) return;

    // ---- A2: one thread per segment.  Segment l lies between candidates l - 1 and l (the chunk's ends for the first
    // and the last one); a candidate that is '\n' or '\r' belongs to neither side, any other one to the segment behind it.
    // Everything here is independent of the chunks before: the table holds every segment that may hold sequence, in
    // order, and the classes U / V / K are runs of it (see lookback_lines), so the look-back's answer only cuts the
    // front off.
    // (TWO consecutive segments per thread and round: the text of a 60-column FASTA file has about 270 segments per
    // chunk, and the workgroup-wide scans of a round cost the same for 14 segments as for 256)
    constexpr uint32_t kPerRound = 2u * kLnThreads;
    const uint32_t nseg = E + 1u, rounds = (nseg + kPerRound - 1u) / kPerRound;  // <= 2
    uint32_t ln_run = 0, lr_run = 0;  // latest line start / record start so far, as 1 + segment index (0 = none)
    unsigned long long run = 0;       // sequence bytes | table entries << 16 | record starts << 32 so far
    for (uint32_t rd = 0; rd < rounds; ++rd) {
        uint32_t start[2], end[2], am[2], bm[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t l = rd * kPerRound + 2u * tid + (uint32_t)h;
            start[h] = end[h] = am[h] = bm[h] = 0;
            if (l < nseg) {
                uint32_t prev = tx[-1];  // the byte in front of the segment's first byte
                if (l > 0) {
                    const uint32_t pp = sh.list[l - 1];
                    prev = tx[pp];
                    const bool sep = prev == (uint32_t)'\n' || prev == (uint32_t)'\r';
                    start[h] = pp + (sep ? 1u : 0u);
                }
                end[h] = l < E ? (uint32_t)sh.list[l] : L;
                if (prev == (uint32_t)'\n') {  // a line starts here
                    am[h] = l + 1u;
                    if (start[h] < L && tx[start[h]] == (uint8_t)'>') bm[h] = l + 1u;
                }
            }
        }
        uint32_t xa, xb, ta, tb;
        block_prev_marks<kLnWaves>(am[1] ? am[1] : am[0], bm[1] ? bm[1] : bm[0], sh.s, xa, xb, ta, tb);
        uint32_t ea = xa ? xa : ln_run, eb = xb ? xb : lr_run;  // latest marks before the thread's first segment
        unsigned long long v[2];
        bool cand[2], is_rec[2];
        uint32_t first_a[2], first_b[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t l = rd * kPerRound + 2u * tid + (uint32_t)h;
            const uint32_t la = am[h] ? am[h] : ea, lb = bm[h] ? bm[h] : eb;  // ... up to and including this segment
            const bool valid = l < nseg;
            is_rec[h] = valid && bm[h] != 0u;
            cand[h] = valid && end[h] > start[h] && !(lb != 0u && lb == la);  // not inside a header of this chunk
            v[h] = (cand[h] ? (unsigned long long)(end[h] - start[h]) | (1ull << 16) : 0ull) | (is_rec[h] ? (1ull << 32) : 0ull);
            first_a[h] = am[h] && !ea;  // the chunk's first line start / record start
            first_b[h] = bm[h] && !eb;
            ea = la;
            eb = lb;
        }
        unsigned long long tot;
        unsigned long long ex = run + block_sum_excl64(v[0] + v[1], sh.s64, tot);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t o = (uint32_t)(ex & 0xffffu);
            if (cand[h]) sh.tab[(uint32_t)(ex >> 16) & 0xffffu] = o | (start[h] << 16);
            if (is_rec[h] && ((uint32_t)(ex >> 32) & 0xffffu) < kLnMaxRec) sh.recs[(uint32_t)(ex >> 32) & 0xffffu] = o | (start[h] << 16);
            if (first_a[h]) sh.u_end = (uint32_t)ex;  // everything before the first line start is class U
            if (first_b[h]) sh.v_end = (uint32_t)ex;  // U and V end at the first record start
            ex += v[h];
        }
        run += tot;
        ln_run = ta ? ta : ln_run;
        lr_run = tb ? tb : lr_run;
    }
    const uint32_t all_b = (uint32_t)(run & 0xffffu), all_t = (uint32_t)(run >> 16) & 0xffffu, nr = (uint32_t)(run >> 32) & 0xffffu;
    if (nr > kLnMaxRec) {  // (uniform) more records than the table holds: as for too many segments
        give_up();
        return;
    }
    if (tid == 0) {
        sh.tab[all_t] = all_b;  // sentinel: where the last segment ends
        if (!ln_run) sh.u_end = (uint32_t)run;
        if (!lr_run) sh.v_end = (uint32_t)run;
    }
    __syncthreads();
    if (debug & 8u) return;
    const uint32_t u_b = sh.u_end & 0xffffu, u_t = sh.u_end >> 16, uv_b = sh.v_end & 0xffffu, uv_t = sh.v_end >> 16;
    const uint32_t kind = (lr_run && lr_run == ln_run) ? 2u : (ln_run ? 1u : 0u);
    if (tid == 0 && bid != 0 && !(debug & 1u))
        ln_publish(st_ctx, bid, all_b - uv_b, uv_b - u_b, u_b, nr, kind, lr_run != 0u);  // the successors can go on

    // ---- B: one lane per output dword; a wave takes a contiguous run of 64-dword rows and keeps its dwords in
    // registers.  The layout depends on what the look-back returns - the bit the chunk's first base lands on
    // (r0 = bases before the chunk mod 16) and whether classes U / V count - so B runs BEFORE the look-back on the
    // common answer (nothing dropped; r0 = 0, put right by one funnel shift between neighbouring lanes when the
    // stores go out) and is repeated the slow way when the answer differs: the chunk's predecessors get the time of B
    // to publish, instead of the chunk waiting for them with nothing to do.
    constexpr int kMaxRows = (int)((kLnMaxQ + kWave - 1) / kWave + kLnWaves - 1) / kLnWaves;  // 5
    uint32_t acc[kMaxRows];
    uint32_t row0 = 0, row1 = 0;
    // marks: a segment signs the first dword boundary it covers (dword q starts at chunk base max(0, 16 q - r))
    auto make_marks = [&](const uint32_t *tab, uint32_t T, uint32_t drop_b, uint32_t r) {
        for (uint32_t t = tid; t < T; t += kLnThreads) {
            const uint32_t o = (tab[t] & 0xffffu) - drop_b, on = (tab[t + 1] & 0xffffu) - drop_b;
            const uint32_t q = o ? (o + r + 15u) >> 4 : 0u;
            const uint32_t bq = q ? 16u * q - r : 0u;
            if (bq < on) sh.marks[q] = (uint16_t)(t + 1u);
        }
    };
    // dwords [0, nq) of the layout with the first base at bit 2 r of dword 0
    auto gather = [&](const uint32_t *tab, uint32_t drop_b, uint32_t nb, uint32_t r, uint32_t nq) {
        const uint32_t qrows = (nq + kWave - 1u) / kWave, per_wave = (qrows + kLnWaves - 1u) / kLnWaves;
        row0 = (uint32_t)wave * per_wave;
        row1 = row0 + per_wave < qrows ? row0 + per_wave : qrows;
        if (row0 >= row1) {
            row0 = row1 = 0;
            return;
        }
        uint32_t carry = 1;  // latest mark below the wave's first dword
        if (row0 > 0) {
            for (int base = (int)(row0 * kWave) - 1;; base -= kWave) {
                const int idx = base - lane;
                const uint32_t v = idx >= 0 ? (uint32_t)sh.marks[idx] : 1u;  // (dword 0 always carries a mark)
                const unsigned long long m = __ballot(v != 0u);
                if (m) {
                    carry = (uint32_t)__shfl((int)v, __builtin_ctzll(m), kWave);
                    break;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < kMaxRows; ++i) {
            acc[i] = 0;
            const uint32_t row = row0 + (uint32_t)i;
            if (row >= row1) continue;  // (uniform)
            const uint32_t q = row * kWave + (uint32_t)lane;
            const uint32_t mk = sh.marks[q < kLnMaxQ ? q : 0u];
            uint32_t b = q ? 16u * q - r : 0u;  // first base of the dword, counted from the chunk's first base
            const bool live = q < nq && b < nb;
            const uint32_t mv = live ? mk : 0u;
            const unsigned long long m = __ballot(mv != 0u);
            const unsigned long long le = m & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
            const uint32_t from = (uint32_t)__shfl((int)mv, le ? 63 - __builtin_clzll(le) : 0, kWave);
            uint32_t t = (le ? from : carry) - 1u;
            if (m) carry = (uint32_t)__shfl((int)mv, 63 - __builtin_clzll(m), kWave);
            const uint32_t pd = (b + r) & 15u;
            uint32_t need = 0;
            if (live) need = (16u - pd) < (nb - b) ? (16u - pd) : (nb - b);
            uint32_t sft = 2u * pd, a32 = 0;
            while (__ballot(need != 0u)) {
                if (need) {
                    const uint32_t e = tab[t], en = tab[t + 1];
                    const uint32_t o = (e & 0xffffu) - drop_b, on = (en & 0xffffu) - drop_b;
                    uint32_t take = need < on - b ? need : on - b;
                    if (take == 0u) take = need;  // (cannot happen - every table entry holds a base; never spin on it)
                    const uint32_t addr = kLnPad + (e >> 16) + (b - o);
                    const uint32_t a4 = addr >> 2, bs = (addr & 3u) * 8u;
                    const uint32_t d0 = sh.text[a4], d1 = sh.text[a4 + 1], d2 = sh.text[a4 + 2], d3 = sh.text[a4 + 3],
                                   d4 = sh.text[a4 + 4];
                    uint32_t v = pack16(__builtin_amdgcn_alignbit(d1, d0, bs), __builtin_amdgcn_alignbit(d2, d1, bs),
                                        __builtin_amdgcn_alignbit(d3, d2, bs), __builtin_amdgcn_alignbit(d4, d3, bs));
                    if (take < 16u) v &= (1u << (2u * take)) - 1u;
                    a32 |= v << sft;
                    sft += 2u * take;
                    b += take;
                    need -= take;
                    ++t;
                }
            }
            acc[i] = a32;
        }
    };
    const uint32_t nq_spec = all_b ? ((all_b + 15u) >> 4) + 1u : 0u;  // (one more: the shift spills into it)
    if (!(debug & 2u)) {
        make_marks(sh.tab, all_t, 0u, 0u);
        __syncthreads();
        gather(sh.tab, 0u, all_b, 0u, nq_spec);
        if (lane == kWave - 1 && row1 > row0) {
            uint32_t last = 0;
#pragma unroll
            for (int i = 0; i < kMaxRows; ++i)
                if (row0 + (uint32_t)i + 1u == row1) last = acc[i];
            sh.edge[wave] = last;
        }
    }
    if (wave == 0) {
        LnPrefix pf;
        if (debug & 1u) {  // timing experiment (wrong results): no look-back
            pf.h = 0;
            pf.started = 1;
            pf.bases = (unsigned long long)bid * 16000ull;
            pf.recs = 0;
        } else {
            pf = lookback_lines(st_ctx, bid, all_b - uv_b, uv_b - u_b, u_b, nr, kind, lr_run != 0u, error, (debug & 16u) != 0u);
            if (debug & 16u) {
                pf.h = 0;
                pf.started = 1;
                pf.bases = (unsigned long long)bid * 16000ull;
                pf.recs = 0;
            }
        }
        if (lane == 0) {
            sh.ctx = pf.h | (pf.started << 1);
            sh.off[0] = pf.bases;
            sh.off[1] = pf.recs;
        }
    }
    __syncthreads();
    // the look-back's answer: which classes count.  Dropped segments are a run at the front of the table.
    const uint32_t h_in = sh.ctx & 1u, started_in = (sh.ctx >> 1) & 1u;
    const uint32_t drop_b = started_in ? (h_in ? u_b : 0u) : uv_b, drop_t = started_in ? (h_in ? u_t : 0u) : uv_t;
    const uint32_t nb = all_b - drop_b, T = all_t - drop_t;
    const unsigned long long G0 = sh.off[0], R0 = sh.off[1];
    const uint32_t r0 = (uint32_t)(G0 & 15ull);
    const uint32_t nq = nb ? (r0 + nb + 15u) >> 4 : 0u;  // output dwords this chunk writes to
    // record table: a record's bases start at the output offset reached at its header
    for (uint32_t i = tid; i < nr; i += kLnThreads) {
        const unsigned long long r = R0 + i;
        if (r < max_records) {
            rec_base[r] = G0 + ((sh.recs[i] & 0xffffu) - drop_b);
            if (rec_pos) rec_pos[r] = c0 + (sh.recs[i] >> 16);
        }
    }
    if (tid == 0 && bid == n_chunks - 1) {
        counts[0] = G0 + nb;
        counts[1] = R0 + nr;
        if (R0 + nr <= max_records) rec_base[R0 + nr] = G0 + nb;
    }
    if (debug & 2u) return;
    uint32_t shift = 2u * r0;  // what the dwords in registers still have to move up by
    if (drop_b != 0u) {        // (uniform) the other answer: once more, with the layout as it is
        for (uint32_t i = tid; i < kLnMaxQ; i += kLnThreads) sh.marks[i] = 0;
        __syncthreads();
        make_marks(sh.tab + drop_t, T, drop_b, r0);
        __syncthreads();
        gather(sh.tab + drop_t, drop_b, nb, r0, nq);
        shift = 0;
    }
    // ---- stores: whole dwords in order; the two dwords shared with the neighbouring chunks are OR-ed in
    const unsigned long long D0 = G0 >> 4;
    const uint32_t last_partial = (r0 + nb) & 15u;
#pragma unroll
    for (int i = 0; i < kMaxRows; ++i) {
        const uint32_t row = row0 + (uint32_t)i;
        if (row >= row1) continue;
        const uint32_t q = row * kWave + (uint32_t)lane;
        uint32_t v = acc[i];
        if (shift) {
            uint32_t below = (uint32_t)__shfl_up((int)v, 1, kWave);
            const uint32_t edge = i > 0 ? (uint32_t)__builtin_amdgcn_readlane((int)acc[i > 0 ? i - 1 : 0], kWave - 1)
                                        : (row0 > 0 ? sh.edge[wave > 0 ? wave - 1 : 0] : 0u);
            if (lane == 0) below = edge;
            v = (v << shift) | (below >> (32u - shift));
        }
        const unsigned long long dw = D0 + q;
        if (q < nq && dw < out_dwords) {
            const bool partial = (q == 0 && r0 != 0u) || (q == nq - 1u && last_partial != 0u);
            if (partial) {
                if (v) atomicOr(&out32[dw], v);
            } else {
                out32[dw] = v;
            }
        }
    }
}

}  // namespace

uint64_t fasta_chunks(uint64_t n_bytes) { return (n_bytes + kChunkBytes - 1) / kChunkBytes; }
static uint64_t fasta_onepass_chunks(uint64_t n_bytes) { return (n_bytes + kLnChunk - 1) / kLnChunk; }
// scratch: six arrays of chunks + 1 64-bit words (three-pass kernels); the one-pass kernel keeps three status words
// per 16 KB chunk in the same area
uint64_t fasta_scratch_bytes(uint64_t n_bytes) {
    const uint64_t a = 6 * (fasta_chunks(n_bytes) + 1), b = 3 * (fasta_onepass_chunks(n_bytes) + 1);
    return (a > b ? a : b) * sizeof(unsigned long long);
}

int launch_fasta_pack(const uint8_t *d_text, uint64_t n_bytes, uint8_t *d_packed, uint64_t packed_capacity_bytes,
                      unsigned long long *d_rec_base, unsigned long long *d_rec_pos, uint64_t max_records,
                      unsigned long long *d_counts, void *scratch, hipStream_t stream, bool one_pass, uint32_t *d_error) {
    const uint64_t chunks = fasta_chunks(n_bytes);
    if (chunks == 0 || chunks >= (1ull << 31)) return -1;
    unsigned long long *a = reinterpret_cast<unsigned long long *>(scratch);
    unsigned long long *last_nl = a, *last_rec = a + (chunks + 1), *ctx_nl = a + 2 * (chunks + 1),
                       *ctx_rec = a + 3 * (chunks + 1), *cnt_b = a + 4 * (chunks + 1), *cnt_r = a + 5 * (chunks + 1);
    const uint64_t out_dwords = packed_capacity_bytes / 4;
    // the packed bytes are OR-ed together where chunks meet: clear what the text can fill at most
    const uint64_t clear = packed_capacity_bytes < (n_bytes + 3) / 4 + 8 ? packed_capacity_bytes : (n_bytes + 3) / 4 + 8;
    if (clear && hipMemsetAsync(d_packed, 0, clear, stream) != hipSuccess) return -1;
    if (one_pass) {
        const uint64_t oc = fasta_onepass_chunks(n_bytes);
        if (oc >= (1ull << 31)) return -1;
        const char *de = mm_exp_env("MM_FASTA_DEBUG");  // (experiments build only: wrong results by design)
        const uint32_t dbg = de ? (uint32_t)atoi(de) : 0u;
        if (hipMemsetAsync(a, 0, (oc + 1) * sizeof(unsigned long long), stream) != hipSuccess) return -1;
        hipLaunchKernelGGL(fasta_lines_kernel, dim3((uint32_t)oc), dim3(kLnThreads), 0, stream, d_text, n_bytes, a,
                           reinterpret_cast<uint32_t *>(d_packed), out_dwords, d_rec_base,
                           d_rec_pos, max_records, d_counts, (uint32_t)oc, d_error, dbg);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    hipLaunchKernelGGL(fasta_marks_kernel, dim3((uint32_t)chunks), dim3(kBlockThreads), 0, stream, d_text, n_bytes,
                       last_nl, last_rec);
    hipLaunchKernelGGL(fasta_scan2_kernel<MaxOp>, dim3(1), dim3(kScanThreads), 0, stream, last_nl, last_rec, chunks,
                       0ull, ctx_nl, ctx_rec);
    hipLaunchKernelGGL(fasta_walk_kernel<false>, dim3((uint32_t)chunks), dim3(kBlockThreads), 0, stream, d_text,
                       n_bytes, ctx_nl, ctx_rec, cnt_b, cnt_r, nullptr, nullptr, nullptr, 0ull, nullptr, nullptr, 0ull,
                       nullptr, (uint32_t)chunks);
    // (the counts are scanned in place of the marks, which are no longer needed)
    hipLaunchKernelGGL(fasta_scan2_kernel<AddOp>, dim3(1), dim3(kScanThreads), 0, stream, cnt_b, cnt_r, chunks, 0ull,
                       last_nl, last_rec);
    hipLaunchKernelGGL(fasta_walk_kernel<true>, dim3((uint32_t)chunks), dim3(kBlockThreads), 0, stream, d_text, n_bytes,
                       ctx_nl, ctx_rec, nullptr, nullptr, last_nl, last_rec, reinterpret_cast<uint32_t *>(d_packed),
                       out_dwords, d_rec_base, d_rec_pos, max_records, d_counts, (uint32_t)chunks);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace mm
