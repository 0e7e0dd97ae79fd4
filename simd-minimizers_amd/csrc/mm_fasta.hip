// mm_fasta.hip — FASTA text -> PackedSeq records on the device, the loader step in front of the hot path
// (the reference's bench harness does it on the CPU: needletail::parse_fastx_file + PackedSeqVec::from_ascii per
// record, bench/src/lib.rs:51-82).  Semantics restated here (needletail is not in the tree: parity unpinned, see
// DESIGN.md): a record starts with '>' at the start of a line; its header runs to the end of that line; its
// sequence is every following line up to the next header line with '\n' and '\r' removed; bytes before the first
// header are ignored; every sequence byte is packed as (c >> 1) & 3 like PackedSeqVec::from_ascii.
//
// All records are packed back to back into ONE 2-bit buffer (record r = bases [rec_base[r], rec_base[r+1]) of it,
// any base offset - what mm_run_batch_device takes), so the job is a stream compaction of the text:
//   K1  per 32 KB chunk: position of its last '\n' and of its last record start (context-free: a '>' right after
//       a '\n' starts a record whatever came before)
//   S1  exclusive max-scan of both over the chunks (one workgroup) -> the line / record context at every chunk start
//   K2  per chunk, now with its context: number of sequence bytes and of record starts
//   S2  exclusive sum-scan of both
//   K3  per chunk: 2-bit codes of its sequence bytes to their final place (staged in LDS, whole dwords stored,
//       the two dwords a chunk shares with its neighbours OR-ed in), record table entries
// Three passes over the text; for 3.1 GB that is ~10 GB of HBM reads against 56 ms of PCIe time to bring the
// text in, so the passes are kept simple rather than fused behind a look-back.
#include "mm_common.h"
#include "mm_launch.h"

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
        s[kWavesPerBlock + wave] = lb;
    }
    __syncthreads();
    ta = tb = 0;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) {
        const uint32_t va = s[w], vb = s[kWavesPerBlock + w];
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

// One-pass kernel, wave-serial (round 3, second version).  A chunk is 16 KB: every wave owns a contiguous 4 KB of it and
// walks its four 1 KB pieces (64 lanes x 16 bytes, coalesced) by itself - the line / record context inside a wave comes
// from a ballot and a shuffle, the offsets from a DPP scan, the packed codes go through a staging area in LDS that
// belongs to the wave - so the workgroup meets at FOUR barriers per chunk (marks, context, counts, offsets) instead
// of five per 4 KB.  The first version kept the three-pass kernels' workgroup-wide iterations behind the look-backs
// and ran 25 % slower than the three passes.
#ifndef MM_FASTA_OP_ITERS
#define MM_FASTA_OP_ITERS 4
#endif
constexpr uint32_t kOpIters = MM_FASTA_OP_ITERS;                  // 1 KB pieces per wave
constexpr uint32_t kOpWaveBytes = 16u * kWave * kOpIters;         // 4 KB
constexpr uint32_t kOpChunkBytes = kOpWaveBytes * kWavesPerBlock;  // 16 KB
constexpr uint32_t kOpStage = 16u * kWave * 2u / 32u + 2u;        // dwords a 1 KB piece can pack into, + 2

// latest mark (position + 1, 0 = none) among the lower lanes of the wave / in the whole wave
__device__ __forceinline__ void wave_prev_mark(uint32_t v, uint32_t &excl, uint32_t &last) {
    const int lane = threadIdx.x & (kWave - 1);
    const unsigned long long m = __ballot(v != 0u);
    const unsigned long long lower = m & ((1ull << lane) - 1ull);
    const uint32_t from_lower = (uint32_t)__shfl((int)v, lower ? 63 - __builtin_clzll(lower) : 0, kWave);
    const uint32_t from_top = (uint32_t)__shfl((int)v, m ? 63 - __builtin_clzll(m) : 0, kWave);
    excl = lower ? from_lower : 0u;
    last = m ? from_top : 0u;
}

__global__ __launch_bounds__(kBlockThreads) void fasta_onepass_kernel(
    const uint8_t *__restrict__ text, uint64_t n, unsigned long long *__restrict__ st_ctx,
    unsigned long long *__restrict__ st_bases, unsigned long long *__restrict__ st_recs, uint32_t *__restrict__ out32,
    uint64_t out_dwords, unsigned long long *__restrict__ rec_base, unsigned long long *__restrict__ rec_pos,
    uint64_t max_records, unsigned long long *__restrict__ counts, uint32_t n_chunks, uint32_t *error) {
    __shared__ uint32_t s_mark[2][kWavesPerBlock];  // last newline / record start of every wave (position + 1 in the chunk)
    __shared__ uint32_t s_cnt[2][kWavesPerBlock];   // sequence bytes / record starts of every wave
    __shared__ uint32_t s_ctx;
    __shared__ unsigned long long s_off[2];
    __shared__ uint32_t s_stage[kWavesPerBlock][kOpStage];
    const uint32_t tid = threadIdx.x, bid = blockIdx.x;
    const int lane = tid & (kWave - 1), wave = tid / kWave;
    const uint64_t c0 = (uint64_t)bid * kOpChunkBytes;
    const uint32_t w0 = (uint32_t)wave * kOpWaveBytes;  // first byte of the wave's part, relative to the chunk
    for (uint32_t i = (uint32_t)lane; i < kOpStage; i += kWave) s_stage[wave][i] = 0;
    Raw raw[kOpIters];
#pragma unroll
    for (uint32_t it = 0; it < kOpIters; ++it) raw[it] = load_raw(text, n, c0 + w0 + it * (16u * kWave) + 16ull * lane);
    // ---- the masks of every piece, made once and kept (the three-pass kernels make them in every pass), and the marks
    // of the wave's part (context-free)
    Piece pc[kOpIters];
    uint32_t pls[kOpIters], prs[kOpIters];
    uint32_t mnl = 0, mrec = 0;
#pragma unroll
    for (uint32_t it = 0; it < kOpIters; ++it) {
        const uint32_t rel = w0 + it * (16u * kWave) + 16u * (uint32_t)lane;
        pc[it] = make_piece(raw[it], text, n, c0 + rel);
        starts(pc[it], pls[it], prs[it]);
        if (pc[it].nl) mnl = rel + top_bit_pos1(pc[it].nl);
        if (prs[it]) mrec = rel + top_bit_pos1(prs[it]);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        mnl = max(mnl, (uint32_t)__shfl_xor((int)mnl, d, kWave));
        mrec = max(mrec, (uint32_t)__shfl_xor((int)mrec, d, kWave));
    }
    if (lane == 0) {
        s_mark[0][wave] = mnl;
        s_mark[1][wave] = mrec;
    }
    __syncthreads();
    if (wave == 0) {
        uint32_t a = 0, b = 0;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) {
            a = max(a, s_mark[0][w]);
            b = max(b, s_mark[1][w]);
        }
        const uint32_t kind = (b > a) ? (uint32_t)kCtxHeader : (a ? (uint32_t)kCtxPlain : 0u);
#ifdef MM_FASTA_NOLB  // timing experiment (wrong results): no look-backs
        const uint32_t ctx = 2u + (kind & 0u);
#else
        const uint32_t ctx = lookback_context(st_ctx, bid, kind, b != 0u, error);
#endif
        if (lane == 0) s_ctx = ctx;
    }
    __syncthreads();
    const uint32_t ctx = s_ctx;
    // context as pseudo-positions (only their order and lr > 0 matter): marks inside the chunk are 3 + position
    unsigned long long ln_run = (ctx & 2u) ? 1ull : 0ull, lr_run = (ctx & 2u) ? ((ctx & 1u) ? 2ull : 1ull) : 0ull;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w)
        if (w < wave) {
            if (s_mark[0][w]) ln_run = 3ull + s_mark[0][w];
            if (s_mark[1][w]) lr_run = 3ull + s_mark[1][w];
        }
    // ---- base masks, counts and offsets inside the wave
    uint32_t bm[kOpIters], ex[kOpIters], tot[kOpIters];
    uint32_t wave_b = 0, wave_r = 0;
#pragma unroll
    for (uint32_t it = 0; it < kOpIters; ++it) {
        const uint32_t rel = w0 + it * (16u * kWave) + 16u * (uint32_t)lane;
        const Piece &p = pc[it];
        const uint32_t ls = pls[it], rs = prs[it];
        const uint32_t tnl = p.nl ? rel + top_bit_pos1(p.nl) : 0u, trec = rs ? rel + top_bit_pos1(rs) : 0u;
        uint32_t xnl, xrec, tnl_w, trec_w;
        wave_prev_mark(tnl, xnl, tnl_w);
        wave_prev_mark(trec, xrec, trec_w);
        const unsigned long long ln = xnl ? 3ull + xnl : ln_run, lr = xrec ? 3ull + xrec : lr_run;
        bm[it] = base_mask(p, ls, rs, ln, lr);
        const uint32_t c = (uint32_t)__builtin_popcount(bm[it]) | ((uint32_t)__builtin_popcount(rs) << 16);
        const uint32_t incl = wave_scan_sum(c);
        ex[it] = incl - c;
        tot[it] = (uint32_t)__builtin_amdgcn_readlane((int)incl, kWave - 1);
        wave_b += tot[it] & 0xffffu;
        wave_r += tot[it] >> 16;
        if (tnl_w) ln_run = 3ull + tnl_w;
        if (trec_w) lr_run = 3ull + trec_w;
    }
    if (lane == 0) {
        s_cnt[0][wave] = wave_b;
        s_cnt[1][wave] = wave_r;
    }
    __syncthreads();
    if (wave < 2) {
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) t += s_cnt[wave][w];
#ifdef MM_FASTA_NOLB
        const unsigned long long e = wave == 0 ? (unsigned long long)bid * 16000ull + (t & 0u) : 0ull;
#else
        const unsigned long long e = lookback_exclusive(wave == 0 ? st_bases : st_recs, bid, t, 0ull, error);
#endif
        if (lane == 0) s_off[wave] = e;
    }
    __syncthreads();
    unsigned long long bases_run = s_off[0], recs_run = s_off[1];
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w)
        if (w < wave) {
            bases_run += s_cnt[0][w];
            recs_run += s_cnt[1][w];
        }
    // ---- pack: the wave's own staging area, no barrier
    uint32_t *stage = s_stage[wave];
#pragma unroll
    for (uint32_t it = 0; it < kOpIters; ++it) {
        const uint32_t rel = w0 + it * (16u * kWave) + 16u * (uint32_t)lane;
        const uint64_t o = c0 + rel;
        const Piece &p = pc[it];
        const uint32_t rs = prs[it];
        const uint32_t m = bm[it], nb = (uint32_t)__builtin_popcount(m);
        const uint32_t tot_b = tot[it] & 0xffffu, tot_r = tot[it] >> 16;
        const unsigned long long g0 = bases_run + (ex[it] & 0xffffu);
        for (uint32_t q = rs, k = 0; q; q &= q - 1u, ++k) {
            const uint32_t j = (uint32_t)__builtin_ctz(q);
            const unsigned long long r = recs_run + (ex[it] >> 16) + k;
            if (r < max_records) {
                rec_base[r] = g0 + (uint32_t)__builtin_popcount(m & ((1u << j) - 1u));
                if (rec_pos) rec_pos[r] = o + j;
            }
        }
        uint32_t v = 0;
        if (m) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint32_t t = (p.w[g] >> 1) & 0x03030303u;
                v |= ((t | (t >> 6) | (t >> 12) | (t >> 18)) & 0xffu) << (8 * g);
            }
            for (uint32_t holes = ~m & 0xffffu; holes;) {
                const uint32_t j = 31u - (uint32_t)__builtin_clz(holes);
                holes &= ~(1u << j);
                const uint32_t low = (1u << (2u * j)) - 1u;
                v = (v & low) | ((v >> 2) & ~low);
            }
        }
        const unsigned long long G0a = bases_run & ~15ull;
        const uint32_t n_stage = (uint32_t)((bases_run - G0a + tot_b + 15ull) / 16ull);  // <= kOpStage - 1
        if (nb) {
            const uint32_t r = (uint32_t)(g0 - G0a), d = r >> 4, bsh = 2u * (r & 15u);
            atomicOr(&stage[d], v << bsh);
            if (bsh && bsh + 2u * nb > 32u) atomicOr(&stage[d + 1], v >> (32u - bsh));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (LDS operations of one wave complete in order)
        for (uint32_t i = (uint32_t)lane; i < n_stage; i += kWave) {
            const unsigned long long dw = G0a / 16ull + i;
            const uint32_t val = stage[i];
            stage[i] = 0;
            if (dw < out_dwords) {
                // the first and the last dword may be shared with the neighbouring piece / wave / chunk
                if (i == 0 || i + 1 == n_stage) {
                    if (val) atomicOr(&out32[dw], val);
                } else {
                    out32[dw] = val;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        bases_run += tot_b;
        recs_run += tot_r;
    }
    if (tid == kBlockThreads - 1 && bid == n_chunks - 1) {  // (the last wave ends where the chunk ends)
        counts[0] = bases_run;
        counts[1] = recs_run;
        if (recs_run <= max_records) rec_base[recs_run] = bases_run;
    }
}

}  // namespace

uint64_t fasta_chunks(uint64_t n_bytes) { return (n_bytes + kChunkBytes - 1) / kChunkBytes; }
static uint64_t fasta_onepass_chunks(uint64_t n_bytes) { return (n_bytes + kOpChunkBytes - 1) / kOpChunkBytes; }
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
        if (hipMemsetAsync(a, 0, 3 * (oc + 1) * sizeof(unsigned long long), stream) != hipSuccess) return -1;
        hipLaunchKernelGGL(fasta_onepass_kernel, dim3((uint32_t)oc), dim3(kBlockThreads), 0, stream, d_text, n_bytes, a,
                           a + (oc + 1), a + 2 * (oc + 1), reinterpret_cast<uint32_t *>(d_packed), out_dwords, d_rec_base,
                           d_rec_pos, max_records, d_counts, (uint32_t)oc, d_error);
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
