// mm_fused_impl.h — the fused gfx950 minimizer kernel (one launch per sequence range).
//
// Formulation (MI355X-first, not a translation of the reference's 8-lane loop):
//
//   phase 1  every lane of a 256-lane workgroup walks S = nblk*W consecutive windows serially,
//            everything in registers.  The lane's 2-bit bases come straight from HBM/L2 through
//            bounds-checked raw buffer loads (three funnel-shifted 16-base views per W-block:
//            base entering the hash, base leaving the hash, base leaving the strand window),
//            prefetched one W-block ahead.  Rolling ntHash costs ONE LDS table look-up per base
//            (s_tab[(out<<2)|in], 16 x uint2 = forward / reverse-complement contribution);
//            keys are (hash_hi16 | pos16) for the leftmost minimum and the complemented key for
//            the rightmost one; the two-stacks sliding minimum runs over blocks of W with the
//            ring held in W registers (W is a template parameter); the strand vote is
//            incremental.  The lane decides in place whether a window emits (adjacent dedup
//            against its own previous window, or the syncmer predicate), shifts the flag into a
//            per-block bit mask, counts it, and stores the offset of the chosen k-mer inside the
//            window to LDS (4 bits per window when W <= 16, else a byte).  Reference semantics:
//            src/sliding_min.rs:86-212, src/canonical.rs:12-31, src/minimizers.rs:117-128,
//            src/collect.rs:15-37, src/syncmers.rs:33-37.
//   phase 2  each wave compacts the windows of its own 64 lanes: DPP prefix sum over the flag
//            words, a decoupled look-back across workgroups for the global output offset, window
//            indices staged in LDS and then written as fully coalesced u32 stores.
//            Output order == window order.
//
// No MFMA: this is integer / byte work bounded by VALU issue and HBM, not GEMM-shaped.
#pragma once
#include "mm_common.h"

namespace mm {

constexpr uint32_t kStageCap = 384;  // staged outputs (u32) per wave and phase-2 iteration

struct FusedParams {
    SeqView seq;
    HashTables ht;
    uint32_t k;
    uint32_t nblk;       // W-blocks per lane; S = W * nblk windows per lane
    uint32_t win_begin;  // window range [win_begin, win_end)
    uint32_t win_end;
    uint32_t nblk_inv;       // ceil(2^32 / nblk): blk / nblk == umulhi(blk, nblk_inv) for blk < 2^16
    uint32_t lds_stage_off;  // byte offset of the per-wave staging buffers in dynamic LDS
    uint32_t use_ticket;     // 1: tile id from an atomic ticket (safe mode), 0: blockIdx.x
    uint32_t debug;          // timing experiments only (MM_DEBUG env): 1 no look-back, 2 no emit,
                             // 4 no phase 1
    OutParams out;
};

// inclusive prefix sum over the 64 lanes of a wave with DPP row shifts / broadcasts
__device__ __forceinline__ uint32_t wave_scan_dpp(uint32_t v) {
    v += __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
    v += __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
    return v;
}

__device__ __forceinline__ uint32_t min3u(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t max3u(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_max3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Geometry shared by the kernel and its launcher.
// Per lane and W-block the kernel keeps NPL dwords in LDS, stored as planes of kPlane dwords
// (one dword per lane + 1 pad, so both the lane-serial writes of phase 1 and the block-order reads
// of phase 2 are bank-conflict free):
//   planes [0, NW)          chosen-k-mer offsets: 4 bits per window when W <= 16 (low nibble of
//                            the lane-relative element index), else one byte per window
//   flag words              emit flags, LSB = first window; packed into the spare high bits of
//                            the last offset plane when they fit (FLAG_PACKED), else NSEG planes
template <int W>
struct FusedGeom {
    static constexpr bool NIB = (W <= 16);
    static constexpr int NW = NIB ? (W + 7) / 8 : (W + 3) / 4;
    static constexpr int NSEG = (W + 31) / 32;
    static constexpr bool FLAG_PACKED = NIB && (W % 8 != 0) && ((W % 8) * 4 + W <= 32);
    static constexpr int FLAG_SHIFT = FLAG_PACKED ? (W % 8) * 4 : 0;
    static constexpr int NPL = NW + (FLAG_PACKED ? 0 : NSEG);
    static constexpr int G = (W <= 32) ? (32 / W) : 1;   // W-blocks combined per phase-2 item
    static constexpr int NSUB = (W + 15) / 16;           // 16-base view words per W-block
};
constexpr uint32_t kPlane = kFusedThreads + 1;  // dwords per plane

template <int W, bool CANON, bool HASH_RC, int MODE>
__global__ __launch_bounds__(kFusedThreads) void fused_kernel(const FusedParams p) {
    static_assert(W >= 1 && W <= 255, "window offsets are stored in at most a byte");
    using GE = FusedGeom<W>;
    constexpr bool NIB = GE::NIB, FLAG_PACKED = GE::FLAG_PACKED;
    constexpr int NW = GE::NW, NPL = GE::NPL, NSEG = GE::NSEG, G = GE::G, NSUB = GE::NSUB;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // static LDS: distinct objects, so table look-ups can be scheduled across the dynamic stores
    __shared__ uint2 s_tab[20];  // [0..15] (out<<2)|in, [16..19] in only (warm-up)
    __shared__ uint32_t s_bid;
    __shared__ uint32_t s_wave_tot[kFusedWaves];
    __shared__ unsigned long long s_excl;

    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    // Tile id.  Default: blockIdx.x (workgroups are dispatched in index order on gfx950, which
    // the look-back needs for forward progress; its spins are bounded and report a violation,
    // upon which the host re-runs in ticket mode where an atomic counter defines the order).
    if (tid == 0) s_bid = p.use_ticket ? atomicAdd(p.out.ticket, 1u) : blockIdx.x;
    if (tid < 16) s_tab[tid] = p.ht.t_in_out[tid];
    else if (tid < 20) s_tab[tid] = p.ht.t_in[tid - 16];
    __syncthreads();
    const uint32_t bid = __builtin_amdgcn_readfirstlane(s_bid);  // keep tile scalars in SGPRs

    const uint32_t nblk = p.nblk;
    const uint32_t S = (uint32_t)W * nblk;
    const uint32_t NB = kFusedThreads * S;
    const uint64_t bw0 = (uint64_t)p.win_begin + (uint64_t)bid * NB;  // first window of the tile
    const uint32_t nvalid =
        (uint32_t)(((uint64_t)p.win_end - bw0) < NB ? ((uint64_t)p.win_end - bw0) : NB);
    const bool partial = nvalid < NB;

    uint32_t *planes = reinterpret_cast<uint32_t *>(smem);  // [nblk][NPL][kPlane]
    uint32_t *stage = reinterpret_cast<uint32_t *>(smem + p.lds_stage_off) + wave * kStageCap;

    // ---------------------------------------------------------------- phase 1
    const uint32_t lw = (uint32_t)tid * S;  // first window of this lane, tile-relative
    uint32_t my_count = 0;
    if (lw < nvalid && !(p.debug & 4u)) {
        // Element 0 of a lane is the k-mer one position before its first window (that window is
        // the dedup predecessor).  P0 = first base of the tile's element 0 in dword-array
        // coordinates; it is -1 only for the very first window of an unshifted buffer.
        const long long P0 = (long long)p.seq.base0 + (long long)bw0 - 1;
        const long long Q0 = P0 >> 4;
        const long long Q0c = Q0 < 0 ? 0 : Q0;
        const int32_t prel0 = (int32_t)(P0 - (Q0c << 4));  // -1 .. 15
        // Bounds-checked view of the packed sequence from dword Q0c on: dwords past the end read
        // as 0, so the halo after the last base needs no clamping.
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint32_t *>(p.seq.d + Q0c), 0, (int)(((long long)p.seq.n_dwords - Q0c) * 4), 0x00020000);
        const int32_t pb = prel0 + (int32_t)lw;  // first base of this lane's element 0
        // 16-base view starting at tile-relative base position pos >= 0
        auto view = [&](int32_t pos) -> uint32_t {
            const auto d = __builtin_amdgcn_raw_buffer_load_b64(rsrc, ((uint32_t)pos >> 4) << 2, 0, 0);
            return __builtin_amdgcn_alignbit(d[1], d[0], 2u * ((uint32_t)pos & 15u));
        };
        // same, but pos may be -1 (only at a lane's start): the missing base reads as code 0
        auto view_first = [&](int32_t pos) -> uint32_t {
            const bool neg = pos < 0;
            const auto d = __builtin_amdgcn_raw_buffer_load_b64(rsrc, neg ? 0u : (((uint32_t)pos >> 4) << 2), 0, 0);
            return __builtin_amdgcn_alignbit(neg ? d[0] : d[1], neg ? 0u : d[0], 2u * ((uint32_t)pos & 15u));
        };

        const uint32_t rot_l = (32u - p.ht.rot) & 31u;  // alignbit amount for rotl(x, rot)
        const uint32_t rot_r = p.ht.rot & 31u;
        const uint32_t k = p.k;
        uint32_t fw = 0, rc = 0;
        const uint8_t *tabb = reinterpret_cast<const uint8_t *>(s_tab);

        // hash of element 0: k add-only steps, 16 bases per view word
        for (uint32_t g = 0; g * 16u < k; ++g) {
            const uint32_t wa = g == 0 ? view_first(pb) : view(pb + 16 * (int32_t)g);
            const uint32_t rem = k - 16u * g;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                if ((uint32_t)jj < rem) {
                    const uint32_t a8 = (jj == 0 ? (wa << 3) : (jj == 1 ? (wa << 1) : (wa >> (2 * jj - 3)))) & 0x18u;
                    const uint2 t = *reinterpret_cast<const uint2 *>(tabb + 128 + a8);
                    fw = __builtin_amdgcn_alignbit(fw, fw, rot_l) ^ t.x;
                    if (HASH_RC) rc = __builtin_amdgcn_alignbit(rc, rc, rot_r) ^ t.y;
                }
            }
        }

        uint32_t ring_l[W], ring_r[W];
        int32_t pos_in = pb + (int32_t)k;  // base entering the hash at step e: pb + k + e
        int32_t pos_out = pb;              // base leaving the hash at step e:  pb + e
        // 0xffff0000 kept in a VGPR so that (h & mask) | e is one v_and_or_b32 with e in an SGPR
        uint32_t kmask;
        asm volatile("v_mov_b32 %0, 0xffff0000" : "=v"(kmask));

        uint32_t va[NSUB], vr[NSUB], v2[NSUB];  // views of the block being processed
#pragma unroll
        for (int g = 0; g < NSUB; ++g) {
            va[g] = view(pos_in + 16 * g);
            vr[g] = g == 0 ? view_first(pos_out) : view(pos_out + 16 * g);
            v2[g] = 0;
        }

        // ---- block 0: keys of elements 0..W-1 fill the ring (no complete window yet)
        {
            uint32_t me[NSUB], mo[NSUB];
#pragma unroll
            for (int g = 0; g < NSUB; ++g) {
                me[g] = (va[g] & 0x33333333u) | ((vr[g] << 2) & 0xccccccccu);
                mo[g] = ((va[g] >> 2) & 0x33333333u) | (vr[g] & 0xccccccccu);
            }
            pos_in += W;
            pos_out += W;
            // prefetch block 1 (its strand stream starts at pb + 1)
#pragma unroll
            for (int g = 0; g < NSUB; ++g) {
                va[g] = view(pos_in + 16 * g);
                vr[g] = view(pos_out + 16 * g);
                if (CANON) v2[g] = view(pb + 1 + 16 * g);
            }
#pragma unroll
            for (int j = 0; j < W; ++j) {
                const uint32_t h = HASH_RC ? fw + rc : fw;
                const uint32_t kl = (h & kmask) | (uint32_t)j;
                ring_l[j] = kl;
                if (CANON) ring_r[j] = kl ^ kmask;
                const int jj = j & 15, g = j >> 4, m = jj >> 1;
                const uint32_t mw = (jj & 1) ? mo[g] : me[g];
                const uint32_t a8 = (m == 0 ? (mw << 3) : (mw >> (4 * m - 3))) & 0x78u;
                const uint2 t = *reinterpret_cast<const uint2 *>(tabb + a8);
                fw = __builtin_amdgcn_alignbit(fw, fw, rot_l) ^ t.x;
                if (HASH_RC) rc = __builtin_amdgcn_alignbit(rc, rc, rot_r) ^ t.y;
            }
#pragma unroll
            for (int j = W - 2; j >= 0; --j) {
                ring_l[j] = min(ring_l[j], ring_l[j + 1]);
                if (CANON) ring_r[j] = max(ring_r[j], ring_r[j + 1]);
            }
        }

        // strand vote: cnt - (#steps done) = #(T|G) among the l bases of the current window;
        // each step adds tg(in) + 1 - tg(leaving base), the threshold moves by 1 per step.
        int cnt = 0;
        const uint32_t l = k + (uint32_t)W - 1;
        const int thr = (int)(l / 2);
        uint32_t prev;             // key of the predecessor window's k-mer (mode 0)
        int32_t pos_r2 = pb + 1;   // window 0 -> 1 drops base pb + 1
        if (CANON) {
            // window -1 covers bases [pb, pb + l)
            uint32_t c = 0;
            for (uint32_t g = 0; g * 16u < l; ++g) {
                uint32_t wd = (g == 0 ? view_first(pb) : view(pb + 16 * (int32_t)g)) & 0xAAAAAAAAu;
                const uint32_t rem = l - 16u * g;
                if (rem < 16u) wd &= (1u << (2u * rem)) - 1u;
                c += __popc(wd);
            }
            cnt = (int)c;
            prev = (cnt > thr) ? ring_l[0] : ring_r[0];
            // move to window 0: + base pb + l, - base pb
            cnt += (int)((view(pb + (int32_t)l) >> 1) & 1u);
            cnt -= (int)((view_first(pb) >> 1) & 1u);
        } else {
            prev = ring_l[0];
        }
        if (bw0 + lw == 0) prev = 0xffffffffu;  // the very first window has no predecessor

        // ---- blocks 1..nblk: one window per step
        uint32_t *pl_blk = planes + tid;  // this lane's column; advances NPL planes per block
        int rem_valid = (int)nvalid - (int)lw;  // windows of this lane still inside the range
        for (uint32_t b = 1; b <= nblk; ++b) {
            uint32_t me[NSUB], mo[NSUB], tgw[NSUB];
#pragma unroll
            for (int g = 0; g < NSUB; ++g) {
                me[g] = (va[g] & 0x33333333u) | ((vr[g] << 2) & 0xccccccccu);
                mo[g] = ((va[g] >> 2) & 0x33333333u) | (vr[g] & 0xccccccccu);
                // 2-bit fields: tg(in) + 1 - tg(leaving) in {0,1,2}
                if (CANON) tgw[g] = ((va[g] >> 1) & 0x55555555u) + (~(v2[g] >> 1) & 0x55555555u);
            }
            pos_in += W;
            pos_out += W;
            pos_r2 += W;
            // prefetch the views of the next block (a harmless over-read after the last block)
#pragma unroll
            for (int g = 0; g < NSUB; ++g) {
                va[g] = view(pos_in + 16 * g);
                vr[g] = view(pos_out + 16 * g);
                if (CANON) v2[g] = view(pos_r2 + 16 * g);
            }

            const uint32_t e0 = b * (uint32_t)W;  // element index of step j = 0
            uint32_t pl = 0, pr_ = 0, fmask = 0, acc = 0;
#pragma unroll
            for (int j = 0; j < W; ++j) {
                const uint32_t e = e0 + (uint32_t)j;  // uniform
                const uint32_t h = HASH_RC ? fw + rc : fw;
                const uint32_t kl = (h & kmask) | e;
                // prefix minimum over the block so far and the window minimum; odd steps fold the
                // previous key in with one v_min3 (3 ops per 2 steps and side instead of 4)
                uint32_t sel;
                if (j == 0) {
                    pl = kl;
                    sel = (W > 1) ? min(kl, ring_l[(W > 1) ? 1 : 0]) : kl;
                } else if (j & 1) {
                    // pl still excludes key j: sel = min3(pl, kl, ring[j+1]); pl is updated on even steps
                    sel = (j + 1 < W) ? min3u(pl, kl, ring_l[(j + 1 < W) ? j + 1 : 0]) : min(pl, kl);
                } else {
                    pl = min3u(pl, ring_l[j - 1], kl);  // ring_l[j-1] holds key j-1
                    sel = (j + 1 < W) ? min(pl, ring_l[(j + 1 < W) ? j + 1 : 0]) : pl;
                }
                ring_l[j] = kl;
                if (CANON) {
                    const uint32_t kr = kl ^ kmask;
                    uint32_t selr;
                    if (j == 0) {
                        pr_ = kr;
                        selr = (W > 1) ? max(kr, ring_r[(W > 1) ? 1 : 0]) : kr;
                    } else if (j & 1) {
                        selr = (j + 1 < W) ? max3u(pr_, kr, ring_r[(j + 1 < W) ? j + 1 : 0]) : max(pr_, kr);
                    } else {
                        pr_ = max3u(pr_, ring_r[j - 1], kr);
                        selr = (j + 1 < W) ? max(pr_, ring_r[(j + 1 < W) ? j + 1 : 0]) : pr_;
                    }
                    ring_r[j] = kr;
                    sel = (cnt > thr + (int)(e - (uint32_t)W)) ? sel : selr;
                }
                // window i = e - W starts at element i + 1.  The emit flag is shifted into fmask
                // with v_cmp + v_addc (fmask = 2*fmask + flag).
                if (MODE == 0) {
                    asm("v_cmp_ne_u16 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                        : "+v"(fmask) : "v"(sel), "v"(prev) : "vcc");
                    prev = sel;
                    // Offsets are shifted into acc from the top (v_alignbit): 4 bits (low nibble of
                    // the chosen element index; phase 2 subtracts the window's own index) or 8 bits
                    // (offset inside the window) per step; a full dword goes to its plane.
                    if (NIB) {
                        acc = __builtin_amdgcn_alignbit(sel, acc, 4);
                        if ((j & 7) == 7 || j == W - 1) {
                            const int cntn = (j & 7) + 1;
                            uint32_t wv = cntn == 8 ? acc : (acc >> (32 - 4 * cntn));
                            if (!(FLAG_PACKED && j == W - 1)) pl_blk[(j >> 3) * kPlane] = wv;
                            else acc = wv;  // stored together with the flags below
                        }
                    } else {
                        acc = __builtin_amdgcn_alignbit(sel - (e - (uint32_t)W + 1u), acc, 8);
                        if ((j & 3) == 3 || j == W - 1) {
                            const int cntb = (j & 3) + 1;
                            pl_blk[(j >> 2) * kPlane] = cntb == 4 ? acc : (acc >> (32 - 8 * cntb));
                        }
                    }
                } else if (MODE == 1) {
                    const uint32_t first = e - (uint32_t)W + 1u;
                    unsigned long long t;
                    asm("v_cmp_eq_u16 vcc, %3, %2\n\tv_cmp_eq_u16 %1, %2, %4\n\ts_or_b64 vcc, vcc, %1\n\t"
                        "v_addc_co_u32 %0, vcc, %0, %0, vcc"
                        : "+v"(fmask), "=&s"(t) : "v"(sel), "s"(first), "s"(e) : "vcc");
                } else {
                    const uint32_t mid = e - (uint32_t)W + 1u + (uint32_t)(W / 2);
                    asm("v_cmp_eq_u16 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                        : "+v"(fmask) : "v"(sel), "s"(mid) : "vcc");
                }
                if ((j & 31) == 31 || j == W - 1) {
                    const int seg = j >> 5;
                    const int seglen = (seg == NSEG - 1) ? (W - 32 * seg) : 32;
                    // LSB = first window of the segment
                    uint32_t f = __builtin_bitreverse32(fmask) >> (32 - seglen);
                    if (partial) {
                        const int v = rem_valid - 32 * seg;
                        f = v <= 0 ? 0u : (v >= seglen ? f : (f & ((1u << v) - 1u)));
                    }
                    my_count += __popc(f);
                    if (FLAG_PACKED) pl_blk[(NW - 1) * kPlane] = (MODE == 0 ? acc : 0u) | (f << GE::FLAG_SHIFT);
                    else pl_blk[(NW + seg) * kPlane] = f;
                    fmask = 0;
                }

                const int jj = j & 15, g = j >> 4, m = jj >> 1;
                const uint32_t mw = (jj & 1) ? mo[g] : me[g];
                const uint32_t a8 = (m == 0 ? (mw << 3) : (mw >> (4 * m - 3))) & 0x78u;
                const uint2 t = *reinterpret_cast<const uint2 *>(tabb + a8);
                fw = __builtin_amdgcn_alignbit(fw, fw, rot_l) ^ t.x;
                if (HASH_RC) rc = __builtin_amdgcn_alignbit(rc, rc, rot_r) ^ t.y;
                if (CANON) cnt += (int)((tgw[g] >> (2 * jj)) & 3u);
            }
#pragma unroll
            for (int j = W - 2; j >= 0; --j) {
                ring_l[j] = min(ring_l[j], ring_l[j + 1]);
                if (CANON) ring_r[j] = max(ring_r[j], ring_r[j + 1]);
            }
            pl_blk += NPL * kPlane;
            rem_valid -= W;
        }
    }

    // ---------------------------------------------------------------- phase 2
    // Wave w owns the windows of its own 64 lanes (a contiguous quarter of the tile), so the
    // only cross-wave exchange is the four wave totals.
    const uint32_t wave_total = __builtin_amdgcn_readlane(wave_scan_dpp(my_count), kWave - 1);
    if (lane == 0) s_wave_tot[wave] = wave_total;
    __syncthreads();
    uint32_t wave_base = 0, block_total = 0;
#pragma unroll
    for (int v = 0; v < kFusedWaves; ++v) {
        const uint32_t t = s_wave_tot[v];
        if (v < wave) wave_base += t;
        block_total += t;
    }
    if (wave == 0) {
        const unsigned long long carry = (bid == 0) ? *p.out.total : 0ull;
        const unsigned long long ex = (p.debug & 1u) ? (unsigned long long)bid * (NB / 6u)
                                                     : lookback_exclusive(p.out.status, bid, block_total, carry, p.out.error);
        if (lane == 0) s_excl = ex;
    }
    __syncthreads();

    // Items: G consecutive W-blocks (W <= 32) or one 32-window segment (W > 32), in window order
    // (lane-major: W-block blk of the wave belongs to lane blk / nblk, block blk % nblk).
    const unsigned long long run0 = s_excl + wave_base;
    uint32_t run = 0;  // outputs of this wave emitted so far
    const uint32_t wave_blocks = kWave * nblk;
    const uint32_t items = (NSEG == 1) ? (wave_blocks + G - 1) / G : wave_blocks * NSEG;
    const uint32_t bw0_lo = (uint32_t)bw0;
    const uint32_t tid0 = (uint32_t)wave * kWave;
    uint32_t *outp = p.out.pos;
    uint32_t *outs = p.out.sk;
    for (uint32_t it0 = 0; it0 < items && !(p.debug & 2u); it0 += kWave) {
        const uint32_t it = it0 + lane;
        uint32_t fg[G], ent[G];  // flags and entry prefix (bb << 8 | tid << 16) per combined block
        uint32_t c = 0;
        uint32_t first_win = 0xffffffffu;  // tile-relative first window of the item
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const uint32_t blk = (NSEG == 1) ? it * G + g : it / NSEG;
            const uint32_t seg = (NSEG == 1) ? 0u : it - blk * NSEG;
            const uint32_t lw_ = __umulhi(blk, p.nblk_inv);  // wave-relative lane
            const uint32_t bb = blk - lw_ * nblk;
            const uint32_t tl = tid0 + lw_;
            const uint32_t win = tl * S + bb * (uint32_t)W + 32u * seg;
            if (g == 0) first_win = win;
            uint32_t f = 0;
            // blocks of lanes past the valid range were never written
            if (it < items && blk < wave_blocks && win < nvalid) {
                const uint32_t w = planes[(bb * NPL + (FLAG_PACKED ? NW - 1 : NW + seg)) * kPlane + tl];
                f = FLAG_PACKED ? (w >> GE::FLAG_SHIFT) : w;
            }
            fg[g] = f;
            ent[g] = (bb << 8) | (tl << 16) | (32u * seg);
            c += __popc(f);
        }
        if (__builtin_amdgcn_readfirstlane(first_win) >= nvalid) break;  // items are in window order
        const uint32_t incl = wave_scan_dpp(c);
        const uint32_t total = __builtin_amdgcn_readlane(incl, kWave - 1);
        const unsigned long long base = run0 + run;
        // value of one emitted window from its entry (j | bb << 8 | tid << 16)
        auto emit_value = [&](uint32_t e, uint32_t &widx) -> uint32_t {
            const uint32_t j = e & 0xffu, bb = (e >> 8) & 0xffu, tl = e >> 16;
            const uint32_t lwin = bb * (uint32_t)W + j;  // lane-relative window
            widx = bw0_lo + tl * S + lwin;
            if (MODE != 0) return widx;
            if (NIB) {
                const uint32_t wv = planes[(bb * NPL + (j >> 3)) * kPlane + tl];
                return widx + (((wv >> (4u * (j & 7u))) - (lwin + 1u)) & 15u);
            }
            const uint32_t wv = planes[(bb * NPL + (j >> 2)) * kPlane + tl];
            return widx + ((wv >> (8u * (j & 3u))) & 0xffu);
        };
        if (total <= kStageCap) {
            // stage the entries in window order, then coalesced stores
            uint32_t slot = incl - c;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                uint32_t f = fg[g];
                while (f) {
                    const uint32_t bit = __builtin_ctz(f);
                    f &= f - 1u;
                    stage[slot++] = ent[g] + bit;
                }
            }
            for (uint32_t i = lane; i < total; i += kWave) {
                uint32_t widx;
                const uint32_t val = emit_value(stage[i], widx);
                if (base + i < p.out.cap) {
                    outp[base + i] = val;
                    if (MODE == 0 && outs) outs[base + i] = widx;
                }
            }
        } else {
            // dense region (more than kStageCap outputs in one iteration): direct ordered stores
            unsigned long long dst = base + (incl - c);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                uint32_t f = fg[g];
                while (f) {
                    const uint32_t bit = __builtin_ctz(f);
                    f &= f - 1u;
                    uint32_t widx;
                    const uint32_t val = emit_value(ent[g] + bit, widx);
                    if (dst < p.out.cap) {
                        outp[dst] = val;
                        if (MODE == 0 && outs) outs[dst] = widx;
                    }
                    ++dst;
                }
            }
        }
        run += total;
    }
    if (tid == 0 && bid == gridDim.x - 1) *p.out.total = s_excl + block_total;
}

}  // namespace mm
