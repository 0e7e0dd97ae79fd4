// mm_fused_impl.h — the fused gfx950 minimizer kernel (one launch per sequence range).
//
// Formulation (MI355X-first, not a translation of the reference's 8-lane loop):
//
//   phase 0  the workgroup (256 lanes = 4 wave64) stages the 2-bit bases of its tile of
//            NB = 256*S windows (+ k+w halo) into LDS with coalesced dword loads.
//   phase 1  every lane walks S = nblk*W consecutive windows serially, everything in
//            registers: rolling ntHash through one LDS table look-up per base
//            (t_in_out[(out<<2)|in], 16 x uint2), keys (hash_hi16 | pos16) for the leftmost
//            minimum and the complemented key for the rightmost one, two-stacks sliding
//            minimum over blocks of W with the ring held in W registers (W is a template
//            parameter), incremental strand vote.  One byte per window -- the offset of the
//            chosen k-mer inside its window -- goes to LDS.  Reference semantics:
//            src/sliding_min.rs:86-212, src/canonical.rs:12-31, src/minimizers.rs:117-128.
//   phase 2  lane-parallel over the tile's window bytes: SWAR predicate (adjacent dedup
//            src/collect.rs:15-37, or the syncmer filter src/syncmers.rs:33-37), ordered
//            compaction (wave scan), a decoupled look-back scan across workgroups for the
//            global output offset, and the u32 stores.  Output order == window order.
//
// No MFMA: this is integer / byte work bounded by VALU issue and HBM, not GEMM-shaped.
#pragma once
#include "mm_common.h"

namespace mm {

struct FusedParams {
    SeqView seq;
    HashTables ht;
    uint32_t k;
    uint32_t nblk;       // W-blocks per lane; S = W * nblk windows per lane
    uint32_t win_begin;  // window range [win_begin, win_end)
    uint32_t win_end;
    uint32_t mode;
    uint32_t n_in_dwords;  // staged input dwords per workgroup
    uint32_t lds_in_off;   // byte offset of the staged input in dynamic LDS
    uint32_t lds_tab_off;  // byte offset of the hash tables in dynamic LDS
    OutParams out;
};

__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) {
    return (a & mask) | (b & ~mask);
}

// bit 7 of each byte set iff that byte of v is zero
__device__ __forceinline__ uint32_t swar_zero_bytes(uint32_t v) {
    return ~(((v & 0x7f7f7f7fu) + 0x7f7f7f7fu) | v) & 0x80808080u;
}
// compress bits 7,15,23,31 into bits 0..3
__device__ __forceinline__ uint32_t swar_pack4(uint32_t f) {
    return (((f >> 7) * 0x00204081u) >> 21) & 0xfu;
}

// Flags of 16 consecutive windows from their offset bytes.
//   mode 0: window differs from its predecessor  <=>  off[i] + 1 != off[i-1]
//   mode 1: closed syncmer                        <=>  off[i] == 0 || off[i] == W-1
//   mode 2: open syncmer                          <=>  off[i] == W/2
template <int W>
__device__ __forceinline__ uint32_t window_flags16(const uint4 x, uint32_t prev_byte,
                                                   uint32_t mode, bool force_first) {
    const uint32_t X[4] = {x.x, x.y, x.z, x.w};
    uint32_t f16 = 0;
    if (mode == 0) {
        uint32_t prevw = prev_byte << 24;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            uint32_t y = __builtin_amdgcn_alignbit(X[m], prevw, 24);  // (X << 8) | (prev >> 24)
            uint32_t d = (X[m] + 0x01010101u) ^ y;
            uint32_t nz = ~swar_zero_bytes(d) & 0x80808080u;
            f16 |= swar_pack4(nz) << (4 * m);
            prevw = X[m];
        }
        if (force_first) f16 |= 1u;
    } else if (mode == 1) {
        constexpr uint32_t kLast = 0x01010101u * (uint32_t)((W - 1) & 0xff);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            uint32_t z = swar_zero_bytes(X[m]) | swar_zero_bytes(X[m] ^ kLast);
            f16 |= swar_pack4(z) << (4 * m);
        }
    } else {
        constexpr uint32_t kMid = 0x01010101u * (uint32_t)((W / 2) & 0xff);
#pragma unroll
        for (int m = 0; m < 4; ++m) f16 |= swar_pack4(swar_zero_bytes(X[m] ^ kMid)) << (4 * m);
    }
    return f16;
}

template <int W, bool CANON, bool HASH_RC>
__global__ __launch_bounds__(kBlockThreads) void fused_kernel(const FusedParams p) {
    static_assert(W >= 1 && W <= 255, "window offsets are stored as bytes");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint32_t s_bid;
    __shared__ uint32_t s_prev_off;  // offset byte of the window preceding the tile (0x100: none)
    __shared__ uint32_t s_wave_tot[kWavesPerBlock];
    __shared__ unsigned long long s_excl;

    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    if (tid == 0) s_bid = atomicAdd(p.out.ticket, 1u);
    __syncthreads();
    const uint32_t bid = s_bid;

    const uint32_t S = (uint32_t)W * p.nblk;
    const uint32_t NB = kBlockThreads * S;
    const uint64_t bw0 = (uint64_t)p.win_begin + (uint64_t)bid * NB;  // first window of the tile
    const uint32_t nvalid =
        (uint32_t)(((uint64_t)p.win_end - bw0) < NB ? ((uint64_t)p.win_end - bw0) : NB);

    uint8_t *offs = smem;
    uint32_t *inw = reinterpret_cast<uint32_t *>(smem + p.lds_in_off);
    uint2 *tab = reinterpret_cast<uint2 *>(smem + p.lds_tab_off);  // [0..15] in/out, [16..19] in

    // ---------------------------------------------------------------- phase 0
    // Element 0 of lane 0 is the k-mer one position before the tile (its window is the
    // dedup predecessor of the tile's first window).
    const long long pblk = (long long)p.seq.base0 + (long long)bw0 - 1;
    const long long q0 = pblk >> 4;
    const uint32_t rb0 = (uint32_t)(pblk - (q0 << 4));
    for (uint32_t i = tid; i < p.n_in_dwords; i += kBlockThreads)
        inw[i] = load_dword_clamped(p.seq, q0 + i);
    if (tid < 16) tab[tid] = p.ht.t_in_out[tid];
    else if (tid < 20) tab[tid] = p.ht.t_in[tid - 16];
    __syncthreads();

    // ---------------------------------------------------------------- phase 1
    const uint32_t lw = (uint32_t)tid * S;  // first window of this lane, tile-relative
    if (lw < nvalid) {
        const uint32_t rot = p.ht.rot;
        const uint32_t k = p.k;
        const uint32_t rb = rb0 + lw;  // staged-base index of the first base of element 0
        uint32_t fw = 0, rc = 0;
        // hash of element 0: k add-only steps
        for (uint32_t j = 0; j < k; ++j) {
            uint32_t pos = rb + j;
            uint32_t c = (inw[pos >> 4] >> (2u * (pos & 15u))) & 3u;
            uint2 t = tab[16 + c];
            fw = rotl32(fw, rot) ^ t.x;
            if (HASH_RC) rc = rotr32(rc, rot) ^ t.y;
        }

        uint32_t ring_l[W], ring_r[W];
        // stream cursors (staged-base index of the first base the stream yields in this block)
        uint32_t pos_in = rb + k;   // base entering the hash at step e: rb + k + e
        uint32_t pos_out = rb;      // base leaving the hash at step e:  rb + e
        constexpr int NSUB = (W + 15) / 16;

        // ---- block 0: fill the ring with the keys of elements 0..W-1 (no complete window yet)
        {
            uint32_t me[NSUB], mo[NSUB];
#pragma unroll
            for (int g = 0; g < NSUB; ++g) {
                uint32_t pa = pos_in + 16 * g, pr = pos_out + 16 * g;
                uint32_t wa = __builtin_amdgcn_alignbit(inw[(pa >> 4) + 1], inw[pa >> 4], 2u * (pa & 15u));
                uint32_t wr = __builtin_amdgcn_alignbit(inw[(pr >> 4) + 1], inw[pr >> 4], 2u * (pr & 15u));
                me[g] = bfi(0x33333333u, wa, wr << 2);
                mo[g] = bfi(0x33333333u, wa >> 2, wr);
            }
#pragma unroll
            for (int j = 0; j < W; ++j) {
                const uint32_t h = HASH_RC ? fw + rc : fw;
                const uint32_t kl = (h & 0xffff0000u) | (uint32_t)j;
                ring_l[j] = kl;
                if (CANON) ring_r[j] = kl ^ 0xffff0000u;
                const int jj = j & 15, g = j >> 4, m = jj >> 1;
                const uint32_t mw = (jj & 1) ? mo[g] : me[g];
                const uint32_t a8 = (m == 0 ? (mw << 3) : (mw >> (4 * m - 3))) & 0x78u;
                const uint2 t = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint8_t *>(tab) + a8);
                fw = rotl32(fw, rot) ^ t.x;
                if (HASH_RC) rc = rotr32(rc, rot) ^ t.y;
            }
#pragma unroll
            for (int j = W - 2; j >= 0; --j) {
                ring_l[j] = min(ring_l[j], ring_l[j + 1]);
                if (CANON) ring_r[j] = max(ring_r[j], ring_r[j + 1]);
            }
            pos_in += W;
            pos_out += W;
        }

        // strand vote state: cnt = #(T|G) in the l bases of the window about to complete
        int cnt = 0;
        const uint32_t l = k + (uint32_t)W - 1;
        const int thr = (int)(l / 2);
        if (CANON) {
            // window -1 covers staged bases [rb, rb + l)
            uint32_t c = 0;
            for (uint32_t q = rb >> 4; q <= (rb + l - 1) >> 4; ++q) {
                uint32_t wd = inw[q] & 0xAAAAAAAAu;
                uint32_t lo = q << 4;
                if (lo < rb) wd &= ~0u << (2u * (rb - lo));
                if (lo + 16 > rb + l) wd &= ~0u >> (2u * (lo + 16 - (rb + l)));
                c += __popc(wd);
            }
            cnt = (int)c;
        }
        if (tid == 0) {
            // predecessor window of the tile: the minimum of block 0
            uint32_t sel = ring_l[0];
            if (CANON) sel = (cnt > thr) ? ring_l[0] : ring_r[0];
            s_prev_off = (bw0 == 0) ? 0x100u : (sel & 0xffffu);
        }
        uint32_t pos_r2 = rb;  // base leaving the strand window after window i: rb + i + 1 ... see below
        if (CANON) {
            // move cnt from window -1 to window 0: + base rb+l, - base rb
            uint32_t pa = rb + l;
            cnt += (int)((inw[pa >> 4] >> (2u * (pa & 15u) + 1u)) & 1u);
            cnt -= (int)((inw[rb >> 4] >> (2u * (rb & 15u) + 1u)) & 1u);
            pos_r2 = rb + 1;  // window 0 -> 1 drops base rb+1
        }

        // ---- blocks 1..nblk: one window per step
        uint8_t *my_offs = offs + lw;
        for (uint32_t b = 1; b <= p.nblk; ++b) {
            uint32_t me[NSUB], mo[NSUB], wa_[NSUB], w2_[NSUB];
#pragma unroll
            for (int g = 0; g < NSUB; ++g) {
                uint32_t pa = pos_in + 16 * g, pr = pos_out + 16 * g;
                uint32_t wa = __builtin_amdgcn_alignbit(inw[(pa >> 4) + 1], inw[pa >> 4], 2u * (pa & 15u));
                uint32_t wr = __builtin_amdgcn_alignbit(inw[(pr >> 4) + 1], inw[pr >> 4], 2u * (pr & 15u));
                me[g] = bfi(0x33333333u, wa, wr << 2);
                mo[g] = bfi(0x33333333u, wa >> 2, wr);
                if (CANON) {
                    uint32_t p2 = pos_r2 + 16 * g;
                    wa_[g] = wa;
                    w2_[g] = __builtin_amdgcn_alignbit(inw[(p2 >> 4) + 1], inw[p2 >> 4], 2u * (p2 & 15u));
                }
            }
            const uint32_t e0 = b * (uint32_t)W;  // element index of step j = 0
            uint32_t pl = 0, pr_ = 0;
#pragma unroll
            for (int j = 0; j < W; ++j) {
                const uint32_t h = HASH_RC ? fw + rc : fw;
                const uint32_t kl = (h & 0xffff0000u) | (e0 + (uint32_t)j);
                const uint32_t kr = kl ^ 0xffff0000u;
                pl = (j == 0) ? kl : min(pl, kl);
                uint32_t sel = (j + 1 < W) ? min(pl, ring_l[(j + 1 < W) ? j + 1 : 0]) : pl;
                ring_l[j] = kl;
                if (CANON) {
                    pr_ = (j == 0) ? kr : max(pr_, kr);
                    uint32_t selr = (j + 1 < W) ? max(pr_, ring_r[(j + 1 < W) ? j + 1 : 0]) : pr_;
                    ring_r[j] = kr;
                    sel = (cnt > thr) ? sel : selr;
                }
                // window i = e - W starts at element i + 1; offset of the chosen k-mer inside it
                const uint32_t off = (sel & 0xffffu) - (e0 + (uint32_t)j - (uint32_t)W + 1u);
                my_offs[(b - 1) * (uint32_t)W + (uint32_t)j] = (uint8_t)off;

                const int jj = j & 15, g = j >> 4, m = jj >> 1;
                const uint32_t mw = (jj & 1) ? mo[g] : me[g];
                const uint32_t a8 = (m == 0 ? (mw << 3) : (mw >> (4 * m - 3))) & 0x78u;
                const uint2 t = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint8_t *>(tab) + a8);
                fw = rotl32(fw, rot) ^ t.x;
                if (HASH_RC) rc = rotr32(rc, rot) ^ t.y;
                if (CANON) {
                    cnt += (int)((wa_[g] >> (2 * jj + 1)) & 1u);
                    cnt -= (int)((w2_[g] >> (2 * jj + 1)) & 1u);
                }
            }
#pragma unroll
            for (int j = W - 2; j >= 0; --j) {
                ring_l[j] = min(ring_l[j], ring_l[j + 1]);
                if (CANON) ring_r[j] = max(ring_r[j], ring_r[j + 1]);
            }
            pos_in += W;
            pos_out += W;
            pos_r2 += W;
        }
    }
    __syncthreads();

    // ---------------------------------------------------------------- phase 2
    // Each wave owns a contiguous quarter of the tile's 16-window groups.
    const uint32_t ngroups = (nvalid + 15u) / 16u;
    const uint32_t gpw = (ngroups + kWavesPerBlock - 1) / kWavesPerBlock;  // groups per wave
    const uint32_t g_begin = wave * gpw;
    const uint32_t g_end = (g_begin + gpw < ngroups) ? g_begin + gpw : ngroups;
    const uint32_t prev_tile = s_prev_off;
    const uint32_t mode = p.mode;

    auto group_flags = [&](uint32_t g) -> uint32_t {
        const uint4 x = *reinterpret_cast<const uint4 *>(offs + 16u * g);
        const uint32_t prevb = (g == 0) ? (prev_tile & 0xffu) : (uint32_t)offs[16u * g - 1u];
        uint32_t f = window_flags16<W>(x, prevb, mode, g == 0 && prev_tile == 0x100u);
        const uint32_t rem = nvalid - 16u * g;
        if (rem < 16u) f &= (1u << rem) - 1u;
        return f;
    };

    // 2a: count
    uint32_t my_cnt = 0;
    for (uint32_t g = g_begin + lane; g < g_end; g += kWave) my_cnt += __popc(group_flags(g));
    uint32_t wave_incl = wave_inclusive_sum(my_cnt);
    if (lane == kWave - 1) s_wave_tot[wave] = wave_incl;
    __syncthreads();
    uint32_t wave_base = 0, block_total = 0;
#pragma unroll
    for (int v = 0; v < kWavesPerBlock; ++v) {
        uint32_t t = s_wave_tot[v];
        if (v < wave) wave_base += t;
        block_total += t;
    }
    if (wave == 0) {
        unsigned long long carry = (bid == 0) ? *p.out.total : 0ull;
        unsigned long long e = lookback_exclusive(p.out.status, bid, block_total, carry);
        if (lane == 0) s_excl = e;
    }
    __syncthreads();

    // 2b: emit, wave by wave in group order
    unsigned long long run = s_excl + wave_base;
    for (uint32_t gb = g_begin; gb < g_end; gb += kWave) {
        const uint32_t g = gb + lane;
        uint32_t f = (g < g_end) ? group_flags(g) : 0u;
        const uint32_t c = __popc(f);
        const uint32_t incl = wave_inclusive_sum(c);
        unsigned long long dst = run + (incl - c);
        while (f) {
            const uint32_t bit = __builtin_ctz(f);
            f &= f - 1u;
            const uint32_t wi = 16u * g + bit;
            const uint32_t widx = (uint32_t)(bw0 + wi);
            if (dst < p.out.cap) {
                p.out.pos[dst] = (mode == 0) ? widx + (uint32_t)offs[wi] : widx;
                if (p.out.sk) p.out.sk[dst] = widx;
            }
            ++dst;
        }
        run += __shfl(incl, kWave - 1, kWave);
    }
    if (tid == 0 && bid == gridDim.x - 1) *p.out.total = s_excl + block_total;
}

}  // namespace mm
