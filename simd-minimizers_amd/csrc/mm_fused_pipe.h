// mm_fused_pipe.h — the pipelined flavour of the fused kernel: persistent workgroups, two list
// buffers, copy-out deferred by one tile, look-back in one memory round trip.
//
// Why.  In mm::fused_kernel a tile walks (phase 1), then waits for the tiles before it (look-back)
// and copies its lists out (phase 2) while its registers and LDS sit idle: measured on MI355X that is
// 13 % of the canonical k=21 w=11 run and 36 % of the forward one (MM_DEBUG=3 against 0,
// tools/gpu_ab2.py), because the forward walk needs all its resident waves to keep the VALU busy.
//
// How.  The grid is one workgroup per resident slot (occupancy x CUs, all co-resident).  A workgroup
// takes its tiles from an atomic ticket, one walk ahead of their use (the hardware favours the oldest
// waves of a CU, so resident workgroups run at very different speeds: with a fixed share of tiles each
// the run lasts as long as the slowest - measured 30 % slower than the unpipelined kernel).  The lists of
// the workgroup's round r go to buffer r & 1 and are copied out AFTER the walk of round r + 1, when the
// tile's output offset has long been known:
//
//     walk(t_r)  [one wave runs look-back(t_{r-1}) in the middle of it]  ->  publish count(t_r)
//                ->  copy-out(t_{r-1})  ->  walk(t_{r+1}) ...
//
//   * look-back(t) = base of t's chunk of 1024 tiles + the counts of the tiles before it IN THE SAME
//     CHUNK: at most 1023 independent 4-byte loads (16 per lane), one round trip, no chain of dependent
//     hops - in the middle of the next walk nearly every one of them was published long ago (tickets
//     are handed out in tile order).  The base of chunk c + 1 is published by the workgroup that owns
//     the last tile of chunk c.
//   * the look-back duty rotates over the four waves (wave r & 3 in round r), so that no wave is
//     permanently behind the others; nothing in the loop is a workgroup barrier.  A wave waits only for
//     (a) the duty wave's result before a copy-out and (b) its siblings' totals inside the look-back.
//   * list entries are 8 bits (lanes are at most 255 - w windows long, so positions inside a lane fit a
//     byte): two buffers cost the LDS of one 16-bit buffer.  Rows of the two buffers are interleaved
//     (row c of buffer q at c * 524 + q * 260 bytes), so entries past the capacity of EITHER buffer run
//     off the end of the allocation, never into the other buffer (same overflow contract as the
//     unpipelined kernel: counted, detected at the end of the walk, tile redone storing directly).
//
// Scope: one sequence or window range, minimizers / closed / open syncmers, positions only, w <= 16.
// Batches, reads, super-k-mer indices, skip-ambiguous runs and ticket mode stay on mm::fused_kernel.
// Same reference semantics as mm_fused_impl.h (the walk IS lane_walk).
#pragma once
#include "mm_fused_impl.h"

namespace mm {

constexpr uint32_t kPipeRow = kFusedThreads + 4u;   // bytes of one row (entry c of all 256 lanes) of one buffer
constexpr uint32_t kPipePitch = 2u * kPipeRow + 4u; // 524 = 131 dwords (odd): rows of both buffers interleaved
constexpr unsigned long long kRoundBaseValid = 1ull << 63;
constexpr uint32_t kPipeMaxSpins = 1u << 22;
constexpr uint32_t kPipeChunk = 1024;  // tiles per look-back chunk (16 counts per lane of the look-back wave)

#ifndef MM_PIPE_LOADS
#define MM_PIPE_LOADS 8  // counts loaded per lane and batch in the look-back
#endif

// the previous tile's look-back as the hook of lane_walk
template <class F>
struct PipeHook {
    static constexpr bool kActive = true;
    F &lb;
    uint32_t tp, rp;
    __device__ __forceinline__ void operator()() const { lb(tp, rp); }
};

// Workgroups per CU the register allocation is bounded for: canonical walks 4 (128 VGPRs, like the
// unpipelined kernel for w <= 16), forward walks 5 (102 VGPRs; a few spills outside the W-block loop).
template <bool CANON>
constexpr int kPipeMinBlocks = CANON ? 4 : 5;

template <int W, bool CANON, bool HASH_RC, int MODE, int MINB = kPipeMinBlocks<CANON>>
__global__ __launch_bounds__(kFusedThreads, MINB) void fused_pipe_kernel(const FusedParams p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // the two list buffers
    __shared__ uint2 s_tab[36];
    // per-tile state, four generations deep (a wave can be at most one walk ahead of a sibling, see
    // the header comment; generation = round & 3)
    __shared__ uint32_t s_wave_tot[4][kFusedWaves];
    __shared__ uint32_t s_done[4];       // waves that have finished the walk, cumulative over the rounds
    __shared__ uint32_t s_overflow[4];   // round + 1 of a tile in which a list overflowed
    __shared__ uint32_t s_excl_tag[4];   // round + 1 whose s_excl is valid
    __shared__ unsigned long long s_excl[4];
    __shared__ unsigned long long s_carry;
    __shared__ uint32_t s_tile[4];       // tile of round r (generation r & 3), from the ticket
    __shared__ uint32_t s_tile_tag[4];   // round + 1 whose s_tile is valid

    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    {
        auto lds_end = [](const void *q, size_t bytes) {
            return (uint32_t)reinterpret_cast<uintptr_t>(q) + (uint32_t)bytes;
        };
        uint32_t st_end = lds_end(s_tab, sizeof(s_tab));
        st_end = max(st_end, lds_end(s_wave_tot, sizeof(s_wave_tot)));
        st_end = max(st_end, lds_end(s_done, sizeof(s_done)));
        st_end = max(st_end, lds_end(s_overflow, sizeof(s_overflow)));
        st_end = max(st_end, lds_end(s_excl_tag, sizeof(s_excl_tag)));
        st_end = max(st_end, lds_end(s_excl, sizeof(s_excl)));
        st_end = max(st_end, lds_end(&s_carry, sizeof(s_carry)));
        st_end = max(st_end, lds_end(s_tile, sizeof(s_tile)));
        st_end = max(st_end, lds_end(s_tile_tag, sizeof(s_tile_tag)));
        if ((uint32_t)reinterpret_cast<uintptr_t>(smem) < st_end) {  // layout contract of the list overflow
            if (tid == 0) flag_error(p.out.error, 2u);
            return;
        }
    }
    if (tid < 4) {
        s_done[tid] = 0;
        s_overflow[tid] = 0;
        s_excl_tag[tid] = 0;
        s_tile_tag[tid] = 0;
    }
    const uint32_t M = p.tiles_per_wg;
    if (tid == 0) {
        s_carry = *p.out.total;  // outputs before this launch (append mode); read before any tile ends
        s_tile[0] = M ? blockIdx.x * M : atomicAdd(p.out.ticket, 1u);
        s_tile_tag[0] = 1u;
    }
    if (tid < 16) s_tab[tid] = p.ht.t_in_out[tid];
    else if (tid < 20) s_tab[tid] = p.ht.t_in[tid - 16];
    else if (tid < 36) s_tab[tid] = p.ht.t_in2[tid - 20];
    __syncthreads();

    const uint32_t n_tiles = p.n_tiles;
    const uint32_t S = (uint32_t)W * p.nblk;
    const uint32_t NB = kFusedThreads * S;
    const uint32_t hook_b = p.nblk / 2u + 1u;
    volatile uint32_t *v_done = s_done;
    volatile uint32_t *v_tag = s_excl_tag;
    volatile uint32_t *v_wtot = &s_wave_tot[0][0];
    volatile uint32_t *v_tile = s_tile;
    volatile uint32_t *v_tile_tag = s_tile_tag;

    LaneCtx ctx;
    ctx.tab = s_tab;
    ctx.list_bytes = p.list_cap * kPipePitch;
    ctx.list_used = 0;
    ctx.dst = 0;
    ctx.nblk = p.nblk;
    ctx.seq_d = p.seq.d;
    ctx.seq_dwords = p.seq.n_dwords;
    ctx.min_rem = 0;
    ctx.hook_block = 0;

    // geometry of tile t: first window, windows inside the range; fills the lane context
    struct Tile {
        uint64_t bw0;
        uint32_t nvalid;
        bool partial, lane_active;
    };
    auto setup_tile = [&](uint32_t t) -> Tile {
        Tile tl;
        tl.bw0 = (uint64_t)p.win_begin + (uint64_t)t * NB;
        const uint64_t left = (uint64_t)p.win_end - tl.bw0;
        tl.nvalid = (uint32_t)(left < NB ? left : NB);
        tl.partial = tl.nvalid < NB;
        const uint32_t lw = (uint32_t)tid * S;
        tl.lane_active = lw < tl.nvalid;
        ctx.p0 = (long long)p.seq.base0 + (long long)tl.bw0 - 1;
        ctx.lane_bases = lw;
        ctx.wbase = (uint32_t)tl.bw0 + lw;
        ctx.no_prev = (tl.bw0 + lw == 0);
        ctx.rem_valid = (int)tl.nvalid - (int)lw;
        ctx.abase = 0;
        if (tl.partial) {  // wave-uniform minimum of rem_valid over the walking lanes (two-body walks)
            int m = tl.lane_active ? ctx.rem_valid : 0x7fffffff;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) m = min(m, __shfl_xor(m, d, kWave));
            ctx.min_rem = __builtin_amdgcn_readfirstlane(m);
        }
        return tl;
    };

    // ---- look-back of tile tp (round rp), run by one whole wave.  Exclusive prefix = base of the round
    // + counts of the tiles before tp in its round; every count is one relaxed agent-scope 4-byte word
    // holding count + 1 (0 = not yet published).
    auto lookback = [&](uint32_t tp, uint32_t rp) {
        const uint32_t genp = rp & 3u;
        const uint32_t all_done = 4u * (rp >> 2) + 4u;
        for (uint32_t spins = 0; v_done[genp] < all_done; ++spins) {  // siblings still in the walk of tp
            __builtin_amdgcn_s_sleep(8);
            if (spins > kPipeMaxSpins) {
                flag_error(p.out.error, 1u);
                break;
            }
        }
        const uint32_t chunk = tp / kPipeChunk;
        // g tiles of the chunk precede tp (timing experiment MM_PIPE_DEBUG=1: none are looked at)
        const uint32_t g = (p.debug & 1u) ? 0u : tp % kPipeChunk;
        unsigned long long base;
        if (chunk == 0 || (p.debug & 1u)) {
            base = (p.debug & 1u) ? (unsigned long long)tp * (NB / 6u) : s_carry;
        } else {
            unsigned long long s0 = 0;
            for (uint32_t spins = 0;; ++spins) {
                if (lane == 0) s0 = ld_status(&p.pipe_round_base[chunk]);
                s0 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(s0 >> 32)) << 32) |
                     (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)s0);
                if (s0 & kRoundBaseValid) break;
                if (spins > kPipeMaxSpins) {
                    flag_error(p.out.error, 1u);
                    break;
                }
                __builtin_amdgcn_s_sleep(32);
            }
            base = s0 & ~kRoundBaseValid;
        }
        uint32_t sum = 0;
        uint32_t *cnt = p.pipe_counts + (size_t)chunk * kPipeChunk;
        constexpr int NL = MM_PIPE_LOADS;
        for (uint32_t i0 = 0; i0 < g; i0 += (uint32_t)(kWave * NL)) {
            uint32_t v[NL];
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                const uint32_t idx = i0 + (uint32_t)(kWave * u) + (uint32_t)lane;
                v[u] = idx < g ? __hip_atomic_load(&cnt[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1u;
            }
            for (uint32_t spins = 0;; ++spins) {  // (normally not one round: published half a walk ago)
                bool missing = false;
#pragma unroll
                for (int u = 0; u < NL; ++u) missing = missing || v[u] == 0u;
                if (__ballot(missing) == 0ull) break;
                if (spins > kPipeMaxSpins) {
                    flag_error(p.out.error, 1u);
                    break;
                }
                __builtin_amdgcn_s_sleep(16);
#pragma unroll
                for (int u = 0; u < NL; ++u) {
                    const uint32_t idx = i0 + (uint32_t)(kWave * u) + (uint32_t)lane;
                    if (v[u] == 0u) v[u] = __hip_atomic_load(&cnt[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#pragma unroll
            for (int u = 0; u < NL; ++u) sum += v[u] == 0u ? 0u : v[u] - 1u;
        }
        const uint32_t before = __builtin_amdgcn_readlane(wave_scan_dpp(sum), kWave - 1);
        const unsigned long long excl = base + before;
        uint32_t tile_total = 0;
#pragma unroll
        for (int v = 0; v < kFusedWaves; ++v) tile_total += v_wtot[genp * kFusedWaves + v];
        if (lane == 0) {
            s_excl[genp] = excl;
            v_tag[genp] = rp + 1u;  // LDS operations of one wave complete in order: the tag follows the value
            if (g == kPipeChunk - 1u && tp + 1u < n_tiles)  // the last tile of a chunk: base of the next one
                st_status(&p.pipe_round_base[chunk + 1u], kRoundBaseValid | (excl + tile_total));
            if (tp == n_tiles - 1u) *p.out.total = excl + tile_total;
        }
    };

    // ---- copy-out of tile tp (round rp): every wave copies the 64 lists of its own lanes in lane
    // order (= window order); pk = (first slot within the wave << 9) | length of this lane's list
    auto copy_out = [&](uint32_t tp, uint32_t rp, uint32_t pk) {
        const uint32_t genp = rp & 3u, qp = rp & 1u;
        for (uint32_t spins = 0; v_tag[genp] != rp + 1u; ++spins) {  // the duty wave's look-back (normally long done)
            __builtin_amdgcn_s_sleep(4);
            if (spins > kPipeMaxSpins) {
                flag_error(p.out.error, 1u);
                break;
            }
        }
        uint32_t wave_base = 0;
#pragma unroll
        for (int v = 0; v < kFusedWaves; ++v)
            if (v < wave) wave_base += v_wtot[genp * kFusedWaves + v];
        const uint32_t wave_total = v_wtot[genp * kFusedWaves + wave];
        const bool overflow = s_overflow[genp] == rp + 1u;
        const unsigned long long run0 = s_excl[genp] + wave_base;  // first output slot of this wave
        const uint32_t my_count = pk & 511u, excl_lane = pk >> 9;
        const uint64_t bw0 = (uint64_t)p.win_begin + (uint64_t)tp * NB;
        if (p.debug & 2u) return;  // timing experiment: no copy-out
        if (!overflow) {
            const uint32_t tid0 = (uint32_t)wave * kWave;
            // entry c of lane t of buffer q sits at c * kPipePitch + q * kPipeRow + t: lane `c` of the
            // copying wave reads entry c of list L - 64 different rows, an odd number of dwords apart
            const uint8_t *rd = smem + (uint32_t)lane * kPipePitch + qp * kPipeRow + tid0;
            const uint32_t vb0 = (uint32_t)bw0 + tid0 * S - (MODE == 0 ? 1u : 0u);
            const uint32_t r_lo = __builtin_amdgcn_readfirstlane((uint32_t)run0);
            const uint32_t r_hi = __builtin_amdgcn_readfirstlane((uint32_t)(run0 >> 32));
            const unsigned long long run0_u = ((unsigned long long)r_hi << 32) | r_lo;
            const unsigned long long room = p.out.cap > run0_u ? p.out.cap - run0_u : 0ull;
            const uint32_t room32 = __builtin_amdgcn_readfirstlane(room > 0x3fffffffull ? 0x3fffffffu : (uint32_t)room);
            const __amdgpu_buffer_rsrc_t opos =
                __builtin_amdgcn_make_buffer_rsrc(p.out.pos + run0_u, 0, (int)(room32 * 4u), 0x00020000);
            constexpr int kBatch = 8;
            const bool fast = room32 >= wave_total;
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const unsigned long long obase = (unsigned long long)reinterpret_cast<uintptr_t>(p.out.pos + run0_u);
            u32x4 odesc;
            odesc.x = __builtin_amdgcn_readfirstlane((uint32_t)obase);
            odesc.y = __builtin_amdgcn_readfirstlane((uint32_t)(obase >> 32) & 0xffffu);
            odesc.z = 0x7fffffffu;  // the capacity was checked for the whole wave
            odesc.w = 0x00020000u;
            const uint32_t lane4 = (uint32_t)lane * 4u;
            if (fast) {
#pragma unroll
                for (int L0 = 0; L0 < kWave; L0 += kBatch) {
                    uint32_t ent[kBatch];
#pragma unroll
                    for (int u = 0; u < kBatch; ++u) ent[u] = rd[L0 + u];
#pragma unroll
                    for (int u = 0; u < kBatch; ++u) {
                        const uint32_t pkl = __builtin_amdgcn_readlane(pk, L0 + u);
                        const uint32_t n = pkl & 511u, off = pkl >> 9;
                        const uint32_t val = vb0 + (uint32_t)(L0 + u) * S + ent[u];
                        uint32_t t0;
                        unsigned long long sv;
                        asm volatile(
                            "s_mov_b64 %[sv], exec\n\t"
                            "s_bfm_b64 exec, %[n], 0\n\t"
                            "s_cmp_lt_u32 %[n], 64\n\t"
                            "s_cselect_b64 exec, exec, -1\n\t"
                            "s_lshl_b32 %[t0], %[off], 2\n\t"
                            "buffer_store_dword %[val], %[lane4], %[desc], %[t0] offen " MM_STORE_MOD "\n\t"
                            "s_mov_b64 exec, %[sv]"
                            : [t0] "=&s"(t0), [sv] "=&s"(sv)
                            : [n] "s"(n), [off] "s"(off), [val] "v"(val), [lane4] "v"(lane4), [desc] "s"(odesc)
                            : "scc", "memory");
                    }
                }
            } else {
#pragma unroll 1
                for (int L0 = 0; L0 < kWave; L0 += kBatch) {
                    uint32_t ent[kBatch];
#pragma unroll
                    for (int u = 0; u < kBatch; ++u) ent[u] = rd[L0 + u];
#pragma unroll
                    for (int u = 0; u < kBatch; ++u) {
                        const uint32_t pkl = __builtin_amdgcn_readlane(pk, L0 + u);
                        const uint32_t n = pkl & 511u, off = pkl >> 9;
                        const uint32_t voff = (uint32_t)lane < n ? (off + (uint32_t)lane) * 4u : 0xffffffffu;
                        __builtin_amdgcn_raw_buffer_store_b32(vb0 + (uint32_t)(L0 + u) * S + ent[u], opos, voff, 0,
                                                              MM_STORE_AUX);
                    }
                }
            }
            // lists longer than one wave: the entries from the 65th on, list by list
            for (unsigned long long longer = __ballot(my_count > (uint32_t)kWave); longer; longer &= longer - 1ull) {
                const uint32_t L = (uint32_t)__builtin_ctzll(longer);
                const uint32_t pkl = __builtin_amdgcn_readlane(pk, L);
                const uint32_t n = pkl & 511u, off = pkl >> 9;
                for (uint32_t c = (uint32_t)kWave + lane; c < n; c += kWave) {
                    const uint32_t e1 = rd[L + (c - lane) * kPipePitch];
                    __builtin_amdgcn_raw_buffer_store_b32(vb0 + L * S + e1, opos, (off + c) * 4u, 0, 0);
                }
            }
        } else {
            // some list of the tile overflowed: walk it again, storing straight to the output
            const Tile tl = setup_tile(tp);
            ctx.hook_block = 0;
            if (tl.lane_active) {
                ctx.dst = run0 + excl_lane;
                bool over;
                if (tl.partial) lane_walk<W, CANON, HASH_RC, MODE, false, true, true, false, 1, (int)kPipePitch>(p, ctx, over);
                else lane_walk<W, CANON, HASH_RC, MODE, false, true, false, false, 1, (int)kPipePitch>(p, ctx, over);
            }
        }
    };

    using HookFn = PipeHook<decltype(lookback)>;

    // ---------------------------------------------------------------- rounds
    // tile of round r: s_tile[r & 3], taken from the ticket by wave r & 3 at the start of round r - 1
    auto tile_of_round = [&](uint32_t r) -> uint32_t {
        const uint32_t gen = r & 3u;
        for (uint32_t spins = 0; v_tile_tag[gen] != r + 1u; ++spins) {
            __builtin_amdgcn_s_sleep(2);
            if (spins > kPipeMaxSpins) {
                flag_error(p.out.error, 1u);
                return 0xffffffffu;
            }
        }
        return __builtin_amdgcn_readfirstlane(v_tile[gen]);
    };
    uint32_t pk_prev = 0;
    uint32_t r = 0, t_prev = 0;
    for (;; ++r) {
        const uint32_t t = tile_of_round(r);
        if (t >= n_tiles) break;
        const uint32_t q = r & 1u, gen = r & 3u;
        // the ticket of the next round, one walk ahead of its use (its latency hides behind this walk)
        if ((uint32_t)wave == ((r + 1u) & 3u) && lane == 0) {
            v_tile[(r + 1u) & 3u] = M ? (r + 1u < M ? t + 1u : 0xffffffffu) : atomicAdd(p.out.ticket, 1u);
            v_tile_tag[(r + 1u) & 3u] = r + 2u;
        }
        const Tile tl = setup_tile(t);
        ctx.list = smem + q * kPipeRow + (uint32_t)tid;
        // the look-back of the previous tile runs in the middle of this walk, on wave r & 3
        const bool duty = r > 0 && (uint32_t)wave == gen;
        const HookFn hook{lookback, t_prev, r - 1u};
        ctx.hook_block = duty ? hook_b : 0u;
        if (duty && tl.partial) {
            // the last tile of a range may leave lanes of the duty wave without windows, and the
            // look-back needs the whole wave: run it before the walk (once per launch)
            hook();
            ctx.hook_block = 0u;
        }
        uint32_t my_count = 0;
        if (tl.lane_active) {
            bool over = false;
            my_count = tl.partial
                ? lane_walk<W, CANON, HASH_RC, MODE, false, false, true, false, 1, (int)kPipePitch, HookFn>(p, ctx, over, hook)
                : lane_walk<W, CANON, HASH_RC, MODE, false, false, false, false, 1, (int)kPipePitch, HookFn>(p, ctx, over, hook);
            if (over) s_overflow[gen] = r + 1u;  // benign race: every writer stores the same value
        }
        const uint32_t incl = wave_scan_dpp(my_count);
        const uint32_t wave_total = __builtin_amdgcn_readlane(incl, kWave - 1);
        if (lane == 0) {
            s_wave_tot[gen][wave] = wave_total;
            // LDS, in order behind the store above.  The wave that finishes the walk last publishes the
            // tile's count (count + 1: zero means "not yet").
            if (atomicAdd(&s_done[gen], 1u) == 4u * (r >> 2) + 3u) {
                uint32_t tot = 0;
#pragma unroll
                for (int v = 0; v < kFusedWaves; ++v) tot += v_wtot[gen * kFusedWaves + v];
                __hip_atomic_store(&p.pipe_counts[t], tot + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        const uint32_t pk_cur = ((incl - my_count) << 9) | (my_count & 511u);
        if (r > 0) copy_out(t_prev, r - 1u, pk_prev);
        pk_prev = pk_cur;
        t_prev = t;
    }
    // ---- drain: the last tile of this workgroup has nothing to hide behind
    if (r > 0) {
        if ((uint32_t)wave == (r & 3u)) lookback(t_prev, r - 1u);
        copy_out(t_prev, r - 1u, pk_prev);
    }
}

}  // namespace mm
