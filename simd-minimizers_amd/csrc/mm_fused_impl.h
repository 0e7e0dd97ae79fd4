// mm_fused_impl.h — the fused gfx950 minimizer kernel (one launch per sequence range).
//
// Formulation (MI355X-first, not a translation of the reference's 8-lane loop):
//
//   phase 1  every lane of a 256-lane workgroup walks S = nblk*W consecutive windows serially,
//            everything in registers.  The lane's 2-bit bases come straight from HBM/L2 through
//            bounds-checked raw buffer loads (two funnel-shifted 16-base views per W-block: base
//            entering the hash, base leaving the hash; the base leaving the strand window comes
//            from the previous block's hash-out view), issued two W-blocks ahead.  Rolling ntHash costs ONE LDS table look-up per base
//            (s_tab[(out<<2)|in], 16 x uint2 = forward / reverse-complement contribution);
//            keys are (hash_hi16 | pos16) for the leftmost minimum and the complemented key for
//            the rightmost one; the two-stacks sliding minimum runs over blocks of W with the
//            ring held in W registers (W is a template parameter; suffix minima kept at every second
//            element only, see ring_turn); the strand vote is taken only in the steps where the two
//            minima are different elements (w <= 36; the count of T|G bases is kept per block), per
//            window otherwise.  A window that emits (adjacent dedup against the lane's own previous
//            window, or the syncmer predicate) appends its 16-bit lane-relative position to the
//            lane's private list in LDS with one exec-masked ds_write.  Reference semantics:
//            src/sliding_min.rs:86-212, src/canonical.rs:12-31, src/minimizers.rs:117-128,
//            src/collect.rs:15-37, src/syncmers.rs:33-37.
//   phase 2  lane counts -> DPP prefix sum -> decoupled look-back across workgroups for the
//            global output offset; then each wave copies the 64 lists of its lanes, in lane
//            order, to the output with coalesced u32 stores.  Output order == window order.
//   redo     a lane list holds `list_cap` entries (1.3 x the expected number + 8; entries past it
//            fall outside the workgroup's LDS allocation and are dropped by the hardware); a tile
//            in which some list overflowed (low-complexity sequence) is walked a second time with
//            the now-known output offsets, storing straight to HBM.
//
// No MFMA: this is integer / byte work bounded by VALU issue and HBM, not GEMM-shaped.
#pragma once
#include "mm_common.h"

#ifndef MM_LB_SLEEP_SHORT
#define MM_LB_SLEEP_SHORT 120
#endif
// Poll interval of a tile that waits for its nearest missing predecessor (s_sleep argument, x 64 clocks).  Forward
// walks keep the longest sleep (MM_LB_SLEEP = 120, mm_common.h: seven workgroups per CU, many pollers - w = 7 runs 3 %
// slower at 32); canonical walks poll four times as often since round 3: with the status words 64 bytes apart and the
// wide sequence loads the polls no longer cost the walking waves anything measurable, and a waiting tile notices its
// predecessor 1-2 us earlier: k=21 w=11 on 3.1 Gbp 1.722 -> 1.681 ms (120 -> 32; 60, 16, 8, 4 and 1 all within 1 % of
// it), closed syncmers w = 17 -0.6 %, w = 51 -0.5 % (tools/gpu_jit_w.py, profiles/r03_lookback_poll.txt).
#ifndef MM_LB_SLEEP_CANON
#define MM_LB_SLEEP_CANON 32
#endif
#ifndef MM_STORE_MOD
#define MM_STORE_MOD "nt"  // cache policy of the copy-out stores (fast path); see MM_STORE_AUX
#endif
#ifndef MM_STORE_AUX
#define MM_STORE_AUX 2  // cache policy bits of the copy-out stores: slc (streaming), measured +1 %
#endif
#ifndef MM_PREFETCH_BLOCKS
#define MM_PREFETCH_BLOCKS 2
#endif
// Stage-incremental timing (tools/gpu_stages.py, mirrors bench/src/bin/paper.rs:231-300; WRONG RESULTS by design, only
// through MM_JIT_DEFS): -DMM_STAGE=n keeps the walk up to stage n and folds what it computed into a sink word so that
// nothing before it is dead code: 1 sequence loads + 2-bit decode (table addresses), 2 + table look-ups and hash
// roll, 3 + keys and the leftmost sliding minimum, 4 + rightmost minimum and strand vote (canonical walks).  The
// emit (5) and phase 2 (6) are timed with the product kernel and MM_DEBUG=3 / 0.
#ifdef MM_STAGE
#define MM_STAGE_GE(n) (MM_STAGE >= (n))
#else
#define MM_STAGE_GE(n) 1
#endif

// 8-byte words between the look-back status words of consecutive tiles.  One word per 64 bytes: tiles that
// finish at about the same time publish and poll neighbouring words, and with the words packed 8 bytes apart
// those accesses contend for the same memory sector - the look-back wait of the forward kernel was mostly
// that (round 2, 3.1 Gbp, stride 1 -> 8: forward k=21 w=11 1.293 -> 1.213 ms, canonical 1.805 -> 1.749;
// stride 2 already gives 1.232, 16 and 32 nothing more; tools/gpu_jit_w.py).  2.5 MB of status words for
// 3.1 Gbp, cleared per launch.
#ifndef MM_STATUS_STRIDE
#define MM_STATUS_STRIDE 8
#endif

namespace mm {

// MM_DEBUG switches (wrong results by design) exist only in kernels compiled with -DMM_EXPERIMENTS
#ifdef MM_EXPERIMENTS
#define MM_DBG(p) ((p).debug)
#else
#define MM_DBG(p) 0u
#endif

constexpr size_t kStatusStride = MM_STATUS_STRIDE;

// Byte distance between consecutive entries of one lane's list: 258 u16 slots per plane
// (256 lanes + pad) = 129 dwords, odd, so that both the per-lane appends of phase 1 and the
// per-entry reads of phase 2 spread over all LDS banks.
constexpr uint32_t kListStride = 2u * (kFusedThreads + 2u);
// Forward walks over small windows keep 8-bit list entries: their default lanes are short enough
// (S + W <= 255, enforced by the launcher) for an element index to fit a byte, and lists half the size
// let a CU hold seven workgroups instead of six - the forward walk needs its resident waves to cover
// the time its tiles spend in look-back and copy-out (round 2: 34 % of the run).  Rows are then
// kFusedThreads + 4 bytes apart (65 dwords, odd).  Canonical walks are bounded by registers, not LDS,
// and want longer lanes; reads-mode positions are read-local (up to the read length); super-k-mer
// entries pack two fields: all of those keep 16 bits.
#ifndef MM_NO_E8  // (experiment: -DMM_NO_E8 for the whole library = 16-bit lists everywhere, longer forward lanes)
constexpr bool kEntry8Rule(uint32_t w, bool canon, bool sk) { return !canon && !sk && w <= 13u; }
#else
constexpr bool kEntry8Rule(uint32_t, bool, bool) { return false; }
#endif
template <int W, bool CANON, bool SK, bool READS>
constexpr bool kEntry8 = !READS && kEntry8Rule((uint32_t)W, CANON, SK);
constexpr uint32_t kListStride8 = kFusedThreads + 4u;
constexpr uint32_t list_stride(bool e8) { return e8 ? kListStride8 : kListStride; }

struct FusedParams {
    SeqView seq;
    HashTables ht;
    uint32_t k;
    uint32_t nblk;       // W-blocks per lane; S = W * nblk windows per lane
    uint32_t win_begin;  // window range [win_begin, win_end)
    uint32_t win_end;
    uint32_t list_cap;   // entries per lane list
    uint32_t use_ticket; // 1: tile id from an atomic ticket (safe mode), 0: blockIdx.x
    uint32_t debug;      // timing experiments (MM_DEBUG env; read only by kernels compiled with -DMM_EXPERIMENTS,
                         // i.e. by the experiments library's run-time specialisation - the product's kernels ignore
                         // it): 1 no look-back, 2 no copy-out, 4 no phase 1, 8 copy-out without stores, 16 half-size
                         // lists, 32 test hook: tile 0 reports a look-back time-out
    uint32_t epoch;      // tag of this launch's look-back status words (kEpochShift, mm_common.h); 0: cleared words
    // Tapered tail (round 4; one sequence or window range, not batches / reads): the last tiles of a launch walk
    // ever shorter lanes.  Tiles [0, taper_first) have p.nblk blocks per lane; tile taper_first + j belongs to level
    // l = min(1 + j / taper_per_level, p.nblk - taper_min_nblk) and walks p.nblk - l blocks per lane; the tapered
    // tiles follow one another from window offset taper_start (relative to win_begin) on.  taper_first = 0xffffffff:
    // no taper.
    uint32_t taper_first;
    uint32_t taper_per_level;
    uint32_t taper_min_nblk;
    unsigned long long taper_start;
    uint32_t append;     // 1: *out.total holds the outputs before this launch (carry-in of tile 0), 0: starts at 0
    // Round 5, skip-ambiguous walks over large windows: bytes of the landing area of the walk's look-ahead loads
    // (kAmbiLand in lane_walk; 4 waves x land_wave_bytes(W)) at the FRONT of the dynamic LDS - the lists start behind it, so
    // that list entries past a list's capacity still fall beyond the workgroup's allocation and never into it; 0 = none.
    uint32_t land_bytes;
    // reads mode (READS kernels): one lane per read, reads at a fixed stride in the buffer
    uint32_t n_reads;
    uint32_t reads_per_lane;            // READS: consecutive reads one lane walks one after the other (1..4)
    uint32_t read_stride;               // bases between the starts of consecutive reads
    uint32_t read_len;                  // length of every read, or the maximum when read_lens != null
    const uint32_t *read_lens;          // optional per-read lengths (device)
    // (round 4) reads packed BACK TO BACK, read r = bases [read_starts[r], read_starts[r + 1]) of the buffer - what the
    // FASTQ / FASTA packers write; null = reads at the fixed stride above.  n_reads + 1 entries (device).
    const unsigned long long *read_starts;
    unsigned long long *read_offsets;   // [n_reads + 1] first output slot of every read (device)
    // (round 6) LANE TABLE: lane t of tile b walks lane_segs[256 b + t] (LaneSeg, mm_common.h) - reads of ANY lengths in one
    // launch at full lane occupancy: a long read takes consecutive lanes, a short one a lane of its own; the blocks a
    // wave walks are those of its longest lane.  seg_tile_origin[b] = the smallest `start` among tile b's lanes (the
    // origin of the tile's buffer descriptor).  The table is padded to the grid.  null = one of the read layouts above.
    const LaneSeg *lane_segs;
    const uint32_t *seg_tile_origin;
    // skip-ambiguous windows (PackedNSeq, src/minimizers.rs:169-214): bit g set = the window that
    // starts at base g (relative to the first base of the sequence / buffer span) is skipped
    const uint32_t *wamb;               // null for a plain PackedSeq
    uint32_t wamb_dwords;
    // Batch mode (sequence-mode kernels): many independent sequences in one launch.  Tile b belongs
    // to sequence batch_tile_seq[b]; the sequence's view, window count and first tile come from
    // batch_seqs[]; positions are sequence-local; batch_offsets[s] receives the first output slot
    // of sequence s (written by its first tile), batch_offsets[n] the total.  Null = one sequence.
    const BatchSeq *batch_seqs;
    const BatchTile *batch_tile_seq;
    unsigned long long *batch_offsets;
    uint32_t batch_n;
    // timing experiments (MM_TRACE): 4 timestamps per tile (start, phase 1 done, look-back done, end)
    unsigned long long *trace;
    // split path (walk_kernel + mm_split.hip): the walk dumps its lists and publishes the tile's count
    uint8_t *dump;                      // n_tiles x dump_stride bytes: 512 bytes of lane counts, then the list rows
    uint32_t dump_stride;
    unsigned long long *tile_status;    // one word per tile: kSplitValid | kSplitOverflow | rows << 32 | count
    const struct RedoEntry *redo_list;  // redo pass (tiles whose lists overflowed): null for the walk itself
    const uint32_t *redo_n;
    OutParams out;
};

// split path: status word of a tile and the redo list the expander writes
constexpr unsigned long long kSplitValid = 1ull << 63, kSplitOverflow = 1ull << 62;
constexpr uint32_t kSplitHeader = 2u * kFusedThreads;  // bytes: one u16 count per lane in front of the list rows
struct RedoEntry {
    uint32_t tile, pad;
    unsigned long long prefix;  // first output slot of the tile
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

// ---- two-stacks sliding minimum with a SPARSE suffix stack (round 3).
// After a block of W keys has been walked, ring[j] holds key j.  The block turn keeps suffix minima at every second
// element only - those with the parity of W - 1: ring[e] = min(key e .. key W-1) - the elements between stay raw
// keys: one v_min3 per TWO elements.  A window that needs the suffix from a raw element r takes
// min3(prefix, ring[r], ring[r + 1]), which costs what min(prefix, suffix) did; a window whose suffix starts at a
// kept element folds its own key in with the third operand instead (the prefix minimum is then brought up to date
// every second step, again with one v_min3).  Per block and side: (W-3)/2 + ~1.5 (W-1) three-operand minima
// instead of (W-1) + 1.5 (W-1) - 19 instead of 25 for W = 11.  min is associative and the keys carry their
// position, so every regrouping returns the same (leftmost / rightmost) element: src/sliding_min.rs:86-212.
template <int W, bool RIGHT>
__device__ __forceinline__ void ring_turn(uint32_t (&ring)[W]) {
#ifdef MM_DENSE_SUFFIX  // (A/B: the rounds-1/2 form, a suffix minimum at every element)
#pragma unroll
    for (int e = W - 2; e >= 0; --e) ring[e] = RIGHT ? max(ring[e], ring[e + 1]) : min(ring[e], ring[e + 1]);
    return;
#endif
#pragma unroll
    for (int e = W - 3; e >= 1; e -= 2)
        ring[e] = RIGHT ? max3u(ring[e], ring[e + 1], ring[e + 2]) : min3u(ring[e], ring[e + 1], ring[e + 2]);
}
// minimum of the whole turned ring (the window that ends with the block; used once per lane)
template <int W, bool RIGHT>
__device__ __forceinline__ uint32_t ring_all(const uint32_t (&ring)[W]) {
#ifdef MM_DENSE_SUFFIX
    return ring[0];
#endif
    if (W == 1) return ring[0];
    if (W == 2) return RIGHT ? max(ring[0], ring[1]) : min(ring[0], ring[1]);
    if (W & 1) return RIGHT ? max3u(ring[0], ring[1], ring[2]) : min3u(ring[0], ring[1], ring[2]);  // ring[2] is kept
    return RIGHT ? max(ring[0], ring[1]) : min(ring[0], ring[1]);                                   // ring[1] is kept
}
// step j of a block: key = the new key, pre = prefix minimum (brought up to date on every second step),
// ring = the turned previous block below j + 1 .. W - 1 and this block's keys below j.  Returns the window minimum.
// (J is a constant after the caller's unrolling)
template <int W, bool RIGHT>
__device__ __forceinline__ uint32_t ring_step(uint32_t (&ring)[W], uint32_t &pre, const uint32_t key, const int J) {
    auto m2 = [](uint32_t a, uint32_t b) { return RIGHT ? max(a, b) : min(a, b); };
    auto m3 = [](uint32_t a, uint32_t b, uint32_t c) { return RIGHT ? max3u(a, b, c) : min3u(a, b, c); };
    const bool kept_next = ((J ^ W) & 1) == 0;  // element J + 1 has the parity of W - 1: a kept suffix
    const int N1 = (J + 1 < W) ? J + 1 : 0, N2 = (J + 2 < W) ? J + 2 : 0, P1 = (J > 0) ? J - 1 : 0;
    uint32_t sel;
#ifdef MM_DENSE_SUFFIX
    if (J == 0) {
        pre = key;
        sel = (W > 1) ? m2(key, ring[N1]) : key;
    } else if (J & 1) {
        sel = (J + 1 < W) ? m3(pre, key, ring[N1]) : m2(pre, key);
    } else {
        pre = m3(pre, ring[P1], key);
        sel = (J + 1 < W) ? m2(pre, ring[N1]) : pre;
    }
    ring[J] = key;
    return sel;
#endif
    if (kept_next) {  // (never the last step: W - 1 has the other parity)
        sel = (J == 0) ? m2(key, ring[N1]) : m3(pre, key, ring[N1]);
    } else {
        pre = (J == 0) ? key : (J == 1) ? m2(ring[0], key) : m3(pre, ring[P1], key);  // ring[J-1] holds key J-1
        sel = (J + 1 < W) ? m3(pre, ring[N1], ring[N2]) : pre;                         // J + 1 raw, J + 2 kept
    }
    ring[J] = key;
    return sel;
}

// gfx950 issues v_bitop3_b32 / v_xor / v_add / shifts at full rate (2 clk per wave64) but
// v_and_or, v_cndmask, v_cmp, v_bfe, v_alignbit, v_min/v_max at half rate (tools/ubench), so the
// walk prefers the former.
__device__ __forceinline__ uint32_t and_or3(uint32_t a, uint32_t b, uint32_t c) {  // (a & b) | c
    uint32_t r;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xea" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}
__device__ __forceinline__ uint32_t select3(uint32_t m, uint32_t a, uint32_t b) {  // m ? a : b (bitwise)
    uint32_t r;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xca" : "=v"(r) : "v"(m), "v"(a), "v"(b));
    return r;
}

// Walks whose blocks are not all inside the range or may hold skipped windows (reads, the last tile of
// a range, skip-ambiguous runs) carry two bodies per block: the fast exec-masked emit for blocks that
// qualify, the flag path otherwise.  For larger W that costs registers and with them occupancy of the
// whole kernel, so only small window sizes get it (the others take the flag path throughout).
template <int W>
constexpr bool kTwoBodies = W <= 12;

// Super-k-mer runs keep ONE 16-bit list per lane as well: an entry packs the window index i (lane-
// relative) and the minimizer's offset inside that window, rel = element - i in [1, W], as
// (i << kSkShift<W>) + rel = element + i * (2^shift - 1); the launcher keeps S << shift <= 65536.
template <int W>
constexpr int kSkShift = W < 2 ? 1 : W < 4 ? 2 : W < 8 ? 3 : W < 16 ? 4 : W < 32 ? 5 : W < 64 ? 6 : W < 128 ? 7 : 8;

// Wide sequence loads (round 3).  The walk reads its bases through per-lane loads: lanes sit S / 4 bytes apart, so ONE
// wave-level load costs the CU's vector memory pipeline about one cycle per cache line it touches whatever its width
// (tools/ubench/lane_load_rate.hip: 29 clk for 64 lanes 60 bytes apart, 8 or 16 bytes each; 110 clk at the 370 bytes
// of k=31 w=51), and that pipeline - not the VALU - set the pace of the forward walk (decode alone 0.71 ms of a 0.95
// ms walk; with every lane of a wave reading the same line the full forward kernel ran 1.27 -> 1.04 ms, k=31 w=51
// 1.87 -> 1.68: tools/gpu_loadsame.py).  With 8-byte loads every W-block re-fetched mostly bytes it already had
// (W = 11 consumes 2.75 bytes per block and stream).  Now a lane loads kWideDwords<W> (4 or 5) DWORD-ALIGNED dwords
// once per GROUP of kWideGroup<W> blocks (loads at odd byte offsets split per dword and cost 2-4 x: same ubench) and
// shifts the buffer down by the byte part of its position with v_alignbyte when the group starts.  The views of the
// group's blocks then begin at bit s + 2 * W * k + 32 * g of the buffer with s = 2 * (pos & 3) <= 6, so the dword pair
// a view comes from is a compile-time function of (k, g) as long as (2 * W * k) % 32 <= 24 and the view ends inside
// the 32 * dwords - 24 bits that are valid for every lane - which bounds the group.  W = 11: one load per stream and
// 4 blocks instead of 4; W = 51: two loads per block and stream instead of four.
// 0 blocks = the window size keeps the per-block 8-byte loads (W >= 64, W a multiple of 16, unlucky phases).
#ifndef MM_WIDE_LOADS
#define MM_WIDE_LOADS 1
#endif
constexpr int wide_group_blocks_nd(int W, int nd) {
    if (!MM_WIDE_LOADS || W % 16 == 0) return 0;
    const int nsub = (W + 15) / 16, valid = 32 * nd - 24;
    int m = 0;
    for (; m < 8; ++m) {
        bool ok = true;
        for (int g = 0; g < nsub; ++g) {
            const int off = 2 * W * m + 32 * g, low = off % 32;
            // bases this view has to hold; the LAST view one more: the strand window's leaving base of the block's last step
            // is the hash-out view moved on by one base (next_base_view), i.e. base W of the block.  (Round 5: without it
            // w = 49 and w = 65 - one base in their last view, right at the edge of the valid bits - read a zero there for
            // lanes at the worst alignment, and a tie whose strand count was off by one took the wrong side:
            // tests/test_gpu_round5.py::test_skip_ambiguous_large_windows_landing found it, on the plain walk too.)
            const int bases = (W - 16 * g < 16 ? W - 16 * g : 16) + (g == nsub - 1 ? 1 : 0);
            if (low > 24 || off + 6 + 2 * bases > valid) ok = false;
        }
        if (!ok) break;
    }
    return m;
}
// loads per block and stream: 1 / blocks with four dwords, 2 / blocks with five (a 16-byte and a 4-byte load)
constexpr int wide_dwords(int W) {
    const int m4 = wide_group_blocks_nd(W, 4), m5 = wide_group_blocks_nd(W, 5);
    return (m4 > 0 && 2 * m4 >= m5) ? 4 : 5;  // (cost 1 / m4 against 2 / m5)
}
constexpr int wide_group_blocks(int W) { return wide_group_blocks_nd(W, wide_dwords(W)); }
template <int W>
constexpr int kWideGroup = wide_group_blocks(W);
template <int W>
constexpr int kWideDwords = wide_dwords(W);

template <int N>
struct IntTag {
    static constexpr int value = N;
};

template <bool B>
struct BoolTag {
    static constexpr bool value = B;
};

// What one lane needs to walk its windows.
struct LaneCtx {
    const uint2 *tab;        // LDS hash tables
    long long p0;            // first base of the tile's element 0, dword-array coordinates (>= -1)
    uint32_t lane_bases;     // first base of this lane's element 0, relative to p0
    uint32_t wbase;          // value of the lane's window 0 (absolute window index, or 0 for reads)
    bool no_prev;            // the lane's first window has no predecessor (always emits)
    int rem_valid;           // windows of this lane inside the range (PARTIAL walks only)
    int min_rem;             // the smallest rem_valid among the walking lanes of the wave (wave-uniform)
    uint8_t *list;           // LDS: this lane's list slot 0 (list mode)
    uint32_t list_bytes;     // list_cap * kListStride
    uint32_t list_used;      // entries already in the list (reads mode: earlier reads of the lane)
    unsigned long long dst;  // first output slot of this lane (DIRECT mode)
    uint32_t abase;          // bit of the lane's window 0 in FusedParams::wamb (AMBI walks)
    uint32_t nblk;           // W-blocks this lane walks
    const uint32_t *seq_d;   // the sequence the tile reads (p.seq, or the batch entry)
    uint32_t seq_dwords;
    uint32_t land;           // LDS byte address of this WAVE's landing area (kAmbiLand walks), 0 = none (wave-uniform)
};

// Skip-ambiguous walks over large windows (round 5).  The walk that knows skipped windows carries its
// ambiguity words and their look-ahead on top of a register budget that is already full above w = 32 (168 registers, three
// waves per SIMD), and the compiler made room by spilling exactly the LANDING registers of the look-ahead loads - the fifth
// dword of both sequence streams and the ambiguity dwords - which puts an s_waitcnt vmcnt(0) behind each load: every block
// waited for the memory latency it was meant to hide, and a dirty wave walked at half speed (k=31 w=51: 1.03 ms per Gbp
// against 0.55 plain).  Those loads now land in LDS (buffer_load ... lds: no register until the data is used): per wave and
// parity 64 x 16 bytes of ambiguity bits and 2 x 64 x 4 bytes of fifth dwords, two parities.  The launcher allocates the
// area in FRONT of the lists (FusedParams::land_bytes) for runs with ambiguity bits whose kernel uses it - in front: the
// lists rely on entries past their capacity falling beyond the workgroup's allocation (see "redo"); behind them the area
// took those entries, a dense tile's walk read its own overflow as ambiguity bits and counted two windows too few
// (k = 1, w = 55: every base its own k-mer, 106 emits into a list of 23).
//
// Late round 5, the 3-workgroup classes (w >= 38): one 16-byte load of window bits per lane and BLOCK touched every bit line
// (128 bytes = 1024 windows = 20 blocks of w = 51) twenty times, and with the lines of 24 576 resident lanes per XCD - two
// sequence streams and the bits - far beyond its 4 MB of L2 none of those touches hit: the dirty walk of k=31 w=51 fetched
// 5.1 GB per Gbp (plain: 1.05 GB; the bits themselves are 0.125 GB) and ran at the fabric's speed, 0.82 ms
// (profiles/r05_skip_dirty_walk.txt, section 7).  There the bits now arrive in CHUNKS of amb_row_dwords(W) dwords per lane -
// one dword load to LDS per row, row r of the chunk at r x 256 bytes + 4 x lane, so that a lane reads dword d of its chunk at
// d x 256 + 4 x lane whatever its alignment (conflict-free; two rows per ds_read2st64) - and a chunk serves
// (32 x rows - 31) / W blocks: six at w = 51 (11 rows), seven at w = 64 (16 rows).  One buffer: the next chunk is asked for when the last block of the current one
// has read its bits, and used a block later.
constexpr uint32_t kLandQ4Bytes = 64u * 4u;
// rows of a chunk: as many as fit beside the DEFAULT lanes' lists three times per CU (w <= 54: the CU's 160 KB are handed out in
// units of 1280 bytes, so a workgroup may take 53 760; 39 520 of lists + 336 of tables leave 13 904: eleven rows and the fifth
// dwords are 13 312), more where two workgroups share a CU
// (w <= 37, four workgroups per CU - 40 960 bytes each - and lanes bounded by geometry(): eight rows)
constexpr uint32_t amb_row_dwords(int W) { return W <= 37 ? 8u : (W <= 54 ? 11u : 16u); }
#ifndef MM_AMBI_LAND
#define MM_AMBI_LAND 1  // (0: A/B, the register look-ahead of rounds 2-4)
#endif
#ifndef MM_AMBI_ROWS_MINW
#define MM_AMBI_ROWS_MINW 36  // (w = 33 .. 35 measured slower with the chunks: 0.69 -> 0.73, 0.81 -> 0.99 ms per Gbp at the 128-register bound; 36, 37 faster: 0.74 -> 0.68, 0.81 -> 0.68.  97: A/B, one 16-byte load per block everywhere)
#endif
constexpr bool ambi_land_rule(int W) { return MM_AMBI_LAND && W >= 32 && W <= 96 && wide_group_blocks(W) != 0; }
constexpr bool ambi_rows_rule(int W) { return ambi_land_rule(W) && W >= MM_AMBI_ROWS_MINW; }
// a wave's slice: the window bits (two parities of 64 x 16 bytes, or ONE chunk of amb_row_dwords rows), then the fifth dwords (two parities, or one buffer)
constexpr uint32_t land_amb_bytes(int W) { return ambi_rows_rule(W) ? amb_row_dwords(W) * 256u : 2u * 64u * 16u; }  // 2048 / 2816 / 4096
constexpr uint32_t land_q4_bytes(int W) { return (ambi_rows_rule(W) ? 2u : 4u) * kLandQ4Bytes; }  // (one buffer beside the chunks)
constexpr uint32_t land_wave_bytes(int W) { return land_amb_bytes(W) + land_q4_bytes(W); }                         // 3072 / 3328 / 4608
constexpr uint32_t ambi_land_bytes(int W) { return ambi_land_rule(W) ? kFusedWaves * land_wave_bytes(W) : 0u; }  // 12288 / 13312 / 18432 per workgroup

// One lane walks its S windows.  List mode: appends emitted 16-bit values to the lane's LDS
// list (entries past the capacity are dropped but counted).  DIRECT mode: stores final values
// to HBM from ctx.dst on.  Returns the number of emitted windows.
template <int W, bool CANON, bool HASH_RC, int MODE, bool SK, bool DIRECT, bool PARTIAL, bool AMBI = false, bool E8 = false>
__device__ __forceinline__ uint32_t lane_walk(const FusedParams &p, const LaneCtx &ctx, bool &overflowed) {
    static_assert(!E8 || !SK, "8-bit list entries hold positions only");
    constexpr uint32_t kStride = list_stride(E8);  // bytes between consecutive entries of the lane's list
    constexpr int NSUB = (W + 15) / 16;  // 16-base view words per W-block
    const uint32_t nblk = ctx.nblk;

    // Element 0 of a lane is the k-mer one position before its first window (that window is the
    // dedup predecessor).  P0 = first base of the tile's element 0 in dword-array coordinates;
    // it is -1 only for the very first window of an unshifted buffer.
    const long long P0 = ctx.p0;
    const long long Q0 = P0 >> 4;
    // (tile-uniform; pinned to SGPRs so that the buffer descriptor below is scalar and the loads
    // need no per-lane descriptor loop)
    const long long Q0v = Q0 < 0 ? 0 : Q0;
    const long long Q0c =
        (long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)((unsigned long long)Q0v >> 32)) << 32) |
                    (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)Q0v));
    const int32_t prel0 = (int32_t)(P0 - (Q0c << 4));  // -1 .. 15
    // Bounds-checked view of the packed sequence from dword Q0c on: dwords past the end read as 0,
    // so the halo after the last base needs no clamping.
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(ctx.seq_d + Q0c), 0, (int)(((long long)ctx.seq_dwords - Q0c) * 4), 0x00020000);
    const int32_t pb = prel0 + (int32_t)ctx.lane_bases;  // first base of this lane's element 0
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

    // the same in two steps, for loads issued two W-blocks ahead of their use
    using RawView = decltype(__builtin_amdgcn_raw_buffer_load_b64(rsrc, 0, 0, 0));
    auto raw = [&](int32_t pos) -> RawView {
#ifdef MM_EXP_LOADSAME  // timing experiment (wrong results): every lane of the wave reads the same cache line
        return __builtin_amdgcn_raw_buffer_load_b64(rsrc, (((uint32_t)pos >> 4) << 2) & 0x78u, 0, 0);
#else
        return __builtin_amdgcn_raw_buffer_load_b64(rsrc, ((uint32_t)pos >> 4) << 2, 0, 0);
#endif
    };
    auto aligned = [&](const RawView &d, int32_t pos) -> uint32_t {
        return __builtin_amdgcn_alignbit(d[1], d[0], 2u * ((uint32_t)pos & 15u));
    };

    const uint32_t rot_l = (32u - p.ht.rot) & 31u;  // alignbit amount for rotl(x, rot)
    const uint32_t rot_r = p.ht.rot & 31u;
    const uint32_t k = p.k;
    uint32_t fw = p.ht.fw0, rc = p.ht.rc0;  // (the hasher's constant XOR terms; 0 for NtHasher)
    const uint8_t *tabb = reinterpret_cast<const uint8_t *>(ctx.tab);

    // hash of element 0: k add-only steps, 16 bases per view word, two bases per look-up
    // (s_tab[20 + ((second << 2) | first)]), a last odd base through the one-base table
    const uint32_t rot2_l = (32u - 2u * p.ht.rot) & 31u, rot2_r = (2u * p.ht.rot) & 31u;
#ifdef MM_EXP_NOWARM  // timing experiment (wrong results): what the hash warm-up of a lane costs
    for (uint32_t g = 0; g * 16u < 0u; ++g) {
#else
    for (uint32_t g = 0; g * 16u < k; ++g) {
#endif
        const uint32_t wa = g == 0 ? view_first(pb) : view(pb + 16 * (int32_t)g);
        const uint32_t rem = k - 16u * g;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if ((uint32_t)(2 * m + 1) < rem) {
                const uint32_t a8 = (m == 0 ? (wa << 3) : (wa >> (4 * m - 3))) & 0x78u;
                const uint2 t = *reinterpret_cast<const uint2 *>(tabb + 160 + a8);
                fw = __builtin_amdgcn_alignbit(fw, fw, rot2_l) ^ t.x;
                if (HASH_RC) rc = __builtin_amdgcn_alignbit(rc, rc, rot2_r) ^ t.y;
            } else if ((uint32_t)(2 * m) < rem) {
                const uint32_t a8 = (m == 0 ? (wa << 3) : (wa >> (4 * m - 3))) & 0x18u;
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
    // bit masks of the per-block word merges, in VGPRs for the same reason (v_bitop3 with three VGPR
    // sources issues at full rate; v_bfi and anything with an SGPR source at half rate)
    uint32_t m33, m55;
    asm volatile("v_mov_b32 %0, 0x33333333" : "=v"(m33));
    asm volatile("v_mov_b32 %0, 0x55555555" : "=v"(m55));

    uint32_t va[NSUB], vr[NSUB], v2[NSUB];  // views of the block being processed
    // The base leaving the strand window lags the base leaving the hash by exactly one W-block minus
    // one base, so its view of block b is the hash-out view of block b - 1 moved on by one base: no
    // third load stream unless W is a multiple of 16 (then the last base of the block is missing).
    constexpr bool kV2Load = CANON && (W % 16 == 0);
    auto next_base_view = [&](const uint32_t (&v)[NSUB], int g) -> uint32_t {
        return g + 1 < NSUB ? __builtin_amdgcn_alignbit(v[g + 1 < NSUB ? g + 1 : g], v[g], 2u) : (v[g] >> 2);
    };
    constexpr int PFD = MM_PREFETCH_BLOCKS;  // global loads run PFD W-blocks ahead of their use
    RawView qa[PFD - 1][NSUB], qr[PFD - 1][NSUB], q2[PFD - 1][NSUB];  // raw dwords of blocks b+2 .. b+PFD
    // wide loads (kWideGroup): ND dwords per lane, stream and group of MG blocks; Wa / Wr[0] serve the group whose
    // views are being made, Wa / Wr[1] are in flight (one group ahead)
    constexpr int MG = kWideGroup<W>;
    constexpr int ND = kWideDwords<W>;
    struct WideBuf {
        uint32_t q[5];
        uint32_t sh;  // 2 * (first base & 15): bit of the first base in q[0]
    };
#ifdef MM_EXP_ONE_STREAM  // timing experiment (wrong results): what a walk WITHOUT the hash-out load stream would cost
    WideBuf Wa[2];
    WideBuf(&Wr)[2] = Wa;
#else
    WideBuf Wa[2], Wr[2];  // [0] working buffer of the current group, [1] landing buffer of the next one
#endif
    constexpr bool kAmbiLand = AMBI && ambi_land_rule(W);  // (the look-ahead loads that land in LDS: see below)
    uint32_t gp_in = 0, gp_out = 0;  // first base (tile-relative) of the next group to load
    auto wide_load = [&](uint32_t gpos, WideBuf &w) {
        const uint32_t off = (gpos >> 4) << 2;
        w.sh = (gpos << 1) & 30u;
        typedef uint32_t u32x4w __attribute__((ext_vector_type(4)));
        const u32x4w v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        w.q[0] = v.x;
        w.q[1] = v.y;
        w.q[2] = v.z;
        w.q[3] = v.w;
        w.q[4] = (ND == 5 && !kAmbiLand) ? __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 16, 0) : 0u;
    };
    // kAmbiLand (see ambi_land_rule): the look-ahead loads whose landing registers the compiler used to spill - the fifth
    // dword of either stream, the ambiguity dwords - land in LDS instead.  `lpar` / `apar`: parity of the landing
    // buffer the NEXT group's fifth dwords / the NEXT block's ambiguity bits go to (wave-uniform).
    const uint32_t land = ctx.land;
    // (the lane's index is made again at every use - two instructions - instead of living in a register across the walk:
    // a register the allocator spills is reloaded with a scratch load, and a scratch load in the loop waits for every
    // look-ahead load in front of it)
    // (volatile: the compiler would otherwise hoist the loop-invariant address out of the block loop - and spill it)
    auto lane_id_now = [&]() -> uint32_t {
        uint32_t x;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(x));
        return x;
    };
    uint32_t lpar = 0, apar = 0;
    constexpr uint32_t kLandAmbBytes = land_amb_bytes(W);
    // (beside the chunked window bits the fifth dwords keep ONE buffer: a group's two values are read at the top of its first
    // block, the next group's loads are issued further down the same block behind a wait for those reads)
    constexpr bool kQ4One = AMBI && ambi_rows_rule(W);
    auto q4_off = [](uint32_t par, uint32_t which) -> uint32_t { return kLandAmbBytes + (2u * (kQ4One ? 0u : par) + which) * kLandQ4Bytes; };
    typedef __attribute__((address_space(3))) void *LdsPtr;
    auto lds_u32 = [&](uint32_t addr) -> uint32_t {
        return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>((uintptr_t)addr);
    };
    // which == 0: stream entering the hash, 1: leaving it
    auto land_q4_load = [&](uint32_t gpos, uint32_t par, uint32_t which) {
        const uint32_t off = (gpos >> 4) << 2;
        const uint32_t base = __builtin_amdgcn_readfirstlane(land + q4_off(par, which));
        // (the 16 bytes go into the SCALAR offset: an instruction offset of a load to LDS moves the LDS address as well)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LdsPtr)(uintptr_t)base, 4, off, 16, 0, 0);
    };
    auto land_q4_read = [&](uint32_t par, uint32_t which) -> uint32_t {
        return lds_u32(land + q4_off(par, which) + 4u * lane_id_now());
    };
    // the buffer that becomes current: shifted down by the byte part of the lane's position; returns the bit part
    auto wide_normalise = [&](WideBuf &dst, const WideBuf &w) -> uint32_t {
        const uint32_t a = w.sh >> 3;
#pragma unroll
        for (int i = 0; i < ND; ++i)
            dst.q[i] = __builtin_amdgcn_alignbyte(i + 1 < ND ? w.q[i + 1 < ND ? i + 1 : i] : 0u, w.q[i], a);
        return w.sh & 6u;
    };
    // views of block k of a group (k compile-time): dword pair and shift as derived above
    auto wide_views = [&](const WideBuf &w, uint32_t sh, auto ktag, uint32_t (&v)[NSUB]) {
        constexpr int K = decltype(ktag)::value;
#pragma unroll
        for (int g = 0; g < NSUB; ++g) {
            const int p = (2 * W * K + 32 * g) / 32;
            v[g] = __builtin_amdgcn_alignbit(p + 1 < ND ? w.q[p + 1 < ND ? p + 1 : 0] : 0u, w.q[p < ND ? p : 0], sh);
        }
    };
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
        // prefetch block 1 (its strand stream starts at pb + 1: block 0's hash-out view, one base on)
        if (CANON && !kV2Load) {
            uint32_t t2[NSUB];
#pragma unroll
            for (int g = 0; g < NSUB; ++g) t2[g] = next_base_view(vr, g);
#pragma unroll
            for (int g = 0; g < NSUB; ++g) v2[g] = t2[g];
        }
#pragma unroll
        for (int g = 0; g < NSUB; ++g) {
            va[g] = view(pos_in + 16 * g);
            vr[g] = view(pos_out + 16 * g);
            if (kV2Load) v2[g] = view(pb + 1 + 16 * g);
        }
        // ... and the raw dwords of block 2: global loads run two W-blocks ahead of their use, so
        // that a burst of copy-out stores of a neighbouring workgroup in the CU's memory pipeline
        // does not stall the walk
        if (MG == 0) {
#pragma unroll
            for (int d = 0; d < PFD - 1; ++d)
#pragma unroll
                for (int g = 0; g < NSUB; ++g) {
                    qa[d][g] = raw(pos_in + (d + 1) * W + 16 * g);
                    qr[d][g] = raw(pos_out + (d + 1) * W + 16 * g);
                    if (kV2Load) q2[d][g] = raw(pb + 1 + (d + 1) * W + 16 * g);
                }
        } else {
            // wide loads: group 0 = blocks 2 .. MG + 1 (the views of block 1 were loaded above)
            gp_in = (uint32_t)(pos_in + W);
            gp_out = (uint32_t)(pos_out + W);
            wide_load(gp_in, Wa[1]);
#ifndef MM_EXP_ONE_STREAM
            wide_load(gp_out, Wr[1]);
#endif
            if (kAmbiLand && ND == 5) {
                land_q4_load(gp_in, lpar, 0u);
                land_q4_load(gp_out, lpar, 1u);
            }
            gp_in += (uint32_t)(MG * W);
            gp_out += (uint32_t)(MG * W);
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
        ring_turn<W, false>(ring_l);
        if (CANON) ring_turn<W, true>(ring_r);
    }

    // strand vote: dn = #(T|G among the l bases of the window) - l/2 - 1, so the window is canonical iff dn >= 0.
    // Kept per BLOCK (the value for the block's first window; see "lazy strand vote" in the steps); with
    // -DMM_VOTE_EAGER (rounds 1-3, A/B) per window: each step adds tg(in) - tg(leaving base) from packed 2-bit signed
    // fields.
#ifndef MM_RANGE_FE
#define MM_RANGE_FE 1  // (0: A/B, partial walks over windows above 12 take the flag path)
#endif
    constexpr bool kRangeFE = MM_RANGE_FE && PARTIAL && !AMBI && MODE == 0 && !SK && !DIRECT && !kTwoBodies<W>;
#ifndef MM_AMBI_FE
#define MM_AMBI_FE 1  // (0: A/B, blocks with skipped windows take the flag path)
#endif
    constexpr bool kAmbiFE = MM_AMBI_FE && AMBI && MODE == 0 && !SK && !DIRECT;
#ifndef MM_VOTE_NO_DEFER
#define MM_VOTE_NO_DEFER 0  // (1: A/B, the branch of the lazy vote right behind its compare)
#endif
#ifdef MM_VOTE_EAGER
    constexpr bool kLazyVote = false;
#else
#ifndef MM_VOTE_LAZY_MAXW
#define MM_VOTE_LAZY_MAXW 35
#endif
    // (not for w = 36, 37: those kernels are bounded to 128 registers - four waves per SIMD - and the pending step's
    // two values push them further into scratch: w = 36 1.69 ms against 1.58, w = 37 2.27 against 1.62, where w = 35
    // still gains 6 %; from w = 38 on the bound is 168 registers)
    constexpr bool kLazyVote = CANON && (W <= MM_VOTE_LAZY_MAXW || W >= 38);
#endif
    int dn = 0;
    const uint32_t l = k + (uint32_t)W - 1;
    const int thr = (int)(l / 2);
    uint32_t prev;            // key of the predecessor window's k-mer (mode 0)
    int32_t pos_r2 = pb + 1;  // window 0 -> 1 drops base pb + 1
    if (CANON) {
        // window -1 covers bases [pb, pb + l)
        uint32_t c = 0;
#ifdef MM_EXP_NOWARM
        for (uint32_t g = 0; g * 16u < 0u; ++g) {
#else
        for (uint32_t g = 0; g * 16u < l; ++g) {
#endif
            uint32_t wd = (g == 0 ? view_first(pb) : view(pb + 16 * (int32_t)g)) & 0xAAAAAAAAu;
            const uint32_t rem = l - 16u * g;
            if (rem < 16u) wd &= (1u << (2u * rem)) - 1u;
            c += __popc(wd);
        }
        int cnt = (int)c;
        prev = (cnt > thr) ? ring_all<W, false>(ring_l) : ring_all<W, true>(ring_r);
        // move to window 0: + base pb + l, - base pb
        cnt += (int)((view(pb + (int32_t)l) >> 1) & 1u);
        cnt -= (int)((view_first(pb) >> 1) & 1u);
        dn = cnt - thr - 1;
    } else {
        prev = ring_all<W, false>(ring_l);
    }
    if (ctx.no_prev) prev = 0xffffffffu;  // no predecessor window: the first window always emits

    // skip-ambiguous windows: one bit per window, 32-window views prefetched one block ahead.
    // A skipped window emits nothing and (like the SIMD collector, src/intrinsics/dedup.rs:147-155)
    // never equals its successor, so the first clean window after it always emits.
    constexpr int NSUBA = AMBI ? (W + 31) / 32 : 1;
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(AMBI ? p.wamb : ctx.seq_d), 0, AMBI ? (int)(p.wamb_dwords * 4u) : 0, 0x00020000);
    auto aview = [&](uint32_t bit) -> uint32_t {
        const auto d = __builtin_amdgcn_raw_buffer_load_b64(arsrc, (bit >> 5) << 2, 0, 0);
        return __builtin_amdgcn_alignbit(d[1], d[0], bit & 31u);
    };
    uint32_t aw[NSUBA], aw_next[NSUBA];
    // kAmbiLand: the 128 bits from the dword that holds the block's first window bit on (31 + W <= 127) land in LDS
    auto land_amb_load = [&](uint32_t bit, uint32_t par) {
        const uint32_t base = __builtin_amdgcn_readfirstlane(land + par * 1024u);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (LdsPtr)(uintptr_t)base, 16, (bit >> 5) << 2, 0, 0, 0);
    };
    auto land_amb_read = [&](uint32_t bit, uint32_t par, uint32_t (&out)[NSUBA]) {
        typedef uint32_t u32x4a __attribute__((ext_vector_type(4)));
        const u32x4a d = *reinterpret_cast<const __attribute__((address_space(3))) u32x4a *>(
            (uintptr_t)(land + par * 1024u + 16u * lane_id_now()));
        const uint32_t dd[5] = {d.x, d.y, d.z, d.w, 0u};
#pragma unroll
        for (int g = 0; g < NSUBA; ++g) out[g] = __builtin_amdgcn_alignbit(dd[g + 1 < 5 ? g + 1 : 4], dd[g < 4 ? g : 3], bit & 31u);
    };
    // kAmbRows (ambi_rows_rule): chunks of kAmbRowDwords dwords per lane, row r of a chunk at r x 256 bytes + 4 x lane
    constexpr bool kAmbRows = kAmbiLand && ambi_rows_rule(W);
    constexpr uint32_t kAmbRowDwords = amb_row_dwords(W);
    constexpr uint32_t kAmbChunkBlocks = kAmbRows ? (32u * kAmbRowDwords - 31u) / (uint32_t)W : 1u;  // blocks a chunk serves
    static_assert(!kAmbRows || kAmbChunkBlocks >= 1u, "a chunk holds at least one block's bits at any alignment");
    uint32_t ka = 0;    // index of the coming block within its chunk (wave-uniform)
    uint32_t arel = 0;  // bit of the coming block's first window relative to the first dword of its chunk (per lane)
    auto land_rows_load = [&](uint32_t bit) {
        const uint32_t base = __builtin_amdgcn_readfirstlane(land);
        const uint32_t voff = (bit >> 5) << 2;
#pragma unroll
        for (uint32_t r = 0; r < kAmbRowDwords; ++r)  // (the row's 4 r bytes in the SCALAR offset, as in land_q4_load)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (LdsPtr)(uintptr_t)(base + r * 256u), 4, voff, r * 4u, 0, 0);
    };
    // (the last dword read, row (rel >> 5) + NSUBA, only contributes when rel & 31 != 0; at the chunk's last block and rel & 31 == 0
    // it lies one row behind the chunk - inside the wave's slice, the fifth dwords' buffer - and none of its bits is used)
    auto land_rows_read = [&](uint32_t rel, uint32_t (&out)[NSUBA]) {
        const uint32_t a = land + ((rel >> 5) << 8) + 4u * lane_id_now();
        uint32_t dd[NSUBA + 1];
#pragma unroll
        for (int g = 0; g <= NSUBA; ++g) dd[g] = lds_u32(a + 256u * (uint32_t)g);
#pragma unroll
        for (int g = 0; g < NSUBA; ++g) out[g] = __builtin_amdgcn_alignbit(dd[g + 1], dd[g], rel);  // (its low five bits)
    };
    if (AMBI && kAmbRows) {
        land_rows_load(ctx.abase);  // block 1's chunk
#pragma unroll
        for (int g = 0; g < NSUBA; ++g) aw_next[g] = 0u;
    } else if (AMBI && kAmbiLand) {
        land_amb_load(ctx.abase, apar);
#pragma unroll
        for (int g = 0; g < NSUBA; ++g) aw_next[g] = 0u;
    } else if (AMBI) {
#pragma unroll
        for (int g = 0; g < NSUBA; ++g) aw_next[g] = aview(ctx.abase + 32u * (uint32_t)g);
    }
    // (a lane whose predecessor window is skipped starts without a predecessor: its first clean window emits)
    if (AMBI && MODE == 0 && !ctx.no_prev && (aview(ctx.abase - 1u) & 1u)) prev = 0xffffffffu;

    // ---- blocks 1..nblk: one window per step
    // next free list slot as a 32-bit LDS address (the low half of the flat address of LDS memory)
    const uint32_t list0 = (uint32_t)reinterpret_cast<uintptr_t>(ctx.list);
    uint32_t lp32 = list0 + ctx.list_used * kStride;
    const uint32_t lp_end = list0 + ctx.list_bytes;
    uint32_t dropped = 0;              // entries that did not fit the list
    uint32_t sink = 0;                 // (MM_STAGE timing builds: keeps the stages alive)
    (void)sink;
    uint32_t valreg = 0;
    (void)valreg;
    uint32_t stride_v;  // list stride in a VGPR: v_add with two VGPR sources issues at full rate
    asm volatile("v_mov_b32 %0, %1" : "=v"(stride_v) : "s"(kStride));
    unsigned long long dst = ctx.dst;  // next output slot (DIRECT mode)
    // value of an emitted window: mode 0: (bw0 + lw - 1) + element index; syncmers: bw0 + lw + i
    const uint32_t wbase = ctx.wbase;
    const uint32_t vbase = wbase - (MODE == 0 ? 1u : 0u);
    // Running per-lane registers for the main loop, advanced by literal adds: an add whose other
    // operand is an SGPR issues at half rate on gfx950 (profiles/r01_valu_issue_rates.txt), and that
    // is the form the compiler picks for base + block * W.  sh_*: funnel-shift amount of the next
    // block's view (the hardware uses its low five bits); pl_*: base position of the loads that run
    // PFD - 1 blocks ahead.
    uint32_t sh_in = 2u * (uint32_t)pos_in, sh_out = 2u * (uint32_t)pos_out;
    uint32_t pl_in = (uint32_t)pos_in + (uint32_t)((PFD - 1) * W), pl_out = (uint32_t)pos_out + (uint32_t)((PFD - 1) * W);
#define MM_BUMP(x, c) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x) : "n"(c))
    uint32_t kn = 0;  // wide loads: index of block b + 1 within its group (wave-uniform)
    // The W-block loop.  Round 5: with wide loads the index of a block within its load group (kn) decides which dword pair
    // its views come from - a wave-uniform switch per block, whose arms the compiler joined with register copies (7 v_mov
    // and a chain of s_cmp / s_cbranch per block in the w = 11 kernel).  The full-tile walk now runs whole GROUPS of MG
    // blocks with kn a compile-time constant and leaves the switch to the blocks behind the last whole group (tapered
    // tiles, lanes whose length is no multiple of MG) and to the walks that are not the hot one (partial tiles, direct
    // stores, skip-ambiguous).  -DMM_GROUP_UNROLL=0: the loop of rounds 3-4 (A/B).
#ifndef MM_GROUP_UNROLL
#define MM_GROUP_UNROLL 1
#endif
    // (round 6: also in the two-body partial walks - reads, lane-table segments: READS 8 M x 150 bp 0.680 -> 0.659 ms, LONGREADS
    // 1.549 -> 1.487, BATCH10K 0.142 -> 0.135, profiles/r06_group_unroll_partial.txt; -DMM_GROUP_UNROLL_PARTIAL=0 is the A/B)
#ifndef MM_GROUP_UNROLL_PARTIAL
#define MM_GROUP_UNROLL_PARTIAL 1
#endif
    constexpr bool kGroupUnroll = MM_GROUP_UNROLL && MG > 1 && !AMBI && !DIRECT &&
                                  (!PARTIAL || (MM_GROUP_UNROLL_PARTIAL && (kTwoBodies<W> || MM_GROUP_UNROLL_PARTIAL > 1) && MODE == 0 && !SK));
    auto block = [&](const uint32_t b, auto kn_tag) {
        constexpr int KN = decltype(kn_tag)::value;  // block b + 1's index within its load group, or -1: in `kn`
        uint32_t me[NSUB], mo[NSUB];
        uint32_t tgw[NSUB];           // eager strand vote: signed 2-bit steps of the count
        uint32_t xt[NSUB], yt[NSUB];  // lazy strand vote (below): T|G bits of the block's entering / leaving bases
#pragma unroll
        for (int g = 0; g < NSUB; ++g) {
#if !defined(MM_VGPR_MASKS) || MM_VGPR_MASKS
            me[g] = select3(m33, va[g], vr[g] << 2);
            mo[g] = select3(m33, va[g] >> 2, vr[g]);
#else
            me[g] = (va[g] & 0x33333333u) | ((vr[g] << 2) & 0xccccccccu);
            mo[g] = ((va[g] >> 2) & 0x33333333u) | (vr[g] & 0xccccccccu);
#endif
            if (CANON) {
                const uint32_t x = va[g], y = v2[g];
                if (!kLazyVote) {
                // 2-bit two's-complement fields: tg(in) - tg(leaving) in {-1,0,1}
                // (bit 2j+1 of a view word = T|G of base j: low bit of the field = in ^ out, high bit =
                // out & ~in; four instructions: xor, shift, and-not, bit-field insert)
#if !defined(MM_VGPR_MASKS) || MM_VGPR_MASKS
                tgw[g] = select3(m55, (x ^ y) >> 1, y & ~x);
#else
                tgw[g] = (((x ^ y) >> 1) & 0x55555555u) | ((y & ~x) & 0xAAAAAAAAu);
#endif
                } else {
                // bit 2j+1 of a view word = T|G of base j; only the block's own W bases count
                const int kCnt = (W - 16 * g) >= 16 ? 16 : (W - 16 * g);  // (constants after unrolling)
                const uint32_t kBlockMask = 0xAAAAAAAAu & (kCnt >= 16 ? 0xffffffffu : ((1u << (2 * (kCnt > 0 ? kCnt : 0))) - 1u));
                xt[g] = x & kBlockMask;
                yt[g] = y & kBlockMask;
                }
            }
        }
        pos_in += W;
        pos_out += W;
        pos_r2 += W;
        MM_BUMP(sh_in, 2 * W);
        MM_BUMP(sh_out, 2 * W);
        if (MG == 0) {
            MM_BUMP(pl_in, W);
            MM_BUMP(pl_out, W);
        }
        if (CANON && !kV2Load) {  // the next block's strand view, before vr moves on
            uint32_t t2[NSUB];
#pragma unroll
            for (int g = 0; g < NSUB; ++g) t2[g] = next_base_view(vr, g);
#pragma unroll
            for (int g = 0; g < NSUB; ++g) {
                v2[g] = t2[g];
                // (made here, not sunk into the next iteration: otherwise the old vr has to be carried
                // around the loop in a second register)
                asm volatile("" : "+v"(v2[g]));
            }
        }
        if (AMBI && kAmbiLand) {
            // This block's window bits come out of LDS (every load of the wave has to be back for that: the wait stands
            // BEFORE this block's look-ahead loads are issued, so that it only waits for loads a block old), the next
            // block's are sent there.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (kAmbRows) {
                // (wave-uniform) a new chunk starts with this block: asked for at the end of the block before it
                if (ka == 0u) arel = (ctx.abase + (b - 1u) * (uint32_t)W) & 31u;
                land_rows_read(arel, aw);
                arel += (uint32_t)W;
            } else {
                land_amb_read(ctx.abase + (b - 1u) * (uint32_t)W, apar, aw);
            }
            if (ND == 5 && (KN >= 0 ? (uint32_t)KN : kn) == 0u) {
                // (a new load group starts with this block: the fifth dwords of its two streams, under the same wait)
                Wa[1].q[4] = land_q4_read(lpar, 0u);
                Wr[1].q[4] = land_q4_read(lpar, 1u);
                lpar ^= 1u;
            }
            if (kAmbRows) {
                // (the chunk's last block has taken its bits: the next chunk goes to the SAME rows - once those reads are
                // back, the hardware does not order an LDS read against a later load to LDS - and is waited for at the top
                // of the next block, a block's time later, like every look-ahead load of this walk)
                if (ka + 1u == kAmbChunkBlocks) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    land_rows_load(ctx.abase + b * (uint32_t)W);
                    ka = 0u;
                } else {
                    ka += 1u;
                }
            } else {
                apar ^= 1u;
                land_amb_load(ctx.abase + b * (uint32_t)W, apar);
            }
        }
        if (MG != 0) {
            // wide loads: block b + 1 is block kn of its group; a new group takes the next buffer and starts the load
            // of the group after it (a harmless over-read after the last block)
            if ((KN >= 0 ? (uint32_t)KN : kn) == 0u) {
                // (the landing buffer [1] is shifted into the working buffer [0]; then it takes the next load)
                // (kAmbiLand: the fifth dwords were taken out of LDS at the top of the block)
                sh_in = wide_normalise(Wa[0], Wa[1]);
#ifdef MM_EXP_ONE_STREAM
                sh_out = sh_in;
                wide_load(gp_in, Wa[1]);
                MM_BUMP(gp_in, MG * W);
#else
                sh_out = wide_normalise(Wr[0], Wr[1]);
                wide_load(gp_in, Wa[1]);
                wide_load(gp_out, Wr[1]);
                if (kAmbiLand && ND == 5) {
                    if (kQ4One) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the reads at the top of this block)
                    land_q4_load(gp_in, lpar, 0u);
                    land_q4_load(gp_out, lpar, 1u);
                }
                MM_BUMP(gp_in, MG * W);
                MM_BUMP(gp_out, MG * W);
#endif
            }
            switch (KN >= 0 ? (uint32_t)KN : kn) {
#define MM_WIDE_CASE(K)                                                   \
    case K:                                                               \
        if (K < (MG ? MG : 1)) {                                          \
            wide_views(Wa[0], sh_in, IntTag<(K < (MG ? MG : 1) ? K : 0)>{}, va); \
            wide_views(Wr[0], sh_out, IntTag<(K < (MG ? MG : 1) ? K : 0)>{}, vr); \
        }                                                                 \
        break;
                MM_WIDE_CASE(0)
                MM_WIDE_CASE(1)
                MM_WIDE_CASE(2)
                MM_WIDE_CASE(3)
                MM_WIDE_CASE(4)
                MM_WIDE_CASE(5)
                MM_WIDE_CASE(6)
                MM_WIDE_CASE(7)
#undef MM_WIDE_CASE
                default: break;
            }
            if (KN < 0) kn = (kn + 1 == (uint32_t)(MG ? MG : 1)) ? 0u : kn + 1u;
        } else
        // views of the next block from the dwords loaded one block ago; issue the loads of the
        // block after it (a harmless over-read after the last block)
#pragma unroll
        for (int g = 0; g < NSUB; ++g) {
            va[g] = __builtin_amdgcn_alignbit(qa[0][g][1], qa[0][g][0], sh_in);
            vr[g] = __builtin_amdgcn_alignbit(qr[0][g][1], qr[0][g][0], sh_out);
            if (kV2Load) v2[g] = aligned(q2[0][g], pos_r2 + 16 * g);
#pragma unroll
            for (int d = 0; d + 1 < PFD - 1; ++d) {
                qa[d][g] = qa[d + 1][g];
                qr[d][g] = qr[d + 1][g];
                if (kV2Load) q2[d][g] = q2[d + 1][g];
            }
            qa[PFD - 2][g] = raw((int32_t)pl_in + 16 * g);
            qr[PFD - 2][g] = raw((int32_t)pl_out + 16 * g);
            if (kV2Load) q2[PFD - 2][g] = raw(pos_r2 + (PFD - 1) * W + 16 * g);
        }
        if (AMBI && !kAmbiLand) {
#pragma unroll
            for (int g = 0; g < NSUBA; ++g) {
                aw[g] = aw_next[g];
                aw_next[g] = aview(ctx.abase + b * (uint32_t)W + 32u * (uint32_t)g);
            }
        }
#ifdef MM_LIST_PARK
        if (!DIRECT) {
            // keep a whole block of appends inside the list: a lane that is about to run out of
            // slots is parked on its last W slots (its tile is then redone in DIRECT mode)
            if (lp32 + (uint32_t)W * kStride > lp_end) {
                const uint32_t park = lp_end - (uint32_t)W * kStride;
                dropped += (lp32 - park) / kStride;
                lp32 = park;
            }
        }
#endif
        // (default: no check inside the walk.  Entry c of a lane sits at c * kStride + 2 * lane, so
        // entries past the capacity lie past the end of the workgroup's LDS allocation - the lists are
        // the dynamic part, placed behind the static variables - where the hardware drops the writes;
        // the slot pointer keeps counting, the overflow shows at the end of the walk and the tile is
        // redone storing directly.)

        const uint32_t e0 = b * (uint32_t)W;  // element index of step j = 0
        uint32_t pl = 0, pr_ = 0;
        // Table look-ups do not depend on the hash state: issue them PF steps ahead so that the
        // LDS latency is off the serial fw/rc chain (the emit below ends a scheduling region at
        // every step, so the compiler cannot hoist them by itself).
#ifdef MM_PF
        constexpr int PF = W < MM_PF ? W : MM_PF;
#else
        // Canonical walks with w >= 19 look only one or two steps ahead: the registers a deeper look-ahead
        // costs are worth more as an extra wave per SIMD, which hides the LDS latency just as well
        // (round 2, 3.1 Gbp, against 12 steps ahead: w = 25 +4 %, 31 +5 %, and together with the tighter
        // register bounds below w = 33 +11 %, 37 +19 %, 51 +18 %; tools/gpu_jit_w.py).  Smaller windows
        // and forward walks are indifferent to it.
        // (canonical walks over small windows: 4 steps ahead since round 3 - the wide loads need the registers, and
        // the depth was measured indifferent there: w = 11 1.786 ms at 11 steps, 1.791 at 4)
        constexpr int PF = (CANON && MG != 0 && W < 19) ? 4 : (W < 12 ? W : (CANON && W >= 19 ? (W >= 48 ? 1 : 2) : 12));
#endif
        uint2 tq[W];
        auto lookup = [&](int j) -> uint2 {
            const int jj = j & 15, g = j >> 4, m = jj >> 1;
            const uint32_t mw = (jj & 1) ? mo[g] : me[g];
            const uint32_t a8 = (m == 0 ? (mw << 3) : (mw >> (4 * m - 3))) & 0x78u;
            if (!MM_STAGE_GE(2)) return uint2{a8, a8};
            return *reinterpret_cast<const uint2 *>(tabb + a8);
        };
#pragma unroll
        for (int j = 0; j < PF; ++j) tq[j] = lookup(j);
        // The W steps of the block.  FE (fast emit): the exec-masked append in inline assembly, legal
        // when every window of the block is inside the range and none is skipped.
        auto steps = [&](auto fe_tag) {
        constexpr int FEK = (int)decltype(fe_tag)::value;  // 0 flag path, 1 fast emit, 2 fast emit that knows skipped windows,
                                                           // 3 fast emit with the lane's range check (two-body walks, round 6)
        constexpr bool FE = FEK != 0;
        // the emit of step jj (window i = e0 + jj - W, which starts at element i + 1) with its decided minimum
        auto emit_step = [&](const int jj, uint32_t sel, const unsigned long long valid) {  // valid: kRangeFE walks only
            const uint32_t e = e0 + (uint32_t)jj;   // uniform
            const uint32_t i = e - (uint32_t)W;     // uniform
            if (FE) {
                // Common path: compare, and under the resulting exec mask append the 16-bit value
                // to the lane's list and advance its slot pointer (2 VALU + 2 SALU + 1 LDS).
                unsigned long long sv;
                if (FEK == 2) {
                    // Skip-ambiguous walks, blocks in which some lane has a skipped window (all blocks for windows
                    // above 12, which carry one body): the same append with the skip in it instead of the flag path
                    // (k=31 w=51 on 1 Gbp with Ns: 1.37 -> ms).  A skipped window turns its value into all ones - it then
                    // never equals a real successor and never a predecessor but another skipped one, like the SIMD
                    // collector's lanes (src/intrinsics/dedup.rs:147-155) - and its lane's slot pointer stands still, so
                    // what the append writes for it is overwritten by the lane's next entry (or falls behind the list's
                    // end).  `valid`: the lane's range check of partial walks (all ones otherwise).
                    // (round 5: in inline assembly - the compiler turned `stride & ~sext(bit)` into a shift to the sign bit, a
                    // compare and a v_cndmask on VCC, 19 issue cycles per window where the three instructions below take 8;
                    // the dirty walk of k=31 w=33 ran 2.0 x the plain one's time, profiles/r05_skip_dirty_walk.txt)
                    // (the bit-field extract stays a builtin - its offset is a constant only after unrolling, which an "n"
                    // constraint does not see under hiprtc - but its result is consumed by assembly, so it is made as such)
                    uint32_t selx, st;
                    const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)aw[AMBI ? (jj >> 5) : 0], jj & 31, 1);  // all ones: skipped
                    asm("v_or_b32 %0, %1, %2" : "=v"(selx) : "v"(sel), "v"(m));
                    asm("v_bitop3_b32 %0, %1, %2, %2 bitop3:0x30" : "=v"(st) : "v"(stride_v), "v"(m));      // stride & ~m
#define MM_EMIT_SKIP(WR, SELB)                                                                        \
    asm volatile("v_cmp_ne_u32_sdwa vcc, %[sel], %[prev] src0_sel:" SELB " src1_sel:" SELB "\n\t"    \
                 "s_and_b64 vcc, vcc, %[ok]\n\t"                                                     \
                 "s_and_saveexec_b64 %[sv], vcc\n\t" WR " %[lp], %[sel]\n\t"                        \
                 "v_add_u32 %[lp], %[st], %[lp]\n\t"                                                 \
                 "s_mov_b64 exec, %[sv]"                                                              \
                 : [lp] "+v"(lp32), [sv] "=&s"(sv)                                                    \
                 : [sel] "v"(selx), [prev] "v"(prev), [st] "v"(st), [ok] "s"(valid)                   \
                 : "vcc", "scc", "memory")
                    if (E8) MM_EMIT_SKIP("ds_write_b8", "BYTE_0");
                    else MM_EMIT_SKIP("ds_write_b16", "WORD_0");
#undef MM_EMIT_SKIP
                    prev = selx;
                } else if (MODE == 0 && SK) {
                    // one more VALU under the mask: the packed (window, offset) entry
                    const uint32_t skc = i * ((1u << kSkShift<W>) - 1u);  // uniform
                    asm volatile(
                        "v_cmp_ne_u32_sdwa vcc, %[sel], %[prev] src0_sel:WORD_0 src1_sel:WORD_0\n\t"
                        "s_and_saveexec_b64 %[sv], vcc\n\t"
                        "v_add_u32 %[val], %[c], %[sel]\n\t"
                        "ds_write_b16 %[lp], %[val]\n\t"
                        "v_add_u32 %[lp], %[st], %[lp]\n\t"
                        "s_mov_b64 exec, %[sv]"
                        : [lp] "+v"(lp32), [sv] "=&s"(sv), [val] "=&v"(valreg)
                        : [sel] "v"(sel), [prev] "v"(prev), [st] "v"(stride_v), [c] "s"(skc)
                        : "vcc", "scc", "memory");  // s_and_saveexec writes SCC
                    prev = sel;
#ifndef MM_EMIT_SAVEEXEC
                } else if (MODE == 0 && !PARTIAL && !AMBI && E8) {
                    asm volatile(
                        "v_cmpx_ne_u32_sdwa vcc, %[sel], %[prev] src0_sel:BYTE_0 src1_sel:BYTE_0\n\t"
                        "ds_write_b8 %[lp], %[sel]\n\t"
                        "v_add_u32 %[lp], %[st], %[lp]\n\t"
                        "s_mov_b64 exec, -1"
                        : [lp] "+v"(lp32)
                        : [sel] "v"(sel), [prev] "v"(prev), [st] "v"(stride_v)
                        : "vcc", "memory");
                    prev = sel;
                } else if (MODE == 0 && !PARTIAL && !AMBI) {
                    // every lane of the wave walks (full tile): the compare writes EXEC itself and all-ones
                    // comes back afterwards - one scalar instruction and the VCC round trip less per step
                    // (round 2: +0.5..0.7 % on 3.1 Gbp; -DMM_EMIT_SAVEEXEC restores the saved-mask form)
                    asm volatile(
                        "v_cmpx_ne_u32_sdwa vcc, %[sel], %[prev] src0_sel:WORD_0 src1_sel:WORD_0\n\t"
                        "ds_write_b16 %[lp], %[sel]\n\t"
                        "v_add_u32 %[lp], %[st], %[lp]\n\t"
                        "s_mov_b64 exec, -1"
                        : [lp] "+v"(lp32)
                        : [sel] "v"(sel), [prev] "v"(prev), [st] "v"(stride_v)
                        : "vcc", "memory");
                    prev = sel;
#endif
                } else if (MODE == 0 && (kRangeFE || FEK == 3)) {
                    // PARTIAL walks over larger windows (the last tile of a sequence or range, reads): the same append
                    // with the lane's range check in it - `valid`: the lanes whose window i is inside, i < rem_valid - instead of the flag
                    // path for the whole tile, which ran 1.7 x as long as a full tile's walk and held up the look-back
                    // of every tile behind it (24 contigs in one launch: 1.82 -> ms, round 3).
#define MM_EMIT_RANGE(WR, SELB)                                                                                 \
    asm volatile("v_cmp_ne_u32_sdwa vcc, %[sel], %[prev] src0_sel:" SELB " src1_sel:" SELB "\n\t"              \
                 "s_and_b64 vcc, vcc, %[ok]\n\t"                                                               \
                 "s_and_saveexec_b64 %[sv], vcc\n\t" WR " %[lp], %[sel]\n\t"                                  \
                 "v_add_u32 %[lp], %[st], %[lp]\n\t"                                                           \
                 "s_mov_b64 exec, %[sv]"                                                                        \
                 : [lp] "+v"(lp32), [sv] "=&s"(sv)                                                              \
                 : [sel] "v"(sel), [prev] "v"(prev), [st] "v"(stride_v), [ok] "s"(valid)                        \
                 : "vcc", "scc", "memory")
                    if (E8) MM_EMIT_RANGE("ds_write_b8", "BYTE_0");
                    else MM_EMIT_RANGE("ds_write_b16", "WORD_0");
#undef MM_EMIT_RANGE
                    prev = sel;
                } else if (MODE == 0 && E8) {
                    // (8-bit entries: every element index of the lane is below 256, the low bytes decide)
                    asm volatile(
                        "v_cmp_ne_u32_sdwa vcc, %[sel], %[prev] src0_sel:BYTE_0 src1_sel:BYTE_0\n\t"
                        "s_and_saveexec_b64 %[sv], vcc\n\t"
                        "ds_write_b8 %[lp], %[sel]\n\t"
                        "v_add_u32 %[lp], %[st], %[lp]\n\t"
                        "s_mov_b64 exec, %[sv]"
                        : [lp] "+v"(lp32), [sv] "=&s"(sv)
                        : [sel] "v"(sel), [prev] "v"(prev), [st] "v"(stride_v)
                        : "vcc", "scc", "memory");
                    prev = sel;
                } else if (MODE == 0) {
                    asm volatile(
                        "v_cmp_ne_u32_sdwa vcc, %[sel], %[prev] src0_sel:WORD_0 src1_sel:WORD_0\n\t"
                        "s_and_saveexec_b64 %[sv], vcc\n\t"
                        "ds_write_b16 %[lp], %[sel]\n\t"
                        "v_add_u32 %[lp], %[st], %[lp]\n\t"
                        "s_mov_b64 exec, %[sv]"
                        : [lp] "+v"(lp32), [sv] "=&s"(sv)
                        : [sel] "v"(sel), [prev] "v"(prev), [st] "v"(stride_v)
                        : "vcc", "scc", "memory");  // s_and_saveexec writes SCC
                    prev = sel;
                } else if (MODE == 1) {
                    unsigned long long t2;
                    const uint32_t first = i + 1u;
#define MM_EMIT_CLOSED(WR)                                                                              \
    asm volatile("v_cmp_eq_u16 vcc, %[a], %[sel]\n\t"                                                   \
                 "v_cmp_eq_u16 %[t2], %[sel], %[b]\n\t"                                                 \
                 "s_or_b64 vcc, vcc, %[t2]\n\t"                                                         \
                 "v_mov_b32 %[val], %[iv]\n\t"                                                          \
                 "s_and_saveexec_b64 %[sv], vcc\n\t" WR " %[lp], %[val]\n\t"                            \
                 "v_add_u32 %[lp], %[st], %[lp]\n\t"                                                    \
                 "s_mov_b64 exec, %[sv]"                                                                \
                 : [lp] "+v"(lp32), [sv] "=&s"(sv), [t2] "=&s"(t2), [val] "=&v"(valreg)                 \
                 : [sel] "v"(sel), [a] "s"(first), [b] "s"(e), [iv] "s"(i), [st] "v"(stride_v)          \
                 : "vcc", "scc", "memory") /* s_and_saveexec writes SCC */
                    if (E8) MM_EMIT_CLOSED("ds_write_b8");
                    else MM_EMIT_CLOSED("ds_write_b16");
#undef MM_EMIT_CLOSED
                } else {
                    const uint32_t mid = i + 1u + (uint32_t)(W / 2);
#define MM_EMIT_OPEN(WR)                                                                   \
    asm volatile("v_cmp_eq_u16 vcc, %[a], %[sel]\n\t"                                      \
                 "v_mov_b32 %[val], %[iv]\n\t"                                             \
                 "s_and_saveexec_b64 %[sv], vcc\n\t" WR " %[lp], %[val]\n\t"               \
                 "v_add_u32 %[lp], %[st], %[lp]\n\t"                                       \
                 "s_mov_b64 exec, %[sv]"                                                   \
                 : [lp] "+v"(lp32), [sv] "=&s"(sv), [val] "=&v"(valreg)                    \
                 : [sel] "v"(sel), [a] "s"(mid), [iv] "s"(i), [st] "v"(stride_v)           \
                 : "vcc", "scc", "memory") /* s_and_saveexec writes SCC */
                    if (E8) MM_EMIT_OPEN("ds_write_b8");
                    else MM_EMIT_OPEN("ds_write_b16");
#undef MM_EMIT_OPEN
                }
            } else {
                bool flag;
                const bool skipped = AMBI && ((aw[AMBI ? (jj >> 5) : 0] >> (jj & 31)) & 1u);
                if (MODE == 0) {
                    flag = (uint16_t)sel != (uint16_t)prev;
                    prev = skipped ? 0xffffffffu : sel;
                } else if (MODE == 1) {
                    flag = ((uint16_t)sel == (uint16_t)(i + 1u)) | ((uint16_t)sel == (uint16_t)e);
                } else {
                    flag = (uint16_t)sel == (uint16_t)(i + 1u + (uint32_t)(W / 2));
                }
                if (PARTIAL) flag = flag && ((int)i < ctx.rem_valid);
                if (AMBI) flag = flag && !skipped;
                if (flag) {
                    if (DIRECT) {
                        if (dst < p.out.cap) {
                            p.out.pos[dst] = (MODE == 0) ? vbase + (sel & 0xffffu) : vbase + i;
                            if (SK) p.out.sk[dst] = wbase + i;
                        }
                        ++dst;
                    } else {
                        uint8_t *lp = ctx.list + (lp32 - list0);
                        if (E8)
                            *lp = (uint8_t)(MODE == 0 ? sel : i);
                        else
                            *reinterpret_cast<uint16_t *>(lp) =
                                (uint16_t)(MODE == 0 ? (SK ? sel + i * ((1u << kSkShift<W>) - 1u) : sel) : i);
                        lp32 += kStride;
                    }
                }
            }

        };
        // LAZY STRAND VOTE (round 3).  The vote only matters where the leftmost and the rightmost minimum are different
        // elements - a tie of the 16 hash bits at the window's minimum, about one window in 7 000 - so the walk no longer
        // keeps the count of T|G bases per window (bit-field extract, add, sign, select: 10 issue cycles per window).  It
        // compares the two positions (one half-rate instruction into an SGPR pair), and the one wave-step in a hundred in
        // which some lane's differ rebuilds that window's count from the count at the block's start (dn) and the T|G bits
        // of the block's entering / leaving bases below the step.  The compare's branch is taken ONE STEP LATER, behind
        // the next window's minima (the step's emit waits with it): a branch right behind its compare stalled the wave for
        // the compare's way to the scalar unit, 7 % of the kernel (the kernel without the branch, wrong results: 1.563 ->
        // 1.451 ms).
        auto decide = [&](const int jj, const uint32_t sel, const uint32_t selr, unsigned long long differ) -> uint32_t {
            // (Lanes past the end of their range read zeros - poly-A, every hash equal, the two minima different at every
            // step: without this test the last tile of a sequence walked 1.4 x as long as a full one and held up the
            // look-back of everything behind it, 24 contigs in one launch 1.82 ms against 1.65 for whole tiles.)
            if (PARTIAL && !(kTwoBodies<W> && FEK == 1)) {
                differ &= __ballot((int)(e0 + (uint32_t)jj - (uint32_t)W) < ctx.rem_valid);
                if (differ == 0ull) return sel;
            }
            int d = dn;
#pragma unroll
            for (int g = 0; g <= (jj >> 4); ++g) {
                const int nb = jj - 16 * g;  // bases of group g below step jj
                const uint32_t m = nb >= 16 ? 0xAAAAAAAAu : (0xAAAAAAAAu & ((1u << (2 * (nb > 0 ? nb : 0))) - 1u));
                if (m) d += (int)__builtin_popcount(xt[g] & m) - (int)__builtin_popcount(yt[g] & m);
            }
            return d < 0 ? selr : sel;  // rightmost on the reverse strand
        };
        uint32_t p_sel = 0, p_selr = 0;      // the step whose branch and emit are pending
        unsigned long long p_differ = 0ull, p_valid = ~0ull;
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const uint32_t e = e0 + (uint32_t)j;  // uniform
            if (j + PF < W) tq[j + PF] = lookup(j + PF);
            const uint32_t h = HASH_RC ? fw + rc : fw;
#ifdef MM_STAGE
            if (!MM_STAGE_GE(3)) {  // timing build: stages 1 / 2 end here
                const uint2 t0 = tq[j];
                if (MM_STAGE_GE(2)) {
                    fw = __builtin_amdgcn_alignbit(fw, fw, rot_l) ^ t0.x;
                    if (HASH_RC) rc = __builtin_amdgcn_alignbit(rc, rc, rot_r) ^ t0.y;
                    sink ^= h;
                } else {
                    sink += t0.x;
                }
                continue;
            }
#endif
            const uint32_t kl = and_or3(h, kmask, e);
            // window minimum (sparse-suffix two-stacks, see ring_step)
            uint32_t sel = ring_step<W, false>(ring_l, pl, kl, j);
            bool deferred = false;
            // (range-checked fast emit: the lanes whose window of this step is inside their range; lanes past it read
            // zeros - poly-A, every hash equal - and must not send the lazy vote down its slow path at every step)
            // (round 6: a two-body walk takes the plain fast emit only in blocks that lie inside EVERY walking lane's range -
            // `ok` below - so there is nothing to check per step there)
            constexpr bool kAllValid = PARTIAL && kTwoBodies<W> && FEK == 1;
            const unsigned long long valid =
                (!kAllValid && (kRangeFE || FEK == 3 || (PARTIAL && CANON && kLazyVote) || (PARTIAL && kAmbiFE)))
                    ? __ballot((int)(e - (uint32_t)W) < ctx.rem_valid)
                    : ~0ull;
            if (CANON && MM_STAGE_GE(4)) {
                const uint32_t selr = ring_step<W, true>(ring_r, pr_, kl ^ kmask, j);
                if (!kLazyVote) {
                    sel = select3((uint32_t)(dn >> 31), selr, sel);  // dn < 0: rightmost
                } else {
                    const unsigned long long differ = __builtin_amdgcn_uicmp(sel & 0xffffu, selr & 0xffffu, 33);  // (NE)
#ifdef MM_EXP_VOTE_NOBRANCH  // (timing experiment, wrong results: the compare without its branch)
                    asm volatile("" ::"s"(differ));
                    const unsigned long long differ_used = 0ull;
#else
                    const unsigned long long differ_used = differ & valid;
#endif
                    if (MM_STAGE_GE(5) && !(MM_VOTE_NO_DEFER)) {
                        if (j > 0) {  // the step before this one: its branch, then its emit
                            if (__builtin_expect(p_differ != 0ull, 0)) p_sel = decide(j - 1, p_sel, p_selr, p_differ);
                            emit_step(j - 1, p_sel, p_valid);
                        }
                        p_sel = sel;
                        p_selr = selr;
                        p_differ = differ_used;
                        p_valid = valid;
                        deferred = true;
                    } else if (__builtin_expect(differ_used != 0ull, 0)) {
                        sel = decide(j, sel, selr, differ_used);
                    }
                }
            }
#ifdef MM_STAGE
            if (!MM_STAGE_GE(5)) {  // timing build: stages 3 / 4 end here (no emit)
                sink ^= sel;
                const uint2 t0 = tq[j];
                fw = __builtin_amdgcn_alignbit(fw, fw, rot_l) ^ t0.x;
                if (HASH_RC) rc = __builtin_amdgcn_alignbit(rc, rc, rot_r) ^ t0.y;
                if (CANON && !kLazyVote && MM_STAGE_GE(4)) dn += __builtin_amdgcn_sbfe((int)tgw[j >> 4], 2 * (j & 15), 2);
                continue;
            }
#endif
            if (!deferred) emit_step(j, sel, valid);

            const uint2 t = tq[j];
            fw = __builtin_amdgcn_alignbit(fw, fw, rot_l) ^ t.x;
            if (HASH_RC) rc = __builtin_amdgcn_alignbit(rc, rc, rot_r) ^ t.y;
            if (CANON && !kLazyVote) dn += __builtin_amdgcn_sbfe((int)tgw[j >> 4], 2 * (j & 15), 2);
        }
        if (CANON && kLazyVote && MM_STAGE_GE(5) && !(MM_VOTE_NO_DEFER)) {  // the block's last step
            if (__builtin_expect(p_differ != 0ull, 0)) p_sel = decide(W - 1, p_sel, p_selr, p_differ);
            emit_step(W - 1, p_sel, p_valid);
        }
        };  // steps
        // The inline-assembly emit needs every window of the block inside the range and none skipped.
        // Full tiles of a plain sequence always qualify; walks that may not (range ends inside the
        // tile, reads, skipped windows) carry both bodies and choose per block with a wave-uniform
        // test - for small W only, see kTwoBodies.
        constexpr bool kCanFast = !DIRECT;
        if (kCanFast && !PARTIAL && !AMBI) {
            steps(IntTag<1>{});
        } else if (kRangeFE) {
            steps(IntTag<kRangeFE ? 1 : 0>{});
        } else if (kCanFast && kTwoBodies<W>) {
            bool ok = true;
            if (AMBI) {
                uint32_t any = 0;
#pragma unroll
                for (int g = 0; g < NSUBA; ++g) {
                    const int bits = W - 32 * g;
                    any |= bits >= 32 ? aw[g] : (aw[g] & ((1u << (bits > 0 ? bits : 0)) - 1u));
                }
                ok = __ballot(any != 0) == 0;
            }
            if (PARTIAL) ok = ok && (int)(b * (uint32_t)W) <= ctx.min_rem;  // windows < b * W all valid
            // (round 6: the blocks of a partial walk that are not inside every lane's range - the last block of a read, of
            // a lane-table segment, of a range - take the range-checked append instead of the flag path: MM_RANGE_BODY=0 is
            // the A/B)
#ifndef MM_RANGE_BODY
#define MM_RANGE_BODY 1
#endif
            constexpr int kNotOk = kAmbiFE ? 2 : ((MM_RANGE_BODY && PARTIAL && MODE == 0 && !SK && !DIRECT) ? 3 : 0);
            if (ok) steps(IntTag<(kCanFast && kTwoBodies<W>) ? 1 : 0>{});
            else steps(IntTag<kNotOk>{});  // (skipped windows and the range check in the append itself)
        } else {
            steps(IntTag<kAmbiFE ? 2 : 0>{});
        }
        if (MM_STAGE_GE(3)) {
            ring_turn<W, false>(ring_l);
            if (CANON && MM_STAGE_GE(4)) ring_turn<W, true>(ring_r);
            if (CANON && kLazyVote && MM_STAGE_GE(4)) {  // the count moves on by the whole block
#pragma unroll
                for (int g = 0; g < NSUB; ++g) dn += (int)__builtin_popcount(xt[g]) - (int)__builtin_popcount(yt[g]);
            }
        }
    };  // block
    {
        uint32_t b = 1;
        if (kGroupUnroll) {
            // (kn == 0 here: block 2 is the first block of load group 0, see "wide loads" in block 0)
            for (; b + (uint32_t)(MG ? MG : 1) - 1u <= nblk; b += (uint32_t)(MG ? MG : 1)) {
#define MM_GROUP_BLOCK(K) \
    if (K < MG) block(b + (uint32_t)K, IntTag<(K < MG ? K : 0)>{});
                MM_GROUP_BLOCK(0)
                MM_GROUP_BLOCK(1)
                MM_GROUP_BLOCK(2)
                MM_GROUP_BLOCK(3)
                MM_GROUP_BLOCK(4)
                MM_GROUP_BLOCK(5)
                MM_GROUP_BLOCK(6)
                MM_GROUP_BLOCK(7)
#undef MM_GROUP_BLOCK
            }
        }
        for (; b <= nblk; ++b) block(b, IntTag<-1>{});
    }
#ifdef MM_STAGE
    if (!MM_STAGE_GE(5)) {  // keep the sink alive: one list slot per lane
        if (!DIRECT) *reinterpret_cast<volatile uint8_t *>(ctx.list) = (uint8_t)(sink ^ (sink >> 8) ^ (sink >> 16) ^ (sink >> 24));
        overflowed = false;
        return 0;
    }
#endif
    overflowed = dropped != 0 || lp32 > lp_end;  // entries were dropped (or a parked list is out of order)
    if (DIRECT) return (uint32_t)(dst - ctx.dst);
    return (lp32 - list0) / kStride + dropped;
}

// Skip-ambiguous runs (round 4): does ANY lane of this wave have a skipped window in its range (or right before it)?
// A wave-uniform answer from the window-ambiguity bits themselves - every lane ORs the S / 32 + 2 dwords that cover
// its own windows, ends unmasked (a neighbour's bit only makes a clean wave look dirty).  A clean wave takes the
// PLAIN walk: the instantiation that knows skipped windows carries the ambiguity words and their look-ahead on top of
// a register budget that is already full for the larger windows, and ran at half speed whether or not anything was
// skipped (k=31 w=51 on 1 Gbp: 1.13 ms against 0.59 plain).  A genome's Ns sit in a few gaps, so
// nearly all waves are clean; the walks agree on clean ranges by construction.
__device__ __forceinline__ bool wave_has_skipped(const FusedParams &p, const LaneCtx &ctx, uint32_t S, bool lane_active) {
    // the wave's lanes walk consecutive ranges: one contiguous stretch of bits, read with coalesced 16-byte loads
    // (a per-lane loop over each lane's own dwords cost 0.12 ms per Gbp at w = 51: 44 dependent strided loads a lane)
    const uint32_t nw = ctx.rem_valid < (int)S ? (uint32_t)(ctx.rem_valid > 0 ? ctx.rem_valid : 0) : S;
    uint32_t b0 = lane_active ? (ctx.abase ? ctx.abase - 1u : 0u) : 0xffffffffu;
    uint32_t b1 = lane_active ? ctx.abase + nw : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        b0 = min(b0, (uint32_t)__shfl_xor((int)b0, d, kWave));
        b1 = max(b1, (uint32_t)__shfl_xor((int)b1, d, kWave));
    }
    b0 = __builtin_amdgcn_readfirstlane(b0);
    b1 = __builtin_amdgcn_readfirstlane(b1);
    if (b1 <= b0) return false;  // (no lane of this wave walks)
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(p.wamb), 0,
                                                                       (int)(p.wamb_dwords * 4u), 0x00020000);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t q0 = (b0 >> 5) & ~3u, q1 = (b1 - 1u) >> 5;  // dwords [q0, q1], 16-byte groups
    const int lane = threadIdx.x & (kWave - 1);
    uint32_t any = 0;
    for (uint32_t q = q0 + 4u * (uint32_t)lane; q <= q1; q += 4u * kWave) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, q * 4u, 0, 0);  // (past the end: zeros)
        any |= v.x | v.y | v.z | v.w;
    }
    return __ballot(any != 0u) != 0ull;
}

// a no-progress round of lookback_overlapped sleeps up to 64 x 3.5 us: give up after about a second
constexpr uint32_t kMaxIdleRounds = 1u << 12;

__device__ __forceinline__ void publish_aggregate(unsigned long long *status, uint32_t bid, uint32_t total,
                                                  unsigned long long etag) {
    st_status(&status[(size_t)bid * kStatusStride], kFlagAgg | etag | (unsigned long long)total);
}

// Look-back of a tile run by wave 0 alone while the other waves of the workgroup may still be in
// phase 1 (every wave bumps *done when it finishes, its total is then in wave_tot[]; the last one
// publishes the tile's aggregate itself).  The wave scans the predecessors without blocking on the
// nearest missing one, consumes what is there (nearest first), and publishes the inclusive prefix
// when the prefix and the tile's own total are both known.
// Status words and bounds as in lookback_exclusive (mm_common.h), tagged with the launch's epoch (`etag`, see
// kEpochShift: a word of another launch reads as "not yet").  Returns the exclusive prefix.
template <int SLEEP>
__device__ __forceinline__ unsigned long long lookback_overlapped(unsigned long long *status, uint32_t bid,
                                                                  unsigned long long carry_in, uint32_t *error,
                                                                  uint32_t *done, const uint32_t *wave_tot,
                                                                  const unsigned long long etag) {
    const int lane = threadIdx.x & (kWave - 1);
    bool have_excl = (bid == 0);
    unsigned long long excl = (bid == 0) ? carry_in : 0ull;
    long long j = (long long)bid - 1;
    unsigned long long block_total = 0;
    uint32_t idle = 0;
    while (true) {
        bool progress = false;
        if (!have_excl) {
            const long long idx = j - lane;
            const unsigned long long s = idx >= 0 ? ld_status(&status[(size_t)idx * kStatusStride]) : (kFlagIncl | etag);
            const uint32_t fl = status_flag(s, etag);
            const unsigned long long zmask = __ballot(fl == 0);
            const unsigned long long pmask = __ballot(fl == 2);
            const int first_zero = zmask ? __builtin_ctzll(zmask) : kWave;
            const int first_p = pmask ? __builtin_ctzll(pmask) : kWave;
            const int take = first_p < first_zero ? first_p + 1 : first_zero;  // lanes [0, take) count
            unsigned long long v = lane < take ? (s & kEpochValMask) : 0ull;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, kWave);
            excl += v;
            if (first_p < first_zero) have_excl = true;
            else j -= take;
            progress = take > 0;
        } else {
            // the prefix is known: wait for the other waves of this workgroup
            const uint32_t dn = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile uint32_t *>(done));
            if (dn == (uint32_t)kFusedWaves) {
#pragma unroll
                for (int v = 0; v < kFusedWaves; ++v) block_total += reinterpret_cast<const volatile uint32_t *>(wave_tot)[v];
                break;
            }
        }
        if (!progress) {
            // Nothing new.  Waiting for a wave of this workgroup costs nothing (LDS); waiting for a
            // predecessor polls ONE status word (the nearest missing one, status[j]) with one lane:
            // a 64-wide poll through the device-coherent path every few hundred clocks by a thousand
            // waiting tiles slows the whole chip down.
            if (++idle > kMaxIdleRounds) {
                flag_error(error, 1u);  // dispatch-order violation: the host redoes the launch in ticket mode
                if (have_excl) break;
                have_excl = true;
            }
            if (!have_excl) {
                unsigned long long s0 = 0;
                for (uint32_t spins = 0; spins < 64u; ++spins) {
                    __builtin_amdgcn_s_sleep(SLEEP);
#ifdef MM_LB_SLEEP2
                    __builtin_amdgcn_s_sleep(MM_LB_SLEEP2);  // experiment: poll even less often than the longest sleep
#endif
                    if (lane == 0) s0 = ld_status(&status[(size_t)j * kStatusStride]);
                    s0 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(s0 >> 32)) << 32) |
                         (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)s0);
                    if (status_flag(s0, etag) != 0) break;
                }
            } else {
                __builtin_amdgcn_s_sleep(MM_LB_SLEEP_SHORT);
            }
        }
    }
    if (lane == 0)
        st_status(&status[(size_t)bid * kStatusStride], kFlagIncl | etag | ((excl + block_total) & kEpochValMask));
    return excl;
}

// (A one-round-trip look-back - per-tile counts in chunks of 1024 + a per-chunk base, up to 1023 independent loads
// per tile instead of a chain of 64-wide hops - was built in round 2 and measured slower twice, 1.88 / 1.41 ms against
// 1.85 / 1.35 and 1.723 against 1.698: both wait for the same event, and the chain waits with ONE lane polling ONE
// word.  Removed in round 4; DESIGN.md appendix A.)

// Phase 2, the copy-out of one wave: the 64 lists of its lanes (LDS, entry c of lane t at c * stride + eb * t)
// go to the output in lane order (= window order) from slot run0 on.  `vbt` = value of list entry 0 of the
// tile's lane 0 (first window of the tile, minus one for minimizer positions; 0 in reads mode), `S` = windows
// per lane, `kSh` = kSkShift of the window size (super-k-mer entries), `my_count` / `excl` = length of the
// lane's list and its first slot relative to run0, `wave_total` = sum of the wave's lengths.  Shared by the
// fused kernel and by the expander of the split path (mm_split.hip).
template <bool E8, bool SK, bool READS>
__device__ __forceinline__ void copy_out_wave(const uint8_t *smem, const OutParams &out, const uint32_t debug,
                                              const int wave, const int lane, const uint32_t vbt, const uint32_t S,
                                              const uint32_t kSh, const unsigned long long run0,
                                              const uint32_t wave_total, const uint32_t my_count,
                                              const uint32_t excl, const uint32_t lane_vb = 0u) {
    // lane_vb (READS): what the entries of THIS lane's list are relative to on top of vbt - the lane's first window
    // inside its read in lane-table launches (LaneSeg::win0), 0 otherwise
    constexpr uint32_t kStride = list_stride(E8), kEB = E8 ? 1u : 2u;
    {
    // Copy the 64 lists of this wave's lanes, in lane order (= window order).  Entry c of
    // lane t sits at smem + c * kListStride + 2 * t, so lane `c` of the copying wave reads
    // entry c of list L: conflict-free, and the stores of one list are contiguous.
    // Eight lists are in flight at a time (LDS reads first, then the stores); lanes past a
    // list's end get an out-of-range offset, which the bounds-checked store drops.
    if (!(debug & 2u)) {
        const uint32_t tid0 = (uint32_t)wave * kWave;
        const uint8_t *rd = smem + (uint32_t)lane * kStride + kEB * tid0;
        // entry u of the eight lists L0 .. L0 + 7 in flight (one byte or one 16-bit word each)
        auto entry = [&](int L) -> uint32_t {
            return E8 ? (uint32_t)rd[L] : (uint32_t)*reinterpret_cast<const uint16_t *>(rd + 2 * L);
        };
        const uint32_t vb0 = vbt + (READS ? 0u : tid0 * S);
        // Output window of this wave as a bounds-checked buffer (wave-uniform, so the
        // descriptor lives in SGPRs): stores past the caller's capacity are dropped by the
        // hardware, offsets stay 32-bit.
        const uint32_t r_lo = __builtin_amdgcn_readfirstlane((uint32_t)run0);
        const uint32_t r_hi = __builtin_amdgcn_readfirstlane((uint32_t)(run0 >> 32));
        const unsigned long long run0_u = ((unsigned long long)r_hi << 32) | r_lo;
        const unsigned long long room = out.cap > run0_u ? out.cap - run0_u : 0ull;
        // (64-bit compares run on the vector unit; pin the result to an SGPR so that the per-list
        // descriptor arithmetic below stays on the scalar unit)
        const uint32_t room32 = __builtin_amdgcn_readfirstlane(room > 0x3fffffffull ? 0x3fffffffu : (uint32_t)room);
        const uint32_t room_bytes = room32 * 4u;
        const __amdgpu_buffer_rsrc_t opos =
            __builtin_amdgcn_make_buffer_rsrc(out.pos + run0_u, 0, (int)room_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t osk = __builtin_amdgcn_make_buffer_rsrc(
            SK ? out.sk + run0_u : out.pos + run0_u, 0, (int)room_bytes, 0x00020000);
#ifndef MM_COPY_BATCH
#define MM_COPY_BATCH 8
#endif
        constexpr int kBatch = MM_COPY_BATCH;
        const uint32_t store_mask =
            __builtin_amdgcn_readfirstlane((debug & 8u) ? 0u : 0xffffffffu);  // timing experiment: no store
        // Fast path (the whole wave fits the caller's capacity, no SK): the lane mask "entry <
        // length of the list" and the list's byte offset are made on the scalar unit (s_bfm ->
        // exec, soffset), so a list costs 2 v_readlane + 1 v_add; of lists with more than 64 entries
        // the first 64 are stored here, the rest by the loop below.
#ifdef MM_NO_FAST
        const bool fast = false;
#else
        const bool fast = room32 >= wave_total;
#endif
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const unsigned long long obase = (unsigned long long)reinterpret_cast<uintptr_t>(out.pos + run0_u);
        u32x4 odesc;
        odesc.x = __builtin_amdgcn_readfirstlane((uint32_t)obase);
        odesc.y = __builtin_amdgcn_readfirstlane((uint32_t)(obase >> 32) & 0xffffu);
        odesc.z = 0x7fffffffu;  // the capacity was checked for the whole wave
        odesc.w = 0x00020000u;
        const unsigned long long sbase =
            (unsigned long long)reinterpret_cast<uintptr_t>((SK ? out.sk : out.pos) + run0_u);
        u32x4 sdesc = odesc;  // super-k-mer indices: same slots of the second output array
        sdesc.x = __builtin_amdgcn_readfirstlane((uint32_t)sbase);
        sdesc.y = __builtin_amdgcn_readfirstlane((uint32_t)(sbase >> 32) & 0xffffu);
        // an SK entry packs (window << shift) + offset-in-window (see kSkShift)
        const uint32_t kRelMask = (1u << kSh) - 1u;
        const uint32_t lane4 = (uint32_t)lane * 4u;
        // length and first slot of a lane's list in one word (one v_readlane per list instead of two;
        // a list that is copied holds at most 315 entries - 159 KB of LDS - and a wave's lists at most
        // 64 times that many)
        const uint32_t pk = (excl << 9) | (my_count & 511u);
        // (Tried in round 2: a lane-owned copy-out - every lane streams its own list with 16-byte stores,
        // a third of the instructions - ran the whole kernel 2.8 x slower, 5.2 ms: a wave's 64 scattered
        // segments per store defeat the memory pipeline.  The stores have to stay coalesced.)
        {
#ifndef MM_COPY_EXEC
        if (fast) {
            // The lanes of a list are selected by the bounds check of its store: a descriptor whose
            // num_records ends behind the list's last entry drops the lanes past it.  No EXEC write, so
            // consecutive lists do not serialise on the mask (a wave's time off the walk is what the
            // copy-out costs: the walk runs where VALU issue and per-wave latency both bind).
            const uint32_t *obase32 = out.pos + run0_u;
            const uint32_t *sbase32 = (SK ? out.sk : out.pos) + run0_u;
#pragma unroll
            for (int L0 = 0; L0 < kWave; L0 += kBatch) {
                uint32_t ent[kBatch];
#pragma unroll
                for (int u = 0; u < kBatch; ++u) ent[u] = entry(L0 + u);
#pragma unroll
                for (int u = 0; u < kBatch; ++u) {
                    const uint32_t pkl = __builtin_amdgcn_readlane(pk, L0 + u);
                    const uint32_t n = pkl & 511u, off = pkl >> 9;
                    const uint32_t vbl = READS ? vb0 + __builtin_amdgcn_readlane(lane_vb, L0 + u) : vb0 + (uint32_t)(L0 + u) * S;
                    const uint32_t iw = ent[u] >> kSh;
                    const uint32_t val = SK ? vbl + iw + (ent[u] & kRelMask) : vbl + ent[u];
                    // (round 2 experiment: storing every list as one or two full, aligned 128-byte lines
                    // instead changed the forward kernel by -1.5 % / +3 % - the stores cost by their bytes,
                    // not by their partial lines, so staging them into aligned runs would not pay)
#ifdef MM_EXP_DENSE_STORES
                    // Timing experiment (WRONG RESULTS by design): what a workgroup-dense copy-out
                    // could gain at best.  The same values and the same bytes leave through FULL, 256-byte-aligned
                    // 64-lane stores - row r of the wave's output region for the first ceil(total / 64) lists, nothing
                    // for the others - i.e. the store stream of a perfect dense copy-out without the LDS transposition
                    // it would need.  (The LDS reads and the value arithmetic of all 64 lists stay.)
                    {
                        (void)n;
                        const uint32_t rows = (wave_total + 63u) / 64u;
                        const unsigned long long base64 = run0_u & ~63ull;
                        const __amdgpu_buffer_rsrc_t dl = __builtin_amdgcn_make_buffer_rsrc(
                            const_cast<uint32_t *>(out.pos + base64), 0,
                            (int)(((uint32_t)(L0 + u) < rows && room32 >= wave_total + 128u ? rows * 256u : 0u) & store_mask), 0x00020000);
                        __builtin_amdgcn_raw_buffer_store_b32(val, dl, lane4, (uint32_t)(L0 + u) * 256u, MM_STORE_AUX);
                    }
                    const uint32_t end_bytes = 0u;  // (the super-k-mer store below: nothing in this experiment)
#else
                    const uint32_t end_bytes = ((off + n) * 4u) & store_mask;  // (store_mask 0: timing experiment)
                    const __amdgpu_buffer_rsrc_t dl = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<uint32_t *>(obase32), 0, (int)end_bytes, 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b32(val, dl, lane4, off * 4u, MM_STORE_AUX);
#endif
                    if (SK) {
                        const __amdgpu_buffer_rsrc_t dl2 = __builtin_amdgcn_make_buffer_rsrc(
                            const_cast<uint32_t *>(sbase32), 0, (int)end_bytes, 0x00020000);
                        __builtin_amdgcn_raw_buffer_store_b32(vbl + 1u + iw, dl2, lane4, off * 4u, MM_STORE_AUX);
                    }
                }
            }
        } else
#endif
        if (fast) {
#pragma unroll
            for (int L0 = 0; L0 < kWave; L0 += kBatch) {
                uint32_t ent[kBatch];
#pragma unroll
                for (int u = 0; u < kBatch; ++u) ent[u] = entry(L0 + u);
#pragma unroll
                for (int u = 0; u < kBatch; ++u) {
                    const uint32_t pkl = __builtin_amdgcn_readlane(pk, L0 + u);
                    const uint32_t n = pkl & 511u, off = pkl >> 9;
                    const uint32_t vbl = READS ? vb0 + __builtin_amdgcn_readlane(lane_vb, L0 + u) : vb0 + (uint32_t)(L0 + u) * S;
                    const uint32_t iw = ent[u] >> kSh;
                    const uint32_t val = SK ? vbl + iw + (ent[u] & kRelMask) : vbl + ent[u];
                    const uint32_t val2 = vbl + 1u + iw;
                    uint32_t t0;
                    unsigned long long sv;
                    if (SK)
                        asm volatile(
                            "s_and_b32 %[t0], %[n], %[sm]\n\t"
                            "s_mov_b64 %[sv], exec\n\t"
                            "s_bfm_b64 exec, %[t0], 0\n\t"
                            "s_cmp_lt_u32 %[t0], 64\n\t"
                            "s_cselect_b64 exec, exec, -1\n\t"
                            "s_lshl_b32 %[t0], %[off], 2\n\t"
                            "buffer_store_dword %[val], %[lane4], %[desc], %[t0] offen " MM_STORE_MOD "\n\t"
                            "buffer_store_dword %[val2], %[lane4], %[desc2], %[t0] offen " MM_STORE_MOD "\n\t"
                            "s_mov_b64 exec, %[sv]"
                            : [t0] "=&s"(t0), [sv] "=&s"(sv)
                            : [n] "s"(n), [off] "s"(off), [val] "v"(val), [val2] "v"(val2), [lane4] "v"(lane4),
                              [desc] "s"(odesc), [desc2] "s"(sdesc), [sm] "s"(store_mask)
                            : "scc", "memory");
                    else
                    asm volatile(
                        "s_and_b32 %[t0], %[n], %[sm]\n\t"
                        "s_mov_b64 %[sv], exec\n\t"
                        "s_bfm_b64 exec, %[t0], 0\n\t"
                        "s_cmp_lt_u32 %[t0], 64\n\t"
                        "s_cselect_b64 exec, exec, -1\n\t"
                        "s_lshl_b32 %[t0], %[off], 2\n\t"
                        "buffer_store_dword %[val], %[lane4], %[desc], %[t0] offen " MM_STORE_MOD "\n\t"
                        "s_mov_b64 exec, %[sv]"
                        : [t0] "=&s"(t0), [sv] "=&s"(sv)
                        : [n] "s"(n), [off] "s"(off), [val] "v"(val), [lane4] "v"(lane4), [desc] "s"(odesc),
                          [sm] "s"(store_mask)
                        : "scc", "memory");
                }
            }
        } else {
            // (capacity-checked offsets; also the super-k-mer flavour.  Kept rolled: eight lists
            // per iteration are enough to cover the LDS latency and the code stays small)
#pragma unroll 1
            for (int L0 = 0; L0 < kWave; L0 += kBatch) {
                uint32_t ent[kBatch];
#pragma unroll
                for (int u = 0; u < kBatch; ++u) ent[u] = entry(L0 + u);
#pragma unroll
                for (int u = 0; u < kBatch; ++u) {
                    const uint32_t pkl = __builtin_amdgcn_readlane(pk, L0 + u);
                    const uint32_t n = pkl & 511u, off = pkl >> 9;
                    const uint32_t vb = READS ? vb0 + __builtin_amdgcn_readlane(lane_vb, L0 + u) : vb0 + (uint32_t)(L0 + u) * S;
                    uint32_t voff = (uint32_t)lane < n ? (off + (uint32_t)lane) * 4u : 0xffffffffu;
                    voff |= ~store_mask;
                    const uint32_t iw = ent[u] >> kSh;
                    __builtin_amdgcn_raw_buffer_store_b32(SK ? vb + iw + (ent[u] & kRelMask) : vb + ent[u], opos,
                                                          voff, 0, MM_STORE_AUX);
                    if (SK) __builtin_amdgcn_raw_buffer_store_b32(vb + 1u + iw, osk, voff, 0, MM_STORE_AUX);
                }
            }
        }
        // lists longer than one wave: the entries from the 65th on, list by list (both paths above
        // have stored the first 64)
        for (unsigned long long longer = __ballot(my_count > (uint32_t)kWave); longer; longer &= longer - 1ull) {
            {
                const uint32_t L = (uint32_t)__builtin_ctzll(longer);
                const uint32_t pkl = __builtin_amdgcn_readlane(pk, L);
                const uint32_t n = pkl & 511u, off = pkl >> 9;
                const uint32_t vb = READS ? vb0 + __builtin_amdgcn_readlane(lane_vb, L) : vb0 + L * S;
                for (uint32_t c = (uint32_t)kWave + lane; c < n; c += kWave) {
                    const uint8_t *q = rd + kEB * L + (c - lane) * kStride;
                    const uint32_t e1 = E8 ? (uint32_t)*q : (uint32_t)*reinterpret_cast<const uint16_t *>(q);
                    const uint32_t iw = e1 >> kSh;
                    __builtin_amdgcn_raw_buffer_store_b32(SK ? vb + iw + (e1 & kRelMask) : vb + e1, opos,
                                                          (off + c) * 4u, 0, 0);
                    if (SK) __builtin_amdgcn_raw_buffer_store_b32(vb + 1u + iw, osk, (off + c) * 4u, 0, 0);
                }
            }
        }
        }
    }
    }
}

// READS = false: one sequence (range of windows), lane t walks windows [t*S, (t+1)*S) of the tile.
// READS = true : a batch of short reads at a fixed stride, lane t walks read (tile*256 + t) alone;
//                positions are read-local and read_offsets[] delimits the reads in the output.
template <int W, bool CANON, bool HASH_RC, int MODE, bool SK, bool READS>
// (small W: at least 4 waves per SIMD, i.e. at most 128 VGPRs - the two-body walks sit right at that
// limit; larger W need more registers and get no such bound)
// Workgroups per CU the register allocation is bounded for.  Small W: 4 (128 VGPRs; the lists allow
// 4 workgroups per CU).  Canonical walks keep two rings of W registers; with the short look-ahead (PF
// above) they fit 128 VGPRs up to w = 37 (4 waves per SIMD), 168 up to w = 54 (3 waves) and 256 up to
// w = 64 with a handful of spills outside - or a few per W-block inside - the main loop (w = 51 at
// 168 VGPRs: 8 scratch operations per 51-window block).  Measured on 3.1 Gbp (tools/gpu_jit_w.py,
// round 2): the next tighter bound loses everywhere (w = 39 at 128: 1207 against 1687 Gbases/s at 168;
// w = 55 at 168: 1270 against 1449 at 256; w = 100 bounded to 256: 519 against 784 unbounded).  Forward
// walks and the reads-mode kernels gain nothing from bounds and stay unbounded.
// Forward walks with 8-bit lists (w <= 13, no super-k-mer indices): 7 workgroups per CU, i.e. at most 72 VGPRs,
// since round 3 - with the wide sequence loads the kernel needed 85 registers (5 waves per SIMD) and a third of a
// forward tile's time is look-back and copy-out, which only other resident waves can cover: k=21 w=11 on 3.1 Gbp
// 1.174 ms unbounded, 1.099 at 6, 1.057 at 7, 1.055 at 8 (tools/gpu_jit_w.py).
#ifndef MM_MIN_BLOCKS
#define MM_MIN_BLOCKS                                                                                             \
    (!CANON && !READS && !SK && W <= 13                                                                           \
         ? 7                                                                                                      \
         : (W <= 12 ? 4                                                                                           \
                    : (CANON && !READS ? (W >= 19 && W <= 37 ? 4 : (W >= 38 && W <= 54 ? 3 : (W >= 55 && W <= 64 ? 2 : 1))) \
                                       : 1)))
#endif
#ifndef MM_MIN_BLOCKS_WALK
#define MM_MIN_BLOCKS_WALK MM_MIN_BLOCKS
#endif
__global__ __launch_bounds__(kFusedThreads, MM_MIN_BLOCKS) void fused_kernel(const FusedParams p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // lane lists
    // static LDS: distinct objects, so table look-ups can be scheduled across the list stores
    __shared__ uint2 s_tab[36];  // [0..15] (out<<2)|in, [16..19] in only, [20..35] two bases in (warm-up)
    __shared__ uint32_t s_bid;
    __shared__ uint32_t s_overflow;
    __shared__ uint32_t s_done;  // waves 1.. that have finished phase 1
    __shared__ uint32_t s_wave_tot[kFusedWaves];
    __shared__ unsigned long long s_excl;

    // kernels that also carry the skip-ambiguous walk: canonical windows, positions only (the
    // reference offers run_skip_ambiguous_windows on canonical builders without super-k-mers,
    // src/lib.rs:451-496)
    constexpr bool kAmbi = CANON && !SK;
    constexpr bool kE8 = kEntry8<W, CANON, SK, READS>;       // 8-bit list entries (see kEntry8)
    constexpr uint32_t kStride = list_stride(kE8), kEB = kE8 ? 1u : 2u;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    // Layout contract of the list overflow (see "redo" in the header comment): the lists are the dynamic
    // LDS and must lie behind every static variable, so that entries past a list's capacity fall beyond
    // the lists (into padding or out of the workgroup's allocation, where the hardware drops them) and
    // never onto the tables or counters.  The addresses are link-time constants; a layout that breaks
    // the contract fails the launch with error code 2 instead of corrupting the redo.
    {
        auto lds_end = [](const void *q, size_t bytes) {
            return (uint32_t)reinterpret_cast<uintptr_t>(q) + (uint32_t)bytes;
        };
        uint32_t st_end = lds_end(s_tab, sizeof(s_tab));
        st_end = max(st_end, lds_end(&s_bid, sizeof(s_bid)));
        st_end = max(st_end, lds_end(&s_overflow, sizeof(s_overflow)));
        st_end = max(st_end, lds_end(&s_done, sizeof(s_done)));
        st_end = max(st_end, lds_end(s_wave_tot, sizeof(s_wave_tot)));
        st_end = max(st_end, lds_end(&s_excl, sizeof(s_excl)));
        if ((uint32_t)reinterpret_cast<uintptr_t>(smem) < st_end) {
            if (tid == 0) flag_error(p.out.error, 2u);
            return;
        }
    }
    // Tile id.  Default: blockIdx.x (workgroups are dispatched in index order on gfx950, which
    // the look-back needs for forward progress; its spins are bounded and report a violation,
    // upon which the host re-runs in ticket mode where an atomic counter defines the order).
    // Redo mode (split path, mm_split.hip): workgroup b walks tile redo_list[b].tile - one whose lists
    // overflowed in walk_kernel - again and stores directly from redo_list[b].prefix on; the lane counts come
    // from the tile's dump slot.  No look-back, nothing is published.
    const bool redo = !READS && p.redo_list != nullptr;
    if (redo && blockIdx.x >= *p.redo_n) return;
    if (tid == 0) {
        s_bid = redo ? p.redo_list[blockIdx.x].tile : (p.use_ticket ? atomicAdd(p.out.ticket, 1u) : blockIdx.x);
        s_overflow = redo ? 1u : 0u;
        s_done = 0;
    }
    if (tid < 16) s_tab[tid] = p.ht.t_in_out[tid];
    else if (tid < 20) s_tab[tid] = p.ht.t_in[tid - 16];
    else if (tid < 36) s_tab[tid] = p.ht.t_in2[tid - 20];
    __syncthreads();
    const uint32_t bid = __builtin_amdgcn_readfirstlane(s_bid);  // keep tile scalars in SGPRs
    const unsigned long long etag = (unsigned long long)p.epoch << kEpochShift;
    // test hook (MM_DEBUG=32): report a look-back time-out although none happened, so that the error
    // plumbing of the asynchronous entry points (mm_workspace_check) can be exercised
    if ((MM_DBG(p) & 32u) && bid == 0 && tid == 0) flag_error(p.out.error, 1u);
    if (p.trace && tid == 0) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        p.trace[10 * (size_t)bid + 0] = wall_clock64();
        p.trace[10 * (size_t)bid + 4] = ((unsigned long long)xcc << 32) | hw;
    }

    // Lane length of THIS tile.  The tail of a launch is tapered (FusedParams::taper_*, plan_taper in mm_fused.hip):
    // with uniform tiles the chip's workgroup slots finish their last tile spread over one whole slot cycle and idle
    // for half of it on average (a 387 M-window shard of the strong split: 19 of 208 us).  The tiles of the last
    // round therefore shrink linearly in dispatch order - a slot that frees up later gets a shorter tile - so that
    // all of them end at about the same time (which is also what the in-order look-back wants: a short tile that ended
    // early would only wait for its longer predecessors).  All tile-uniform scalars.
    uint32_t nblk_t = p.nblk;
    unsigned long long tile_off = 0;  // first window of the tile relative to the range's first (single-sequence runs)
    bool tapered = false;
    const bool batch = !READS && p.batch_tile_seq != nullptr;
    if (batch) {
        nblk_t = __builtin_amdgcn_readfirstlane(p.batch_tile_seq[bid].nblk);  // (batches: the tile table says)
        if (nblk_t > p.nblk) nblk_t = 0u;  // (never longer than the lists were sized for: refused below)
    } else if (!READS && bid >= p.taper_first) {
        const uint32_t j = bid - p.taper_first, lmax = p.nblk - p.taper_min_nblk;
        uint32_t l = 1u + j / p.taper_per_level;
        l = l < lmax ? l : lmax;
        nblk_t = p.nblk - l;
        // blocks per lane of the tapered tiles before this one: the full levels 1 .. l-1, then this level's share
        const unsigned long long before = (unsigned long long)p.taper_per_level * ((unsigned long long)(l - 1u) * p.nblk - (unsigned long long)(l - 1u) * l / 2u) +
                                          (unsigned long long)(j - (l - 1u) * p.taper_per_level) * nblk_t;
        tile_off = p.taper_start + before * (kFusedThreads * (uint32_t)W);
        tapered = true;
    }
    const uint32_t S = (uint32_t)W * nblk_t;
    const uint32_t NB = kFusedThreads * S;
    // (one 32 x 32 -> 64-bit product: stays on the scalar unit, so everything derived from the tile
    // origin - the buffer descriptors of the sequence loads above all - lives in SGPRs)
    // the sequence this tile belongs to (tile-uniform scalars)
    const uint32_t *seq_d = p.seq.d;
    uint32_t seq_dwords = p.seq.n_dwords, seq_base0 = p.seq.base0;
    uint32_t win_begin = p.win_begin, win_end = p.win_end, local_tile = bid, batch_s = 0, batch_win0 = 0;
    if (batch) {
        const BatchTile bt = p.batch_tile_seq[bid];
        batch_s = __builtin_amdgcn_readfirstlane(bt.seq);
        batch_win0 = __builtin_amdgcn_readfirstlane(bt.win0);
        const BatchSeq bs = p.batch_seqs[batch_s];
        {
            const unsigned long long a = (unsigned long long)reinterpret_cast<uintptr_t>(bs.d);
            // (readfirstlane returns int: without the casts a low half with bit 31 set sign-extends)
            const unsigned long long pa =
                ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32)) << 32) |
                (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a);
            seq_d = reinterpret_cast<const uint32_t *>(static_cast<uintptr_t>(pa));
        }
        seq_dwords = __builtin_amdgcn_readfirstlane(bs.n_dwords);
        seq_base0 = __builtin_amdgcn_readfirstlane(bs.base0);
        win_begin = 0;
        win_end = __builtin_amdgcn_readfirstlane(bs.n_windows);
        local_tile = batch_win0;  // (only "is this the sequence's first tile" is asked of it below)
        // a table entry that does not describe this tile must never be dereferenced
        if (batch_s >= p.batch_n || win_end == 0u || batch_win0 >= win_end || seq_d == nullptr || nblk_t == 0u) {
            if (tid == 0) {
                flag_error(p.out.error, 0xbad00000u | (bid & 0xfffffu));
                p.out.error[1] = batch_s;
            }
            return;
        }
    }
    if (p.trace && tid == 0) {
        p.trace[10 * (size_t)bid + 6] = (unsigned long long)reinterpret_cast<uintptr_t>(seq_d);
        p.trace[10 * (size_t)bid + 7] = ((unsigned long long)seq_dwords << 32) | seq_base0;
        p.trace[10 * (size_t)bid + 8] = ((unsigned long long)win_end << 32) | local_tile;
        p.trace[10 * (size_t)bid + 9] = batch_s;
    }
    const uint64_t bw0 = READS ? 0ull
                     : batch ? (uint64_t)batch_win0
                             : (uint64_t)win_begin + (tapered ? tile_off : (uint64_t)local_tile * NB);  // first window of the tile
    const uint32_t nvalid = READS ? NB
        : (uint32_t)(((uint64_t)win_end - bw0) < NB ? ((uint64_t)win_end - bw0) : NB);
    const bool partial = READS || nvalid < NB;

    // the lane lists: the dynamic LDS behind the landing area of the skip-ambiguous walk (none: land_bytes = 0)
    uint8_t *const lists = smem + p.land_bytes;
    LaneCtx ctx;
    ctx.tab = s_tab;
    ctx.list = lists + kEB * (uint32_t)tid;
    ctx.list_bytes = p.list_cap * kStride;
    ctx.dst = 0;
    ctx.nblk = nblk_t;
    ctx.seq_d = seq_d;
    ctx.seq_dwords = seq_dwords;
    // landing area of the skip-ambiguous walk's look-ahead loads (behind the lists, one slice per wave; see kAmbiLand)
    ctx.land = 0;
    if (kAmbi && ambi_land_rule(W) && p.wamb) {
        if (p.land_bytes < ambi_land_bytes(W)) {  // (a launcher that did not allocate it: never walk with a null landing area)
            if (tid == 0) flag_error(p.out.error, 4u);
            return;
        }
        ctx.land = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>(smem) + (uint32_t)wave * land_wave_bytes(W));
    }
    bool lane_active = false;
    // READS: a lane walks reads_per_lane consecutive reads; read j of lane t is read
    // (bid * 256 + t) * R + j of the batch
    // (round 6) lane-table launches: lane t of the tile walks the segment lane_segs[256 * bid + t], one per lane
    const bool segs = READS && p.lane_segs != nullptr;
    const uint32_t R = READS ? (segs ? 1u : p.reads_per_lane) : 1u;
    const uint32_t tile_read0 = bid * kFusedThreads * R;          // first read of the tile
    const uint32_t lane_read0 = tile_read0 + (uint32_t)tid * R;   // first read of the lane
    ctx.list_used = 0;
    // sets ctx up for read j of the lane (READS); returns whether the lane has windows to walk
    auto setup_read = [&](uint32_t j) -> bool {
        if (segs) {
            const LaneSeg sg = p.lane_segs[(size_t)bid * kFusedThreads + (uint32_t)tid];
            const uint32_t origin = __builtin_amdgcn_readfirstlane(p.seg_tile_origin[bid]);  // (tile-uniform: SGPR descriptor)
            const uint32_t cnt = sg.count < S ? sg.count : S;  // (never more than the lists were sized for)
            ctx.p0 = (long long)seq_base0 + (long long)origin - 1;
            ctx.lane_bases = sg.start - origin;
            ctx.wbase = sg.win0;
            ctx.no_prev = sg.win0 == 0u;  // the read's first window: no predecessor, always emits
            ctx.rem_valid = (int)cnt;
            ctx.abase = sg.start;
            // the wave walks the blocks of its longest lane (the others' windows past their count are masked)
            uint32_t m = cnt;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, kWave));
            ctx.nblk = (__builtin_amdgcn_readfirstlane(m) + (uint32_t)W - 1u) / (uint32_t)W;
            return cnt != 0u;
        }
        const uint32_t r = lane_read0 + j;
        const bool in = r < p.n_reads;
        uint32_t len, lane_bases, abase;
        unsigned long long tile_start;
        if (p.read_starts) {
            // reads back to back: the tile's origin is its first read's start (tile-uniform), a lane's read lies where
            // the table says; a read longer than read_len is cut to it (like read_lens above read_len)
            tile_start = p.read_starts[tile_read0 < p.n_reads ? tile_read0 : p.n_reads];
            const unsigned long long s0 = in ? p.read_starts[r] : tile_start, s1 = in ? p.read_starts[r + 1] : tile_start;
            const unsigned long long ln = s1 > s0 ? s1 - s0 : 0ull;
            len = ln < (unsigned long long)p.read_len ? (uint32_t)ln : p.read_len;
            lane_bases = (uint32_t)(s0 - tile_start);
            abase = (uint32_t)s0;
        } else {
            tile_start = (unsigned long long)tile_read0 * p.read_stride;
            len = in ? (p.read_lens ? p.read_lens[r] : p.read_len) : 0u;
            lane_bases = ((uint32_t)tid * R + j) * p.read_stride;
            abase = r * p.read_stride;
        }
        const uint32_t l = p.k + (uint32_t)W - 1u;
        const uint32_t nw = len >= l ? len - l + 1u : 0u;
        ctx.p0 = (long long)seq_base0 + (long long)tile_start - 1;
        ctx.lane_bases = lane_bases;
        ctx.wbase = 0;
        ctx.no_prev = true;
        ctx.rem_valid = (int)(nw < S ? nw : S);
        ctx.abase = abase;
        // (round 6) a wave walks the blocks of ITS longest read, not of the batch's: reads shorter than the declared maximum -
        // trimmed reads, a max_read_len above the actual lengths - no longer pay for blocks in which every window is masked
        // (8 M reads of 150 bp declared as "up to 200": 0.97 -> 0.82 ms; tools/gpu_reads_var_bound.py)
        {
            uint32_t m = (uint32_t)ctx.rem_valid;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, kWave));
            const uint32_t nb = (__builtin_amdgcn_readfirstlane(m) + (uint32_t)W - 1u) / (uint32_t)W;
            ctx.nblk = nb < nblk_t ? nb : nblk_t;
        }
        return nw != 0u;
    };
    // wave-uniform minimum of rem_valid over the lanes that walk (all lanes take part)
    auto set_min_rem = [&](bool active) {
        int m = active ? ctx.rem_valid : 0x7fffffff;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m = min(m, __shfl_xor(m, d, kWave));
        ctx.min_rem = __builtin_amdgcn_readfirstlane(m);
    };

    // ---------------------------------------------------------------- phase 1
    uint32_t my_count = 0;
    unsigned long long read_counts = 0;  // READS: 16 bits per read of the lane
    if (READS) {
        for (uint32_t j = 0; j < R; ++j) {
            const bool act = setup_read(j);
            set_min_rem(act);
            ctx.list_used = my_count;
            uint32_t c = 0;
            if (act && !(MM_DBG(p) & 4u)) {
                bool over = false;
                const uint32_t tot = (kAmbi && p.wamb)
                                         ? lane_walk<W, CANON, HASH_RC, MODE, SK, false, true, kAmbi, kE8>(p, ctx, over)
                                         : lane_walk<W, CANON, HASH_RC, MODE, SK, false, true, false, kE8>(p, ctx, over);
                c = tot - my_count;
                if (over) s_overflow = 1;  // benign race: every writer stores 1
            }
            read_counts |= (unsigned long long)(c & 0xffffu) << (16u * j);
            my_count += c;
            lane_active = lane_active || act;
        }
    } else {
        const uint32_t lw = (uint32_t)tid * S;  // first window of the lane, tile-relative
        lane_active = lw < nvalid;
        ctx.p0 = (long long)seq_base0 + (long long)bw0 - 1;
        ctx.lane_bases = lw;
        ctx.wbase = (uint32_t)bw0 + lw;
        ctx.no_prev = (bw0 + lw == 0);
        ctx.rem_valid = (int)nvalid - (int)lw;
        ctx.abase = (uint32_t)bw0 + lw;
        set_min_rem(lane_active);
        if (redo) {
            my_count = reinterpret_cast<const uint16_t *>(p.dump + (size_t)bid * p.dump_stride)[tid];
        } else if (const bool dirty = (kAmbi && p.wamb) ? wave_has_skipped(p, ctx, S, lane_active) : false;
                   lane_active && !(MM_DBG(p) & 4u)) {
            bool over = false;
            if (kAmbi && p.wamb && dirty)
                my_count = (partial || !kTwoBodies<W>)
                               ? lane_walk<W, CANON, HASH_RC, MODE, SK, false, true, kAmbi, kE8>(p, ctx, over)
                               : lane_walk<W, CANON, HASH_RC, MODE, SK, false, kAmbi && !kTwoBodies<W>, kAmbi, kE8>(p, ctx, over);
            else my_count = partial ? lane_walk<W, CANON, HASH_RC, MODE, SK, false, true, false, kE8>(p, ctx, over)
                                    : lane_walk<W, CANON, HASH_RC, MODE, SK, false, false, false, kE8>(p, ctx, over);
            if (over) s_overflow = 1;  // benign race: every writer stores 1
        }
    }

    // ---------------------------------------------------------------- phase 2
#ifdef MM_PRIO_P2
    __builtin_amdgcn_s_setprio(MM_PRIO_P2);  // experiment: phase 2 ahead of the other tiles' walks
#endif
    if (p.trace && tid == 0) p.trace[10 * (size_t)bid + 1] = wall_clock64();
    const uint32_t incl = wave_scan_dpp(my_count);
    const uint32_t wave_total = __builtin_amdgcn_readlane(incl, kWave - 1);
    if (lane == 0) {
        s_wave_tot[wave] = wave_total;
        // LDS, in order behind the store above.  The wave that finishes phase 1 last publishes the
        // tile's aggregate at once (successors wait for that, never for this tile's look-back).
        if (atomicAdd(&s_done, 1u) == (uint32_t)(kFusedWaves - 1) && bid != 0 && !(MM_DBG(p) & 1u) && !redo) {
            uint32_t tot = 0;
#pragma unroll
            for (int v = 0; v < kFusedWaves; ++v) tot += reinterpret_cast<volatile uint32_t *>(s_wave_tot)[v];
            publish_aggregate(p.out.status, bid, tot, etag);
        }
    }
    if (wave == 0) {
        // Wave 0 runs the look-back (non-blocking, see lookback_overlapped) as soon as its own lanes
        // are done; the wave that finishes last has published the tile's aggregate.
        // (the outputs before this launch are read only by appending runs: a run that starts at 0 needs no cleared word)
        const unsigned long long carry = (bid == 0 && p.append) ? *p.out.total : 0ull;
        const unsigned long long ex =
            redo ? p.redo_list[blockIdx.x].prefix
                 : ((MM_DBG(p) & 1u) ? (unsigned long long)bid * (NB / 6u)
                                   : lookback_overlapped<(CANON ? MM_LB_SLEEP_CANON : MM_LB_SLEEP)>(
                                         p.out.status, bid, carry, p.out.error, &s_done, s_wave_tot, etag));
        if (lane == 0) s_excl = ex;
        if (p.trace && lane == 0) p.trace[10 * (size_t)bid + 2] = wall_clock64();
    }
    __syncthreads();
    if (p.trace && tid == 0) p.trace[10 * (size_t)bid + 5] = p.trace[10 * (size_t)bid + 1];
    uint32_t wave_base = 0, block_total = 0;
#pragma unroll
    for (int v = 0; v < kFusedWaves; ++v) {
        const uint32_t t = s_wave_tot[v];
        if (v < wave) wave_base += t;
        block_total += t;
    }
    const bool overflow = s_overflow != 0 && !(MM_DBG(p) & 16u);  // 16: timing experiment with half-size lists
    const unsigned long long run0 = s_excl + wave_base;  // first output slot of this wave
    const uint32_t excl = incl - my_count;
    uint32_t lane_vb = 0;  // (READS) what this lane's list entries are relative to: see copy_out_wave
    if (READS && segs) {
        // (loaded again rather than kept in registers across the walk)
        const LaneSeg sg = p.lane_segs[(size_t)bid * kFusedThreads + (uint32_t)tid];
        lane_vb = sg.win0;
        if (sg.win0 == 0u && sg.read < p.n_reads) p.read_offsets[sg.read] = run0 + excl;
    } else if (READS) {
        unsigned long long o = run0 + excl;
        for (uint32_t j = 0; j < R; ++j) {
            if (lane_read0 + j < p.n_reads) p.read_offsets[lane_read0 + j] = o;
            o += (read_counts >> (16u * j)) & 0xffffu;
        }
    }
    if (batch && tid == 0 && local_tile == 0 && !redo) p.batch_offsets[batch_s] = s_excl;

    if (!overflow) {
        copy_out_wave<kE8, SK, READS>(lists, p.out, MM_DBG(p), wave, lane,
                                      (READS ? 0u : (uint32_t)bw0) - (MODE == 0 ? 1u : 0u), S, (uint32_t)kSkShift<W>,
                                      run0, wave_total, my_count, excl, lane_vb);
    } else if (READS) {
        // some list overflowed: walk the lane's reads again, now storing straight to the output
        unsigned long long o = run0 + excl;
        for (uint32_t j = 0; j < R; ++j) {
            const bool act = setup_read(j);
            ctx.dst = o;
            bool over;
            if (act) {
                if (kAmbi && p.wamb) lane_walk<W, CANON, HASH_RC, MODE, SK, true, true, kAmbi, kE8>(p, ctx, over);
                else lane_walk<W, CANON, HASH_RC, MODE, SK, true, true, false, kE8>(p, ctx, over);
            }
            o += (read_counts >> (16u * j)) & 0xffffu;
        }
    } else if (lane_active) {
        // some list overflowed: walk the tile again, now storing straight to the output
        ctx.dst = run0 + excl;
        bool over;
        if (kAmbi && p.wamb) lane_walk<W, CANON, HASH_RC, MODE, SK, true, true, kAmbi, kE8>(p, ctx, over);
        else if (partial) lane_walk<W, CANON, HASH_RC, MODE, SK, true, true, false, kE8>(p, ctx, over);
        else lane_walk<W, CANON, HASH_RC, MODE, SK, true, false, false, kE8>(p, ctx, over);
    }
    if (p.trace) {
        __syncthreads();
        if (tid == 0) p.trace[10 * (size_t)bid + 3] = wall_clock64();
    }
    if (tid == 0 && bid == gridDim.x - 1 && !redo) {
        *p.out.total = s_excl + block_total;
        // (the caller's count word and the host's page-locked copy, so that a run costs ONE stream operation)
        if (p.out.count_out) *p.out.count_out = s_excl + block_total;
        if (p.out.total_host)
            __hip_atomic_store(p.out.total_host, s_excl + block_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (READS) p.read_offsets[p.n_reads] = s_excl + block_total;
        if (batch) p.batch_offsets[p.batch_n] = s_excl + block_total;
    }
}

// ------------------------------------------------------------------------------------------ split path
// (round 3) The walk without phase 2.  In the fused kernel a tile that has finished walking keeps its
// registers and its lists while it waits for the counts of all earlier tiles and while it copies its lists
// out - a third of the forward run.  Here the workgroup stores the raw list rows and the lane counts to a
// per-tile slot in HBM (write-through 16-byte stores), publishes the tile's count in ONE status word and
// exits: it never waits for another tile, needs no dispatch order and no ticket.  Persistent expander
// workgroups on a second stream (mm_split.hip) sum the published counts, load the rows back and run the
// same copy-out (copy_out_wave).  A tile whose list overflowed publishes its count with kSplitOverflow; the
// expander notes its first output slot in the redo list and a launch of fused_kernel in redo mode
// (FusedParams::redo_list) walks those tiles again, storing directly.
// One sequence (or a window range of it) per launch; reads and batches of sequences keep the fused kernel.
template <int W, bool CANON, bool HASH_RC, int MODE, bool SK, bool READS = false>
__global__ __launch_bounds__(kFusedThreads, MM_MIN_BLOCKS_WALK) void walk_kernel(const FusedParams p) {
    static_assert(!READS, "the split path has no reads mode");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // lane lists
    __shared__ uint2 s_tab[36];
    __shared__ uint32_t s_overflow, s_total, s_maxrows;
    __shared__ __attribute__((aligned(16))) uint16_t s_cnt[kFusedThreads];
    constexpr bool kAmbi = CANON && !SK;
    constexpr bool kE8 = kEntry8<W, CANON, SK, false>;
    constexpr uint32_t kStride = list_stride(kE8), kEB = kE8 ? 1u : 2u;
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    {  // layout contract of the list overflow, as in fused_kernel
        auto lds_end = [](const void *q, size_t bytes) {
            return (uint32_t)reinterpret_cast<uintptr_t>(q) + (uint32_t)bytes;
        };
        uint32_t st_end = lds_end(s_tab, sizeof(s_tab));
        st_end = max(st_end, lds_end(&s_overflow, sizeof(s_overflow)));
        st_end = max(st_end, lds_end(&s_total, sizeof(s_total)));
        st_end = max(st_end, lds_end(&s_maxrows, sizeof(s_maxrows)));
        st_end = max(st_end, lds_end(s_cnt, sizeof(s_cnt)));
        if ((uint32_t)reinterpret_cast<uintptr_t>(smem) < st_end) {
            if (tid == 0) flag_error(p.out.error, 2u);
            return;
        }
    }
    if (tid < 16) s_tab[tid] = p.ht.t_in_out[tid];
    else if (tid < 20) s_tab[tid] = p.ht.t_in[tid - 16];
    else if (tid < 36) s_tab[tid] = p.ht.t_in2[tid - 20];
    const uint32_t S = (uint32_t)W * p.nblk;
    const uint32_t NB = kFusedThreads * S;
    {
        // (one tile per workgroup and no loop around it: a loop here - the redo pass was one at first - costs
        // the walk 17 to 24 registers)
        const uint32_t bid = blockIdx.x;
        if (tid == 0) {
            s_overflow = 0;
            s_total = 0;
            s_maxrows = 0;
        }
        __syncthreads();
        // (one sequence or one window range of it; batches of sequences keep the fused kernel)
        const uint32_t *seq_d = p.seq.d;
        const uint32_t seq_dwords = p.seq.n_dwords, seq_base0 = p.seq.base0;
        const uint32_t win_begin = p.win_begin, win_end = p.win_end, local_tile = bid;
        const uint64_t bw0 = (uint64_t)win_begin + (uint64_t)local_tile * NB;  // first window of the tile
        const uint32_t nvalid = (uint32_t)(((uint64_t)win_end - bw0) < NB ? ((uint64_t)win_end - bw0) : NB);
        const bool partial = nvalid < NB;

        LaneCtx ctx;
        ctx.tab = s_tab;
        ctx.list = smem + kEB * (uint32_t)tid;
        ctx.list_bytes = p.list_cap * kStride;
        ctx.list_used = 0;
        ctx.dst = 0;
        ctx.nblk = p.nblk;
        ctx.seq_d = seq_d;
        ctx.seq_dwords = seq_dwords;
        const uint32_t lw = (uint32_t)tid * S;  // first window of the lane, tile-relative
        const bool lane_active = lw < nvalid;
        ctx.p0 = (long long)seq_base0 + (long long)bw0 - 1;
        ctx.lane_bases = lw;
        ctx.wbase = (uint32_t)bw0 + lw;
        ctx.no_prev = (bw0 + lw == 0);
        ctx.rem_valid = (int)nvalid - (int)lw;
        ctx.abase = (uint32_t)bw0 + lw;
        {
            int m = lane_active ? ctx.rem_valid : 0x7fffffff;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) m = min(m, __shfl_xor(m, d, kWave));
            ctx.min_rem = __builtin_amdgcn_readfirstlane(m);
        }
        uint8_t *slot = p.dump + (size_t)bid * p.dump_stride;
        {
            // ------------------------------------------------------------ the walk, into the lane lists
            uint32_t my_count = 0;
            const bool dirty = (kAmbi && p.wamb) ? wave_has_skipped(p, ctx, S, lane_active) : false;
            if (lane_active && !(MM_DBG(p) & 4u)) {
                bool over = false;
                if (kAmbi && p.wamb && dirty)
                    my_count = (partial || !kTwoBodies<W>)
                                   ? lane_walk<W, CANON, HASH_RC, MODE, SK, false, true, kAmbi, kE8>(p, ctx, over)
                                   : lane_walk<W, CANON, HASH_RC, MODE, SK, false, kAmbi && !kTwoBodies<W>, kAmbi, kE8>(p, ctx, over);
                else my_count = partial ? lane_walk<W, CANON, HASH_RC, MODE, SK, false, true, false, kE8>(p, ctx, over)
                                        : lane_walk<W, CANON, HASH_RC, MODE, SK, false, false, false, kE8>(p, ctx, over);
                if (over) s_overflow = 1;  // benign race: every writer stores 1
            }
            // ------------------------------------------------------------ dump and publish
            uint32_t tot = my_count, mx = my_count;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                tot += __shfl_xor(tot, d, kWave);
                mx = max(mx, __shfl_xor(mx, d, kWave));
            }
            s_cnt[tid] = (uint16_t)my_count;
            if (lane == 0) {
                atomicAdd(&s_total, tot);
                atomicMax(&s_maxrows, mx);
            }
            __syncthreads();
            const uint32_t total = s_total, overflow = s_overflow;
            const uint32_t rows = s_maxrows < p.list_cap ? s_maxrows : p.list_cap;
            const __amdgpu_buffer_rsrc_t dr =
                __builtin_amdgcn_make_buffer_rsrc(slot, 0, (int)p.dump_stride, 0x00020000);
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            // aux 16 = sc1: write-through, so that no release fence (a write-back of the XCD's whole L2) is
            // needed before the status word
            if (tid < (int)(kSplitHeader / 16u))
                __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(reinterpret_cast<const uint8_t *>(s_cnt) + 16 * tid),
                                                       dr, 16u * (uint32_t)tid, 0, 16);
            if (!overflow) {
                const uint32_t pieces = (rows * kStride + 15u) / 16u;
                for (uint32_t i = (uint32_t)tid; i < pieces; i += kFusedThreads)
                    __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(smem + 16u * i), dr,
                                                           kSplitHeader + 16u * i, 0, 16);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave drains its own stores ...
            __syncthreads();                                    // ... before one lane publishes
            if (tid == 0)
                st_status(&p.tile_status[bid], kSplitValid | (overflow ? kSplitOverflow : 0ull) |
                                                   ((unsigned long long)rows << 32) | (unsigned long long)total);
        }
    }
}

}  // namespace mm
