// mm_fused.hip — instantiations and launcher of the fused kernel (mm_fused_impl.h).
//
// The kernel is specialised per window size w (the sliding-min ring lives in w registers);
// k and the hasher tables are runtime parameters.  Window sizes without an instance fall
// back to the generic family (mm_generic.hip) — still HIP, never the CPU.
#include "mm_fused_impl.h"
#include "mm_fused_inst.h"
#include "mm_env.h"
#include "mm_launch.h"
#include "mm_split.h"

#include <cstdlib>
#include <cstdio>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

namespace mm {

namespace {

using KernelFn = FusedKernelFn;
using Instance = FusedInstance;

// MM_DEBUG (timing experiments that give WRONG RESULTS by design, and the test hook 32) exists only in the experiments
// build of the library (-DMM_EXPERIMENTS, libsimd_minimizers_amd_exp.so): the product never reads it, and its
// prebuilt kernels ignore FusedParams::debug (MM_DBG in mm_fused_impl.h).  In the experiments build a non-zero value
// also routes every launch through the run-time specialisation, whose kernels are compiled with -DMM_EXPERIMENTS.
uint32_t debug_switches() {
#ifdef MM_EXPERIMENTS
    const char *dbg = mm_exp_env("MM_DEBUG");
    return dbg ? (uint32_t)atoi(dbg) : 0u;
#else
    return 0u;
#endif
}
bool force_jit_wanted() {
#ifdef MM_EXPERIMENTS
    return mm_exp_env("MM_JIT_FORCE") != nullptr || debug_switches() != 0u;
#else
    return false;
#endif
}

const Instance *find_instance(uint32_t w, int canonical_windows, int hasher_canonical) {
    using Getter = const Instance *(*)(int *);
    static const Getter kGroups[] = {fused_instances_a, fused_instances_b, fused_instances_c, fused_instances_d, fused_instances_e, fused_instances_f, fused_instances_g, fused_instances_h, fused_instances_i,
                                     fused_instances_j, fused_instances_k, fused_instances_l, fused_instances_m};
    for (Getter get : kGroups) {
        int n = 0;
        const Instance *inst = get(&n);
        for (int i = 0; i < n; ++i)
            if (inst[i].w == w && inst[i].canon == (canonical_windows != 0) &&
                inst[i].hash_rc == (hasher_canonical != 0))
                return &inst[i];
    }
    return nullptr;
}

double emit_density(uint32_t w, uint32_t mode) {
    return mode == 2 ? 1.0 / w : (mode == 1 ? 2.0 / w : 2.0 / (w + 1.0));
}

// Entries per lane list: 1.3 x the expected number of emitted windows + 8 (> 5 sigma on random
// sequence; denser tiles take the in-kernel redo path).
uint32_t list_capacity(uint32_t w, uint32_t mode, uint32_t S) {
    static const uint32_t extra = mm_exp_env("MM_CAP_EXTRA") ? (uint32_t)atoi(mm_exp_env("MM_CAP_EXTRA")) : 0u;  // experiments
    uint32_t cap = (uint32_t)(1.3 * emit_density(w, mode) * S) + 8u + extra;
    return cap > S + w ? S + w : cap;
}

// W-blocks per lane (S = w * nblk windows per lane).  Default: the longest lane whose list keeps
// 76 entries of kListStride bytes, i.e. a workgroup near 39 KB of LDS (4 workgroups per CU), but
// at least 12 W-blocks so that the k+w warm-up of a lane is amortised.
// Longest list a default lane may hold.  76 entries (39 KB of lists, 4 workgroups per CU) is the LDS
// bound; what decides below it is the cache footprint of the resident lanes: every lane streams its
// own span of the sequence, S / 4 bytes apart from its neighbour's, and once the spans of all resident
// lanes of a CU (1024 for the register-bound canonical walks, up to 1792 for the forward walks) no
// longer fit its share of the XCD's 4 MB L2, the loads of the walk go to HBM line by line.  Measured
// on 2 Gbp (cap limit 76 / 62 / 51 / 44 / 38): forward w = 19: 0.99 / 0.79 / 0.74 / 0.77 / 0.76 ms,
// w = 33: 1.45 / 1.34 / 0.95 / 0.75 / 0.71; canonical w = 11: 1.20 / 1.23 / 1.32 / 1.41,
// w = 25: 1.26 / 1.19 / 1.17 / 1.16, w = 33 (3 workgroups per CU): 1.35 / 1.30 / 1.25 / 1.30.
// words the host reserves and clears per tile status (kStatusStride of the prebuilt kernels; experiments
// with run-time specialised kernels of another stride set MM_STATUS_STRIDE_HOST to at least that stride)
uint64_t status_stride_host() {
    static const uint64_t v = mm_exp_env("MM_STATUS_STRIDE_HOST") ? (uint64_t)atoi(mm_exp_env("MM_STATUS_STRIDE_HOST")) : kStatusStride;
    return v < kStatusStride ? kStatusStride : v;
}

uint32_t default_cap_limit(uint32_t w, bool canonical, bool e8 = false) {
    // (forward w <= 13: 8-bit list entries, half the LDS per list - a slightly longer lane pays: round 2, 3.1 Gbp,
    // limit 51 -> 60: w = 7 1.629 -> 1.579 ms, w = 10 1.395 -> 1.343, w = 11 1.340 -> 1.318; the lane length is
    // bounded by S + w <= 255 anyway)
    // Round 3 re-swept the limits after the wide sequence loads (a lane's loads are four times fewer, so longer lanes
    // cost the cache less; tools/gpu_caplimit.py, profiles/r03_caplimit.txt, 3.1 Gbp, ms at limit 44 / 51 / 62 / 68 /
    // 76): canonical w = 19: 1.805 / 1.746 / 1.677 / 1.656 / 1.649, w = 25: 1.741 / 1.711 / 1.699 / 1.758 / 1.768,
    // w = 33: 1.760 / 1.748 / 1.707 / 1.702 / 1.700, w = 11 and w = 51 still best at 76; forward w = 25 (30 / 38 / 44 /
    // 51 / 62): 1.069 / 1.070 / 1.026 / 1.003 / 1.174, w = 33: 0.976 / 0.973 / 0.965 / 1.168 / 1.234, w = 19 unchanged.
    if (!canonical) return e8 ? 60u : (w <= 28u ? 51u : 44u);
    if (w <= 20u) return 76u;
    if (w <= 31u) return 62u;
    return 76u;
}

uint32_t legal_nblk(uint32_t w, uint32_t mode, uint32_t want, uint32_t cap_limit = 76u) {
    if (want == 0) {
        if (const char *e = mm_env("MM_CAP_LIMIT")) cap_limit = (uint32_t)atoi(e);  // experiments
        want = (uint32_t)(68.0 / (1.3 * emit_density(w, mode)) / w) + 1u;
        while (want > 12u && list_capacity(w, mode, w * want) > cap_limit) --want;
        // open syncmers come in irregular clumps: their lists overflow far more often at the same
        // relative head-room, so keep the expected length near 30 entries (measured at k=15 w=17 on
        // 3.1 Gbp: 28 blocks 1.77 ms, 36: 2.03, 53: 2.49)
        while (mode == 2 && want > 12u && emit_density(w, mode) * w * want > 30.0) --want;
        if (want < 12u) want = 12u;
    }
    if (want < 1u) want = 1u;
    while (w * want > kFusedMaxLaneWindows && want > 1u) --want;  // 16-bit element positions inside a lane
    return want;
}

// Landing area of the skip-ambiguous walk's look-ahead loads (kAmbiLand, mm_fused_impl.h): behind the lists, for runs with
// ambiguity bits whose kernel uses it.  Returns its bytes (0: none) and the offset the kernel is told.
static uint32_t ambi_landing(uint32_t w, const void *wamb) {
    return wamb ? ambi_land_bytes((int)w) : 0u;  // (in FRONT of the lists; a multiple of 16 bytes; 0 for window sizes without it)
}

struct Geometry {
    uint32_t nblk, S, NB, list_cap, lds_bytes;
    uint64_t nblocks;
};


constexpr uint32_t kMaxLdsBytes = 159u * 1024u;  // 160 KB per CU minus the static tables


// 8-bit list entries (kEntry8 in mm_fused_impl.h; sequence and batch mode): the lane length is bounded by
// S + w <= 255 so that every element index of a lane fits a byte
bool entry8(const RunArgs &a) { return kEntry8Rule(a.w, a.canonical_windows != 0, a.out.sk != nullptr && a.mode == 0); }
uint32_t stride_of(const RunArgs &a) { return list_stride(entry8(a)); }

Geometry geometry(const RunArgs &a) {
    Geometry g;
    g.nblk = legal_nblk(a.w, a.mode, a.nblk, default_cap_limit(a.w, a.canonical_windows != 0, entry8(a)));
    if (entry8(a)) {
        const uint32_t max_nblk = (255u - a.w) / a.w;
        // (MM_E8_NOLIMIT, experiments build: a TIMING run with longer lanes - the entries wrap, the outputs are wrong, the
        // instructions and the LDS are what a walk with two 8-bit list segments per lane would have)
        if (g.nblk > max_nblk && !mm_exp_env("MM_E8_NOLIMIT")) g.nblk = max_nblk;
    }
    // Default lanes are as long as the lists (and the cache, see default_cap_limit) allow; a run too
    // short to fill the chip once with such tiles (1024 resident workgroups) gets shorter lanes, down
    // to 6 W-blocks (measured, k=21 w=11 canonical: 64 Mbp 65.7 us with 28 blocks per lane, 58.6 with
    // 24; 16 Mbp 36.2 with 28, 29.9 with 6..12; from 256 Mbp on the long lanes win).
    if (a.nblk == 0 && a.work_windows != 0) {
        const uint64_t fit = a.work_windows / (1024ull * kFusedThreads * a.w);
        if (fit < g.nblk) g.nblk = fit < 6u ? 6u : (uint32_t)fit;
    }
    // Skip-ambiguous runs (round 5): the walk streams a third array beside the two sequence streams - one bit per window
    // - and with the default lanes of the middle window sizes the resident lanes' spans no longer fit the L2: the dirty walk
    // of k=31 w=33 missed the L2 3.2 x as often as the plain walk and took 1.07 ms per Gbp with the default 26 blocks per
    // lane, 0.70 with 13; w = 25: 0.73 -> 0.63 with 14; w <= 20 and w >= 41 are flat (tools/gpu_skip_nblk.py,
    // profiles/r05_skip_dirty_walk.txt).  Lanes of at most 400 windows there.
    // (w = 36 .. 40 took part until the chunked window bits, ambi_rows_rule: with them 18 - 22 blocks read 0.71 ms against 0.81 with 10
    // at w = 38; w = 36 0.74 -> 0.68, w = 37 0.81 -> 0.68)
    if (a.nblk == 0 && a.wamb && a.w >= 21u && a.w <= 40u && !ambi_rows_rule((int)a.w)) {
        const uint32_t lim = 400u / a.w;
        if (g.nblk > lim) g.nblk = lim < 6u ? 6u : lim;
    }
    // ... and in the three-workgroup classes (38 <= w <= 54, 168 registers) the landing area of the window bits' chunks
    // (ambi_rows_rule) has to fit beside the lists three times per CU - 160 KB in units of 1280 bytes: 53 760 per workgroup,
    // 512 of them for the static tables.  The chunks are sized so that the default lanes pass (amb_row_dwords); this is the
    // guard for whatever changes either side (two workgroups per CU instead of three: k=31 w=51 0.74 -> 1.06 ms per Gbp).
    if (a.nblk == 0 && a.wamb && ambi_rows_rule((int)a.w) && a.w <= 54u) {
        const uint32_t per_cu = a.w <= 37u ? 4u : 3u;  // (the register bounds of the canonical walks, MM_MIN_BLOCKS)
        const uint32_t room = (160u * 1024u / per_cu / 1280u) * 1280u - 512u - ambi_land_bytes((int)a.w);
        while (g.nblk > 6u && list_capacity(a.w, a.mode, a.w * g.nblk) * stride_of(a) > room) --g.nblk;
    }
    // super-k-mer runs pack (window << shift) + offset-in-window into the 16-bit list entry
    // (kSkShift in mm_fused_impl.h): the lane length is bounded by S << shift <= 65536
    if (a.out.sk && a.mode == 0) {
        uint32_t sh = 1;
        while ((1u << sh) <= a.w) ++sh;
        while (g.nblk > 1u && ((uint64_t)a.w * g.nblk << sh) > 65536u) --g.nblk;
    }
    for (;;) {
        g.S = a.w * g.nblk;
        const uint32_t cap = list_capacity(a.w, a.mode, g.S);
        g.list_cap = cap;
        g.lds_bytes = cap * stride_of(a);
        // large w (run-time specialised kernels): shorten the lanes until the lists - and the skip-ambiguous walk's landing
        // area in front of them - fit the LDS
        if (g.lds_bytes + ambi_landing(a.w, a.wamb) <= kMaxLdsBytes || g.nblk == 1) break;
        g.nblk = g.nblk > 4 ? g.nblk * 7 / 8 : g.nblk - 1;
    }
    g.NB = kFusedThreads * g.S;
    if (a.batch_tile_seq) {
        g.nblocks = a.batch_tiles;
    } else {
        const uint64_t nwin = a.win_end - a.win_begin;
        g.nblocks = (nwin + g.NB - 1) / g.NB;
    }
    return g;
}

// A kernel to launch: a prebuilt instance (host symbol) or a run-time specialisation (module
// function, mm_jit.hip).
struct KernelRef {
    KernelFn host = nullptr;
    hipFunction_t mod = nullptr;
    explicit operator bool() const { return host || mod; }
};

thread_local std::string t_jit_error;

KernelRef resolve_kernel(uint32_t w, int canonical_windows, int hasher_canonical, uint32_t mode, bool sk) {
    KernelRef kr;
    if (mode > 2) return kr;
    const bool force_jit = force_jit_wanted();  // tuning experiments (experiments build only)
    const Instance *inst = force_jit ? nullptr : find_instance(w, canonical_windows, hasher_canonical);
    if (inst) {
        kr.host = inst->fn[(mode == 0 && sk) ? 3 : mode];
        return kr;
    }
    kr.mod = jit_fused_kernel(w, canonical_windows != 0, hasher_canonical != 0, (int)mode, mode == 0 && sk,
                              false, &t_jit_error);
    return kr;
}

int launch_kernel(const KernelRef &kr, uint32_t grid, uint32_t lds_bytes, hipStream_t stream, FusedParams &p,
                  hipEvent_t ev0, hipEvent_t ev1) {
    if (kr.host) {
        if (lds_bytes > 64u * 1024u &&
            hipFuncSetAttribute(reinterpret_cast<const void *>(kr.host),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return -1;
        if (mm_exp_env("MM_PRINT_OCC")) {  // occupancy experiments: what the runtime computes from the kernel's resources
            int nb = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(kr.host),
                                                               (int)kFusedThreads, lds_bytes);
            hipFuncAttributes fa{};
            (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kr.host));
            fprintf(stderr, "[mm] occupancy: %d workgroups per CU (lds %u B, %d registers, static lds %zu B)\n", nb,
                    lds_bytes, fa.numRegs, fa.sharedSizeBytes);
        }
        if (ev0) hipEventRecord(ev0, stream);
        hipLaunchKernelGGL(kr.host, dim3(grid), dim3(kFusedThreads), lds_bytes, stream, p);
        if (ev1) hipEventRecord(ev1, stream);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    // (module functions: the runtime takes the dynamic LDS size from the launch; the attribute call is a
    // hint that some runtimes reject for a hipFunction_t - its result does not matter)
    if (lds_bytes > 64u * 1024u)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kr.mod),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    (void)hipGetLastError();
    void *args[] = {&p};
    if (ev0) hipEventRecord(ev0, stream);
    const hipError_t e = hipModuleLaunchKernel(kr.mod, grid, 1, 1, kFusedThreads, 1, 1, lds_bytes, stream, args,
                                               nullptr);
    if (ev1) hipEventRecord(ev1, stream);
    if (e != hipSuccess) {
        // a launch the runtime rejects (too much dynamic LDS for a module function) is not a failed
        // run: the caller takes the generic family
        (void)hipGetLastError();
        t_jit_error = std::string("launch of the run-time specialised kernel rejected: ") + hipGetErrorString(e);
        return -2;
    }
    return 0;
}

}  // namespace

const char *fused_unavailable_reason() { return t_jit_error.c_str(); }

// True when launch_fused can be attempted: a prebuilt instance exists, or the window size is in
// the range of the run-time specialisation (the compile itself happens at the first launch; if it
// fails launch_fused returns -2 and the caller takes the generic family).
bool fused_supported(uint32_t k, uint32_t w, int canonical_windows, int hasher_canonical) {
    (void)k;
    if (find_instance(w, canonical_windows, hasher_canonical)) return true;
    return jit_enabled() && w >= 1 && w <= kJitMaxW;
}

// lower end of the lane lengths tune_whole_rounds may choose from (blocks per lane)
static uint32_t whole_rounds_lo(uint32_t nblk) { return nblk * 85u / 100u > 6u ? nblk * 85u / 100u : 6u; }

// upper bound of the tiles a tapered launch adds to a uniform one (for the status words reserved ahead of the launch)
static uint64_t taper_extra_tiles() {
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) {
        (void)hipGetLastError();
        cus = 512;
    }
    return 8ull * (uint64_t)cus + 64ull;  // one round of at most 8 workgroups per CU, the rounding of the levels, the remainder
}

uint64_t fused_status_words(const RunArgs &a) {
    // Sized for the shortest lanes tune_whole_rounds can pick (it may shorten them to 85 % of the default,
    // floor, but not below 6 blocks: 8 -> 6 blocks are 1.33 x the tiles), with the same integer arithmetic;
    // + 8: the chunked look-back variant keeps per-tile counts and per-chunk bases in the same words.
    // launch_fused checks the tuned grid against RunArgs::status_avail and keeps the default lanes if a
    // caller reserved less.
    const Geometry g = geometry(a);
    uint64_t tiles = g.nblocks;
    if (!a.batch_tile_seq && a.nblk == 0) {
        const uint32_t lo = whole_rounds_lo(g.nblk);
        if (lo < g.nblk) {
            const uint64_t nwin = a.win_end - a.win_begin, nb = (uint64_t)kFusedThreads * a.w * lo;
            const uint64_t t = (nwin + nb - 1) / nb;
            if (t > tiles) tiles = t;
        }
    }
    if (!a.batch_tile_seq && a.nblk == 0) tiles += taper_extra_tiles();  // (the tapered tail, plan_taper)
    return (tiles + 8) * status_stride_host();
}
uint32_t fused_tile_windows(const RunArgs &a) { return geometry(a).NB; }
uint64_t fused_overread_bytes() {
    // the longest lane + two load groups of the widest window (8 blocks each at most, kWideGroup) + the 20 bytes of a
    // wide load, in bytes of packed sequence (4 bases per byte), rounded up to whole cache lines
    return ((uint64_t)kFusedMaxLaneWindows + 2ull * 8ull * kJitMaxW + 3ull) / 4ull + 20ull + 127ull & ~127ull;
}

static uint32_t g_lds_pad = 0;

// Whole rounds.  A long run is walked in "rounds" of as many tiles as the chip holds workgroups
// (occupancy x CUs); with long lanes and few workgroups per CU there are only about ten of them
// (canonical k=31 w=51 on 3.1 Gbp: 8 794 tiles over 768 slots), and a last round that is one third full
// costs as much as a full one.  Within +-15 % of the default lane length, pick the one that wastes
// least: cost = ceil(tiles / slots) x (blocks per lane + the per-tile overhead in blocks).  Measured on
// 3.1 Gbp, k=31 w=51: 27 blocks (default) 1.974 ms, 29 blocks 1.904 ms, 30 blocks 2.005 ms; w=33: 12
// blocks 1.829, 13 blocks 1.890, 14 blocks 1.816 (tools/gpu_nblk2.py).  Runs of 24 rounds or more are
// left alone (at most 4 % to win), as are batches (their tile table is built from the default).
// tiles(S) = tiles of the run with S windows per lane; returns the chosen blocks per lane (g.nblk if
// nothing is to be gained)
// workgroups of this kernel the chip holds at once (occupancy x CUs); false if the runtime cannot say.
// (Answers are kept per kernel, LDS size and device: the two runtime queries cost microseconds, and a caller that
// runs one short sequence per call pays them on every launch otherwise.)
static bool resident_slots(const KernelRef &kr, uint32_t lds_bytes, int *per_cu_out, int *cus_out) {
    int per_cu = 0, cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    struct Key {
        const void *fn;
        uint32_t lds;
        int dev;
        bool operator<(const Key &o) const { return fn != o.fn ? fn < o.fn : (lds != o.lds ? lds < o.lds : dev < o.dev); }
    };
    static std::mutex mu;
    static std::map<Key, std::pair<int, int>> cache;
    const Key key{kr.host ? reinterpret_cast<const void *>(kr.host) : reinterpret_cast<const void *>(kr.mod), lds_bytes, dev};
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = cache.find(key);
        if (it != cache.end()) {
            *per_cu_out = it->second.first;
            *cus_out = it->second.second;
            return it->second.first > 0;
        }
    }
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return false;
    const hipError_t e =
        kr.host ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kr.host),
                                                               kFusedThreads, lds_bytes)
                : hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kr.mod, kFusedThreads, lds_bytes);
    if (e != hipSuccess || per_cu < 1 || cus < 1) {
        (void)hipGetLastError();
        return false;
    }
    {
        std::lock_guard<std::mutex> lock(mu);
        cache[key] = std::make_pair(per_cu, cus);
    }
    *per_cu_out = per_cu;
    *cus_out = cus;
    return true;
}

template <class TilesFn>
static uint32_t whole_rounds_nblk(const RunArgs &a, const KernelRef &kr, const Geometry &g, TilesFn tiles_of) {
    static const bool off = getenv("MM_NO_ROUNDS") != nullptr;
    if (off || a.nblk != 0 || g.nblocks <= 512) return g.nblk;
    int per_cu = 0, cus = 0;
    // (the launch's dynamic LDS is lists + landing area: both decide how many workgroups a CU holds - ADVICE r5)
    const uint32_t land = ambi_landing(a.w, a.wamb);
    if (!resident_slots(kr, g.lds_bytes + land, &per_cu, &cus)) return g.nblk;
    const double slots = 0.98 * per_cu * cus;  // (a round that is 99 % full spills into the next one)
    const double kOverheadBlocks = 2.4;  // warm-up, look-back and copy-out of a tile, in W-blocks of walking
    if ((double)g.nblocks / slots >= 24.0 || (double)g.nblocks / slots <= 1.0) return g.nblk;
    const uint32_t lds_share = (kMaxLdsBytes - (uint32_t)per_cu * 512u) / (uint32_t)per_cu;
    const uint32_t lds_limit = lds_share > land ? lds_share - land : 0u;
    uint32_t sh = 0;
    if (a.out.sk && a.mode == 0)
        for (sh = 1; (1u << sh) <= a.w;) ++sh;
    double best = 0.0;
    uint32_t best_nb = g.nblk;
    const uint32_t lo = whole_rounds_lo(g.nblk), hi = g.nblk * 115u / 100u;
    for (uint32_t nb = lo; nb <= hi; ++nb) {
        const uint64_t S = (uint64_t)a.w * nb;
        if (S > kFusedMaxLaneWindows || (sh && (S << sh) > 65536u)) continue;
        if (list_capacity(a.w, a.mode, (uint32_t)S) * stride_of(a) > lds_limit) continue;
        if (entry8(a) && S + a.w > 255u) continue;
        const double rounds = (double)tiles_of(S) / slots;
        const double cost = (double)(uint64_t)(rounds + 0.999999) * (nb + kOverheadBlocks);
        if (best == 0.0 || cost < best * 0.995 || (nb == g.nblk && cost <= best * 1.005)) {
            best = cost;
            best_nb = nb;
        }
    }
    return best_nb;
}

static void tune_whole_rounds(const RunArgs &a, const KernelRef &kr, Geometry &g) {
    if (a.batch_tile_seq) return;  // (a batch's tile table is already built: see fused_batch_nblk)
    const uint64_t nwin = a.win_end - a.win_begin;
    const uint32_t nb = whole_rounds_nblk(a, kr, g, [&](uint64_t S) {
        return (nwin + kFusedThreads * S - 1) / (kFusedThreads * S);
    });
    if (nb == g.nblk) return;
    g.nblk = nb;
    g.S = a.w * g.nblk;
    g.list_cap = list_capacity(a.w, a.mode, g.S);
    g.lds_bytes = g.list_cap * stride_of(a);
    g.NB = kFusedThreads * g.S;
    g.nblocks = (nwin + g.NB - 1) / g.NB;
}

// The same for a batch of sequences (every sequence starts a tile of its own): blocks per lane to
// build the tile table with, or 0 to keep the default.  `n_windows[s]` = windows of sequence s.
uint32_t fused_batch_nblk(const RunArgs &a, const uint64_t *n_windows, uint64_t n_seqs) {
    if (a.nblk != 0) return 0;
    Geometry g = geometry(a);
    uint64_t tiles = 0;
    for (uint64_t s = 0; s < n_seqs; ++s) tiles += (n_windows[s] + g.NB - 1) / g.NB;
    g.nblocks = tiles;
    if (tiles <= 512) return 0;
    const KernelRef kr = resolve_kernel(a.w, a.canonical_windows, (int)a.ht.canonical, a.mode, a.out.sk != nullptr);
    if (!kr) return 0;
    const uint32_t nb = whole_rounds_nblk(a, kr, g, [&](uint64_t S) {
        uint64_t t = 0;
        for (uint64_t s = 0; s < n_seqs; ++s) t += (n_windows[s] + kFusedThreads * S - 1) / (kFusedThreads * S);
        return t;
    });
    return nb == g.nblk ? 0u : nb;
}

// Tapered tail (FusedParams::taper_*).  With uniform tiles every workgroup slot of the chip finishes its last tile at
// some point of one slot cycle, so the slots idle for half a cycle on average at the end of a launch: 19 us of the 208
// a 387 M-window shard of the strong split takes (per-tile trace, tools/gpu_small_trace.py; profiles/r04_small_runs.txt),
// 14 of the 99 us of BASELINE config 2.  The tiles of the LAST round therefore shrink linearly in dispatch order, from
// the default lane length down to kTaperMinBlocks blocks, `slots / levels` tiles per level: slots free up evenly over
// a cycle in steady state, so a tile that starts later is shorter by as much and they all end together.  (A first
// design - one round of half-length and one of quarter-length tiles - measured SLOWER, 0.218 against 0.213 ms: the
// look-back completes tiles in order, so a short tile that ends before its longer predecessors waits for them while
// it holds the slot its successor needs.)  Half a round of work takes a whole round of tiles: about 1.2 blocks of
// per-tile overhead more per slot.  Single sequences and window ranges with default lanes only (a caller who pins
// the lane length gets exactly that; batches keep their tile table).  MM_NO_TAPER=1 switches it off (A/B; results
// are identical either way).
constexpr uint32_t kTaperMinBlocksDefault = 4;
// (tuning: MM_TAPER_MIN_BLOCKS = lane length of the last level; MM_TAPER_ROUNDS_PCT = tiles of the tapered stretch in % of
// the chip's slots - 100: one tile per slot)
static uint32_t taper_min_blocks() {
    const char *e = mm_env("MM_TAPER_MIN_BLOCKS");
    const int v = e ? atoi(e) : 0;
    return v >= 1 ? (uint32_t)v : kTaperMinBlocksDefault;
}
static uint64_t taper_slots_pct() {
    const char *e = mm_env("MM_TAPER_ROUNDS_PCT");
    const int v = e ? atoi(e) : 0;
    return v >= 10 ? (uint64_t)v : 100u;
}
#define kTaperMinBlocks taper_min_blocks()
struct Taper {
    uint32_t first = 0xffffffffu, per_level = 1, min_nblk = kTaperMinBlocksDefault;
    unsigned long long start = 0;
    uint64_t tiles = 0;  // tiles of the whole launch
};

static Taper plan_taper(const RunArgs &a, const KernelRef &kr, const Geometry &g) {
    Taper t;
    t.tiles = g.nblocks;
    const bool off = mm_env("MM_NO_TAPER") != nullptr;
    if (off || a.batch_tile_seq || a.nblk != 0 || g.nblk < kTaperMinBlocks + 4u) return t;
    // (tests: MM_TAPER_SLOTS pretends the chip holds this many workgroups, so that runs of a few tiles already taper -
    // and the planner then needs no device at all: tests/test_abi.py checks its tiling on the CPU)
    uint64_t slots = 0;
    if (const char *e = mm_env("MM_TAPER_SLOTS")) slots = (uint64_t)atoi(e) > 0 ? (uint64_t)atoi(e) : 0;
    if (!slots) {
        int per_cu = 0, cus = 0;
        if (!resident_slots(kr, g.lds_bytes + ambi_landing(a.w, a.wamb), &per_cu, &cus)) return t;
        slots = (uint64_t)per_cu * cus;
    }
    const uint64_t nwin = a.win_end - a.win_begin;
    const uint64_t lmax = g.nblk - kTaperMinBlocks;          // levels 1 .. lmax: g.nblk - 1 .. kTaperMinBlocks blocks
    const uint64_t per_level = (slots * taper_slots_pct() / 100 + lmax - 1) / lmax;
    const uint64_t blk_w = (uint64_t)kFusedThreads * a.w;    // windows of one block of one tile
    const uint64_t cap_w = per_level * blk_w * (lmax * g.nblk - lmax * (lmax + 1) / 2);  // windows the levels hold
    // (a run has to hold the tapered round and at least half a round of whole tiles before it: shorter ones keep
    // uniform tiles - geometry() shortens their lanes, tune_whole_rounds picks the length that fills whole rounds)
    uint64_t min_pct = 50;
    if (const char *e = mm_env("MM_TAPER_MIN_PCT")) min_pct = (uint64_t)atoi(e);  // (tuning: % of a round of whole tiles)
    if (nwin < cap_w + slots * (uint64_t)g.NB * min_pct / 100) return t;
    const uint64_t f1 = (nwin - cap_w) / g.NB;
    const uint64_t rest = nwin - f1 * g.NB;  // in [cap_w, cap_w + NB): the levels, then a few more tiles of the last level
    const uint64_t extra = (rest - cap_w + blk_w * kTaperMinBlocks - 1) / (blk_w * kTaperMinBlocks);
    const uint64_t tiles = f1 + per_level * lmax + extra;
    if (tiles >= (1ull << 31) || per_level >= (1ull << 31)) return t;
    t.first = (uint32_t)f1;
    t.min_nblk = kTaperMinBlocks;
    t.per_level = (uint32_t)per_level;
    t.start = f1 * (uint64_t)g.NB;
    t.tiles = tiles;
    return t;
}

// The tile table of a batch launch.  Sequences in input order, whole tiles each (the last one of a sequence partial).
// Long batches (the tapered round + half a round of whole tiles, as for single sequences) keep the default lanes and
// taper the LAST ROUND OF THE LAUNCH: from the point where `cap_w` windows remain in the concatenated window stream,
// tile j of the tapered stretch walks nblk - min(1 + j / per_level, lmax) blocks per lane (cut short where its
// sequence ends).  k=31 w=51 on the 24 CHM13-like contigs: a slot cycle is 140 us, so uniform tiles idle the chip for
// about 70 of the launch's 1 660 us at its end.  Shorter batches get uniform tiles of the whole-rounds tuner's length.
bool fused_batch_tiles(const RunArgs &a0, const uint64_t *n_windows, uint64_t n_seqs, std::vector<BatchTile> &tiles,
                       uint32_t *nblk_out) {
    RunArgs a = a0;
    Geometry g = geometry(a);
    uint64_t total = 0;
    for (uint64_t s = 0; s < n_seqs; ++s) total += n_windows[s];
    const uint64_t blk_w = (uint64_t)kFusedThreads * a.w;
    uint64_t zone_start = ~0ull, per_level = 1, lmax = 0;
    uint64_t fake_slots = 0;
    if (const char *e = mm_env("MM_TAPER_SLOTS")) fake_slots = (uint64_t)atoi(e) > 0 ? (uint64_t)atoi(e) : 0;
    const KernelRef kr = (a.nblk == 0 && total && !fake_slots) ? resolve_kernel(a.w, a.canonical_windows, (int)a.ht.canonical, a.mode, a.out.sk != nullptr) : KernelRef();
    int per_cu = 0, cus = 0;
    if (a.nblk == 0 && total && !mm_env("MM_NO_TAPER") && g.nblk >= kTaperMinBlocks + 4u &&
        (fake_slots || (kr && resident_slots(kr, g.lds_bytes, &per_cu, &cus)))) {
        const uint64_t slots = fake_slots ? fake_slots : (uint64_t)per_cu * cus;
        lmax = g.nblk - kTaperMinBlocks;
        per_level = (slots * taper_slots_pct() / 100 + lmax - 1) / lmax;
        const uint64_t cap_w = per_level * blk_w * (lmax * g.nblk - lmax * (lmax + 1) / 2);
        if (total >= cap_w + slots * (uint64_t)g.NB / 2) zone_start = total - cap_w;
    }
    if (zone_start == ~0ull && a.nblk == 0) {
        const uint32_t nb = fused_batch_nblk(a, n_windows, n_seqs);  // uniform tiles: whole rounds
        if (nb) {
            a.nblk = nb;
            g = geometry(a);
        }
    }
    *nblk_out = g.nblk;
    uint64_t cursor = 0, j = 0;  // windows of the concatenated stream before the next tile; tapered tiles so far
    for (uint64_t s = 0; s < n_seqs; ++s) {
        uint64_t w0 = 0;
        const uint64_t nw = n_windows[s];
        while (w0 < nw) {
            uint32_t nb = g.nblk;
            if (cursor >= zone_start) {
                uint64_t l = 1 + j / per_level;
                l = l < lmax ? l : lmax;
                nb = g.nblk - (uint32_t)l;
                ++j;
            }
            const uint64_t span = (uint64_t)nb * blk_w, take = nw - w0 < span ? nw - w0 : span;
            if (tiles.size() + 1 >= (1ull << 31)) return false;
            tiles.push_back(BatchTile{(uint32_t)s, (uint32_t)w0, nb, 0u});
            w0 += take;
            cursor += take;
        }
    }
    return true;
}

// The launch plan of a single-sequence run without a device (MM_TAPER_SLOTS set): what launch_fused would pass the
// kernel.  out[0..6] = blocks per lane, tiles, taper_first, taper_per_level, taper_min_nblk, taper_start, windows per block
// of a tile (256 x w).  Host logic only: the CPU test-suite checks that the tiles tile the window range.
// The one-round rule of launch_fused (below): a run of 0.6 to 1 round of the chip's `slots` workgroup slots with the
// default lanes gets exactly one round - every slot one tile, lanes as long as that takes.  Returns true and the new
// geometry when the rule applies.
static bool one_round_geometry(const RunArgs &a, uint64_t slots, const Geometry &gd, Geometry *g) {
    const uint64_t nwin = a.win_end - a.win_begin, blk_w = (uint64_t)kFusedThreads * a.w;
    if (!(nwin <= slots * (uint64_t)gd.NB && nwin * 10u >= slots * (uint64_t)gd.NB * 6u)) return false;
    // (the legal maximum LAST: gd.nblk carries geometry()'s bounds - 16-bit super-k-mer entries, 8-bit entries, the
    // LDS - and is below 6 for super-k-mer runs with w >= 86; the lanes are then rebuilt by geometry() itself, so that
    // every bound is applied again.  ADVICE r4, high: the floor of 6 used to come last and overflowed the 16-bit entries.)
    uint32_t nb = (uint32_t)((nwin + slots * blk_w - 1) / (slots * blk_w));
    nb = nb < 6u ? 6u : nb;
    nb = nb > gd.nblk ? gd.nblk : nb;
    RunArgs one = a;
    one.nblk = nb;
    *g = geometry(one);
    return true;
}

int fused_debug_plan(const RunArgs &a, unsigned long long *out) {
    Geometry g = geometry(a);
    // (with MM_TAPER_SLOTS the planner pretends the chip holds that many workgroups: the one-round rule needs no device
    // either, so the CPU suite covers it - tests/test_abi.py)
    if (const char *e = mm_env("MM_TAPER_SLOTS")) {
        const uint64_t slots = (uint64_t)atoi(e) > 0 ? (uint64_t)atoi(e) : 0;
        if (slots && g.nblocks >= 128 && a.nblk == 0 && !a.batch_tile_seq && !mm_env("MM_NO_ONE_ROUND")) {
            RunArgs full = a;
            full.work_windows = 0;
            one_round_geometry(a, slots, geometry(full), &g);
        }
    }
    const Taper t = plan_taper(a, KernelRef(), g);
    out[0] = g.nblk;
    out[1] = t.tiles;
    out[2] = t.first;
    out[3] = t.per_level;
    out[4] = t.min_nblk;
    out[5] = t.start;
    out[6] = (unsigned long long)kFusedThreads * a.w;
    return 0;
}

// dynamic LDS of a launch with the default lanes: the lane lists and the skip-ambiguous walk's landing area (CPU tests)
void fused_debug_lds(const RunArgs &a, unsigned long long *out2) {
    const Geometry g = geometry(a);
    out2[0] = g.lds_bytes;
    out2[1] = ambi_landing(a.w, a.wamb);
}

int launch_fused(const RunArgs &a, hipStream_t stream) {
    Geometry g = geometry(a);
    if (g.nblocks == 0) return 0;
    if (g.lds_bytes > kMaxLdsBytes) return -2;
    const KernelRef kr = resolve_kernel(a.w, a.canonical_windows, (int)a.ht.canonical, a.mode,
                                        a.out.sk != nullptr);
    if (!kr) return -2;
    // A run long enough for the tapered tail keeps the default (longest) lanes: the taper removes what the whole-rounds
    // tuner is there to avoid, and longer lanes pay less per-tile overhead (k=21 w=11 on 775 Mbp: 0.382 ms with the
    // default lanes and the taper, 0.396 with the tuner's lanes and the taper, 0.391 with neither).
    // A run of 0.6 to 1 round of the chip's slots with the default lanes gets exactly one round: every slot one tile,
    // lanes as long as that takes (geometry()'s own rule for short runs assumes 1024 slots; the kernels of the large
    // windows hold 768 or 512, the forward ones 1792).  Measured (tools/gpu_size_curve.py, profiles/r04_small_runs.txt):
    // k=31 w=51 on 268 Mbp 0.227 -> 0.180 ms, forward k=21 w=11 on 67 Mbp 0.0398 -> 0.0368; BELOW about half a round
    // the rule loses (k=31 w=51 on 134 Mbp: 0.151 ms with 768 tiles of 14 blocks against 0.111 with 1 028 of 10), so
    // shorter runs keep geometry()'s lanes.
    // (runs of fewer than 128 tiles are below every rule's threshold on any chip: no queries, no planning)
    const bool small_run = g.nblocks < 128 && !mm_env("MM_TAPER_SLOTS");
    if (!small_run && a.nblk == 0 && !a.batch_tile_seq && !mm_env("MM_NO_ONE_ROUND")) {
        RunArgs full = a;
        full.work_windows = 0;  // (the default lanes of a long run)
        const Geometry gd = geometry(full);
        int per_cu = 0, cus = 0;
        uint64_t slots = 0;
        if (const char *e = mm_env("MM_TAPER_SLOTS")) slots = (uint64_t)atoi(e) > 0 ? (uint64_t)atoi(e) : 0;  // (tests)
        if (!slots && resident_slots(kr, gd.lds_bytes + ambi_landing(a.w, a.wamb), &per_cu, &cus)) slots = (uint64_t)per_cu * cus;
        if (slots) one_round_geometry(a, slots, gd, &g);
    }
    const bool tapers = !small_run && plan_taper(a, kr, g).first != 0xffffffffu && !mm_env("MM_TUNE_ALWAYS");
    if (!tapers && !small_run) {
        const Geometry untuned = g;
        tune_whole_rounds(a, kr, g);
        // never launch (or clear) more tile status words than the caller allocated: keep the default lanes
        static const bool strict = getenv("MM_STATUS_STRICT") != nullptr;  // tests: an under-sized request fails
        if (a.status_avail && (g.nblocks + 8) * status_stride_host() > a.status_avail) {
            if (strict) return -1;
            g = untuned;
        }
        if (a.status_avail && (g.nblocks + 8) * status_stride_host() > a.status_avail) return -1;
    }
    Taper taper = small_run ? Taper() : plan_taper(a, kr, g);
    if (small_run) taper.tiles = g.nblocks;
    if (a.status_avail && (taper.tiles + 8) * status_stride_host() > a.status_avail) {
        taper = Taper();  // (a caller that reserved less than fused_status_words asked for: uniform tiles)
        taper.tiles = g.nblocks;
    }
    g.nblocks = taper.tiles;
    if (a.status_avail && (g.nblocks + 8) * status_stride_host() > a.status_avail) return -1;  // (never past the caller's words)

    FusedParams p;
    p.seq = a.seq;
    p.ht = a.ht;
    p.k = a.k;
    p.nblk = g.nblk;
    p.win_begin = (uint32_t)a.win_begin;
    p.win_end = (uint32_t)a.win_end;
    p.list_cap = g.list_cap;
    p.n_reads = 0;
    p.reads_per_lane = 1;
    p.read_stride = p.read_len = 0;
    p.read_lens = nullptr;
    p.read_starts = nullptr;
    p.lane_segs = nullptr;
    p.seg_tile_origin = nullptr;
    p.read_offsets = nullptr;
    p.wamb = a.wamb;
    p.wamb_dwords = a.wamb_dwords;
    const uint32_t land_bytes = ambi_landing(a.w, a.wamb);
    p.land_bytes = land_bytes;
    if (g.lds_bytes + land_bytes > kMaxLdsBytes) return -2;  // (cannot happen with the default lanes: 39 KB of lists above w = 32)
    p.batch_seqs = a.batch_seqs;
    p.batch_tile_seq = a.batch_tile_seq;
    p.batch_offsets = a.batch_offsets;
    p.batch_n = a.batch_n;
    p.out = a.out;
    p.dump = nullptr;
    p.dump_stride = 0;
    p.tile_status = nullptr;
    p.redo_list = nullptr;
    p.redo_n = nullptr;
    p.use_ticket = a.use_ticket ? 1u : 0u;
    p.debug = debug_switches();
    p.epoch = a.status_epoch;
    p.append = a.append ? 1u : 0u;
    p.taper_first = taper.first;
    p.taper_per_level = taper.per_level;
    p.taper_min_nblk = taper.min_nblk;
    p.taper_start = taper.start;
    // (a tagged launch reads every word of another epoch as "not yet": nothing to clear, see kEpochShift)
    if (a.status_epoch == 0 &&
        hipMemsetAsync(a.out.status, 0, sizeof(unsigned long long) * (g.nblocks + 8) * status_stride_host(), stream) != hipSuccess)
        return -1;
    // (the ticket is only read in ticket mode: one stream operation less per run otherwise)
    if (a.use_ticket && hipMemsetAsync(a.out.ticket, 0, sizeof(uint32_t), stream) != hipSuccess) return -1;
    uint32_t lds_bytes = g.lds_bytes + land_bytes;
    if (p.debug & 16u) {  // timing experiment (wrong results): lists of half the capacity, overflow ignored
        p.list_cap = g.list_cap / 2 > a.w + 2 ? g.list_cap / 2 : a.w + 2;
        lds_bytes = p.list_cap * stride_of(a) + land_bytes;
    }
    if (const char *pad = mm_env("MM_LDS_PAD")) g_lds_pad = (uint32_t)atoi(pad);  // occupancy experiments
    p.trace = nullptr;
    if (const char *tr = mm_exp_env("MM_TRACE")) {
        // timing experiment: per-tile timestamps dumped to the file MM_TRACE (synchronous)
        unsigned long long *d_tr = nullptr;
        const size_t bytes = sizeof(unsigned long long) * 10 * g.nblocks;
        if (hipMalloc(reinterpret_cast<void **>(&d_tr), bytes) != hipSuccess) return -1;
        hipMemsetAsync(d_tr, 0, bytes, stream);
        p.trace = d_tr;
        int r = launch_kernel(kr, (uint32_t)g.nblocks, lds_bytes, stream, p, a.timing_start, a.timing_stop);
        hipStreamSynchronize(stream);
        std::vector<unsigned long long> h(10 * g.nblocks);
        hipMemcpy(h.data(), d_tr, bytes, hipMemcpyDeviceToHost);
        hipFree(d_tr);
        if (FILE *f = fopen(tr, "wb")) {
            fwrite(h.data(), 1, bytes, f);
            fclose(f);
        }
        return r;
    }
    return launch_kernel(kr, (uint32_t)g.nblocks, lds_bytes + g_lds_pad, stream, p, a.timing_start, a.timing_stop);
}

// ------------------------------------------------------------------ split path (walk + expander)
// EXPERIMENTS build only since round 5: measured slower than the fused kernel in round 3
// (profiles/r03_split_ab.txt), kept as an independent cross-check of the copy-out; the shipped library carries neither
// this launcher nor mm_split.hip / mm_walk_inst.hip and does not read MM_SPLIT*.
#ifdef MM_EXPERIMENTS
namespace {

// prebuilt walk kernels (mm_walk_inst.hip); everything else is specialised at first use
}  // namespace

namespace {
KernelRef resolve_walk_kernel(uint32_t w, int canonical_windows, int hasher_canonical, uint32_t mode, bool sk) {
    KernelRef kr;
    if (mode > 2) return kr;
    const bool force_jit = force_jit_wanted();
    int n = 0;
    const WalkInstance *inst = walk_instances(&n);
    for (int i = 0; i < n && !force_jit; ++i)
        if (inst[i].w == w && inst[i].canon == (canonical_windows != 0) && inst[i].hash_rc == (hasher_canonical != 0) &&
            inst[i].mode == mode && inst[i].sk == (mode == 0 && sk)) {
            kr.host = inst[i].fn;
            return kr;
        }
    kr.mod = jit_fused_kernel(w, canonical_windows != 0, hasher_canonical != 0, (int)mode, mode == 0 && sk, false,
                              &t_jit_error, true);
    return kr;
}

uint32_t split_dump_stride(const Geometry &g) { return (kSplitHeader + g.lds_bytes + 255u) & ~255u; }

// expander workgroups: enough to keep up with the walk (measured, tools/ubench/concurrent_kernels.hip: 256
// workgroups move 2.3 TB/s, 512 move 3.6)
uint32_t split_expanders(const RunArgs &a, uint64_t tiles) {
    uint32_t e = a.canonical_windows ? 256u : 512u;
    if (const char *v = mm_env("MM_SPLIT_E")) e = (uint32_t)atoi(v);
    if (e < 1u) e = 1u;
    if (e > tiles) e = (uint32_t)tiles;
    return e;
}
}  // namespace

void split_requirements(const RunArgs &a, uint64_t *tiles, uint64_t *dump_bytes) {
    const Geometry g = geometry(a);
    *tiles = g.nblocks;
    *dump_bytes = g.nblocks * (uint64_t)split_dump_stride(g);
    if (g.lds_bytes > kMaxLdsBytes || g.nblocks >= (1ull << 31) || a.batch_tile_seq) *tiles = 0;
}

bool split_wanted(const RunArgs &a) {
    if (const char *v = mm_env("MM_SPLIT")) return v[0] == '1';
    return false;
}

int launch_split(const RunArgs &a, const SplitBuffers &b, hipStream_t stream) {
    const Geometry g = geometry(a);
    if (g.nblocks == 0) return 0;
    if (g.lds_bytes > kMaxLdsBytes || a.batch_tile_seq) return -2;
    if (a.wamb && ambi_land_rule((int)a.w)) return -2;  // (the walk kernel has no landing area: the fused kernel takes these)
    const bool sk = a.out.sk != nullptr && a.mode == 0;
    const KernelRef kr = resolve_walk_kernel(a.w, a.canonical_windows, (int)a.ht.canonical, a.mode, sk);
    if (!kr) return -2;
    const KernelRef redo_kr = resolve_kernel(a.w, a.canonical_windows, (int)a.ht.canonical, a.mode, a.out.sk != nullptr);
    if (!redo_kr) return -2;
    const uint32_t stride = split_dump_stride(g);
    if (g.nblocks * (uint64_t)stride > b.dump_bytes || g.nblocks > b.status_words || g.nblocks > b.redo_entries) return -1;

    FusedParams p;
    p.seq = a.seq;
    p.ht = a.ht;
    p.k = a.k;
    p.nblk = g.nblk;
    p.win_begin = (uint32_t)a.win_begin;
    p.win_end = (uint32_t)a.win_end;
    p.list_cap = g.list_cap;
    p.n_reads = 0;
    p.reads_per_lane = 1;
    p.read_stride = p.read_len = 0;
    p.read_lens = nullptr;
    p.read_starts = nullptr;
    p.lane_segs = nullptr;
    p.seg_tile_origin = nullptr;
    p.read_offsets = nullptr;
    p.land_bytes = 0;  // (plans that need the landing area were refused above)
    p.wamb = a.wamb;
    p.wamb_dwords = a.wamb_dwords;
    p.batch_seqs = a.batch_seqs;
    p.batch_tile_seq = a.batch_tile_seq;
    p.batch_offsets = a.batch_offsets;
    p.batch_n = a.batch_n;
    p.out = a.out;
    p.use_ticket = 0;
    p.debug = debug_switches();
    p.taper_first = 0xffffffffu;
    p.taper_per_level = 1;
    p.taper_min_nblk = 0;
    p.taper_start = 0;
    p.epoch = 0;
    p.append = 1;  // (the redo pass takes its offsets from the redo list; the walk itself has no look-back)
    p.trace = nullptr;
    p.dump = b.dump;
    p.dump_stride = stride;
    p.tile_status = b.tile_status;
    p.redo_list = nullptr;
    p.redo_n = b.redo_n;

    ExpandParams e;
    e.dump = b.dump;
    e.dump_stride = stride;
    e.tile_status = b.tile_status;
    e.n_tiles = (uint32_t)g.nblocks;
    e.S = g.S;
    e.NB = g.NB;
    e.list_cap = g.list_cap;
    e.mode_sub = a.mode == 0 ? 1u : 0u;
    e.sk_shift = 1;
    while ((1u << e.sk_shift) <= a.w && e.sk_shift < 8u) ++e.sk_shift;  // kSkShift<W>
    e.win_begin = (uint32_t)a.win_begin;
    e.redo_list = reinterpret_cast<RedoEntry *>(b.redo_list);
    e.redo_n = b.redo_n;
    e.carry = b.carry;
    e.debug = p.debug;
    e.out = a.out;

    // stream `stream`: clear, walk ........................ join, redo
    // stream `aux`   :        fork -> expander (persistent) -^
    if (hipMemsetAsync(b.tile_status, 0, sizeof(unsigned long long) * g.nblocks, stream) != hipSuccess) return -1;
    if (hipMemsetAsync(b.redo_n, 0, sizeof(uint32_t), stream) != hipSuccess) return -1;
    if (hipMemcpyAsync(b.carry, a.out.total, sizeof(unsigned long long), hipMemcpyDeviceToDevice, stream) != hipSuccess)
        return -1;
    if (a.timing_start) hipEventRecord(a.timing_start, stream);
    if (hipEventRecord(b.ev_fork, stream) != hipSuccess) return -1;
    if (hipStreamWaitEvent(b.aux, b.ev_fork, 0) != hipSuccess) return -1;
    // the walk first: it never waits for anything, so the pair cannot deadlock whatever the hardware does
    // with the two queues (if they were serialised the expander would simply run after the walk)
    int r = launch_kernel(kr, (uint32_t)g.nblocks, g.lds_bytes + g_lds_pad, stream, p, nullptr, nullptr);
    if (r) return r;
    // timing experiments: MM_SPLIT_SERIAL=1 starts the expander only when the walk has finished (each kernel
    // alone on the chip), MM_SPLIT_NO_EXPAND=1 leaves it out (wrong results: the walk and its dump alone)
    static const bool serial = mm_exp_env("MM_SPLIT_SERIAL") != nullptr, no_expand = mm_exp_env("MM_SPLIT_NO_EXPAND") != nullptr;
    if (serial) {
        if (hipEventRecord(b.ev_fork, stream) != hipSuccess) return -1;
        if (hipStreamWaitEvent(b.aux, b.ev_fork, 0) != hipSuccess) return -1;
    }
    if (!no_expand && launch_expand(e, entry8(a), sk, split_expanders(a, g.nblocks), g.lds_bytes, b.aux)) return -1;
    if (hipEventRecord(b.ev_join, b.aux) != hipSuccess) return -1;
    if (hipStreamWaitEvent(stream, b.ev_join, 0) != hipSuccess) return -1;
    // tiles whose lists overflowed (low-complexity sequence): walked again by the fused kernel in redo mode,
    // one workgroup per entry of the redo list (the grid covers the worst case; the others exit at once)
    static const bool no_redo = mm_exp_env("MM_SPLIT_NO_REDO") != nullptr;  // timing experiment
    if (!no_redo) {
        p.redo_list = reinterpret_cast<const RedoEntry *>(b.redo_list);
        r = launch_kernel(redo_kr, (uint32_t)g.nblocks, g.lds_bytes, stream, p, nullptr, nullptr);
        if (r) return r;
    }
    if (a.timing_stop) hipEventRecord(a.timing_stop, stream);
    return 0;
}
#endif  // MM_EXPERIMENTS

// ------------------------------------------------------------------ reads mode
namespace {
const FusedReadsInstance *find_reads_instance(uint32_t w, int canonical_windows, int hasher_canonical) {
    using Getter = const FusedReadsInstance *(*)(int *);
    static const Getter kGroups[] = {fused_reads_instances_a, fused_reads_instances_b,
                                     fused_reads_instances_c, fused_reads_instances_d,
                                     fused_reads_instances_e};
    for (Getter get : kGroups) {
        int n = 0;
        const FusedReadsInstance *inst = get(&n);
        for (int i = 0; i < n; ++i)
            if (inst[i].w == w && inst[i].canon == (canonical_windows != 0) &&
                inst[i].hash_rc == (hasher_canonical != 0))
                return &inst[i];
    }
    return nullptr;
}
}  // namespace

int fused_prebuilt_windows(bool canonical, bool reads, uint32_t *out, int capacity) {
    int n = 0;
    for (uint32_t w = 1; w <= kJitMaxW; ++w) {
        const bool have = reads ? find_reads_instance(w, canonical, canonical) != nullptr
                                : find_instance(w, canonical, canonical) != nullptr;
        if (!have) continue;
        if (out && n < capacity) out[n] = w;
        ++n;
    }
    return n;
}

bool fused_reads_supported(uint32_t w, int canonical_windows, int hasher_canonical, uint32_t mode) {
    if (mode > 2) return false;
    // (syncmer modes have no prebuilt reads-mode kernels: compiled at first use, cached on disk)
    if (mode == 0 && find_reads_instance(w, canonical_windows, hasher_canonical)) return true;
    return jit_enabled() && w >= 1 && w <= kJitMaxW;
}

uint64_t fused_reads_status_words(const ReadsArgs &a) {
    return ((a.n_reads + kFusedThreads - 1) / kFusedThreads + 8) * status_stride_host();
}
uint64_t fused_status_stride() { return status_stride_host(); }

namespace {
KernelRef resolve_reads_kernel(const ReadsArgs &a) {
    KernelRef kr;
    const bool sk = a.out.sk != nullptr && a.mode == 0;
    // (experiments build: MM_JIT_FORCE / MM_DEBUG route the reads-mode launches through the run-time specialisation too)
    const FusedReadsInstance *inst =
        (a.mode == 0 && !sk && !force_jit_wanted()) ? find_reads_instance(a.w, a.canonical_windows, (int)a.ht.canonical) : nullptr;
    if (inst)
        kr.host = inst->fn;
    else
        kr.mod = jit_fused_kernel(a.w, a.canonical_windows != 0, a.ht.canonical != 0, (int)a.mode, sk, true,
                                  &t_jit_error);
    return kr;
}
}  // namespace

// ---- lane-table launches (round 6; mm_lanes.hip, FusedParams::lane_segs)
// Lane length: the default lanes of the sequence mode with 16-bit list entries (the reads-mode kernels keep 16 bits) -
// the same trade of per-lane fixed cost against the L2 footprint of the resident lanes' spans.
int fused_segments_plan(const ReadsArgs &a, uint64_t total_bases, uint32_t nblk_want, SegPlan *plan) {
    if (a.n_reads == 0 || a.n_reads >= (1ull << 32) || a.mode > 2) return -3;
    const bool sk = a.out.sk != nullptr && a.mode == 0;
    uint32_t nblk = legal_nblk(a.w, a.mode, nblk_want, default_cap_limit(a.w, a.canonical_windows != 0, false));
    if (sk) {  // packed (window, offset) list entries: S << shift <= 65536 (kSkShift, mm_fused_impl.h)
        uint32_t sh = 1;
        while ((1u << sh) <= a.w) ++sh;
        while (nblk > 1u && ((uint64_t)a.w * nblk << sh) > 65536u) --nblk;
        if (((uint64_t)a.w * nblk << sh) > 65536u) return -3;
    }
    // skip-ambiguous runs: the landing area of the walk's look-ahead loads in front of the lists
    const uint32_t land_bytes = ambi_landing(a.w, a.wamb);
    uint32_t cap = 0;
    for (;;) {
        cap = list_capacity(a.w, a.mode, a.w * nblk);
        if (cap * kListStride + land_bytes <= kMaxLdsBytes || nblk == 1u) break;
        nblk = nblk > 4u ? nblk * 7u / 8u : nblk - 1u;
    }
    if (cap * kListStride + land_bytes > kMaxLdsBytes || a.w * nblk + a.w > kFusedMaxLaneWindows) return -3;
    plan->nblk = nblk;
    plan->S = a.w * nblk;
    plan->list_cap = cap;
    plan->lds_bytes = cap * kListStride;
    // every read owns at least one lane and ceil(windows / S) <= windows / S + 1 of them
    const uint64_t lanes = a.n_reads + total_bases / plan->S + 1u;
    plan->tiles = (lanes + kFusedThreads - 1) / kFusedThreads;
    plan->lanes_cap = plan->tiles * kFusedThreads;
    if (plan->lanes_cap >= (1ull << 32) || plan->tiles >= (1ull << 31)) return -3;
    return 0;
}

int launch_fused_segments(const ReadsArgs &a, const SegSource &src, const SegPlan &plan, const SegBuffers &b,
                          hipStream_t stream) {
    const KernelRef kr = resolve_reads_kernel(a);
    if (!kr) return -2;
    if (launch_lane_table(src, a.n_reads, a.k + a.w - 1u, plan, b, a.out.error, stream)) return -1;
    FusedParams p;
    p.seq = a.seq;
    p.ht = a.ht;
    p.k = a.k;
    p.nblk = plan.nblk;
    p.win_begin = p.win_end = 0;
    p.list_cap = plan.list_cap;
    p.use_ticket = a.use_ticket ? 1u : 0u;
    p.debug = 0;
    p.epoch = a.status_epoch;
    p.append = 0;
    p.taper_first = 0xffffffffu;
    p.taper_per_level = 1;
    p.taper_min_nblk = 0;
    p.taper_start = 0;
    p.n_reads = (uint32_t)a.n_reads;
    p.reads_per_lane = 1;
    p.read_stride = a.read_stride;
    p.read_len = a.read_len;
    p.read_lens = nullptr;
    p.read_starts = nullptr;
    p.lane_segs = b.table;
    p.seg_tile_origin = b.tile_origin;
    p.read_offsets = a.read_offsets;
    p.wamb = a.wamb;
    p.wamb_dwords = a.wamb_dwords;
    p.land_bytes = ambi_landing(a.w, a.wamb);
    p.batch_seqs = nullptr;
    p.batch_tile_seq = nullptr;
    p.batch_offsets = nullptr;
    p.batch_n = 0;
    p.trace = nullptr;
    p.dump = nullptr;
    p.dump_stride = 0;
    p.tile_status = nullptr;
    p.redo_list = nullptr;
    p.redo_n = nullptr;
    p.out = a.out;
    if (a.status_epoch == 0 &&
        hipMemsetAsync(a.out.status, 0, sizeof(unsigned long long) * (plan.tiles + 8) * status_stride_host(), stream) != hipSuccess)
        return -1;
    if (a.use_ticket && hipMemsetAsync(a.out.ticket, 0, sizeof(uint32_t), stream) != hipSuccess) return -1;
    if (const char *pad = mm_env("MM_LDS_PAD")) g_lds_pad = (uint32_t)atoi(pad);
    return launch_kernel(kr, (uint32_t)plan.tiles, plan.lds_bytes + p.land_bytes + g_lds_pad, stream, p, a.timing_start, a.timing_stop);
}

int launch_fused_reads(const ReadsArgs &a, hipStream_t stream) {
    if (a.n_reads == 0) return 0;
    const uint32_t l = a.k + a.w - 1;
    const uint32_t max_nw = a.read_len >= l ? a.read_len - l + 1 : 1;
    const uint32_t nblk = (max_nw + a.w - 1) / a.w;  // every lane must be able to walk its whole read
    const uint32_t S = nblk * a.w;
    if (S + a.w > kFusedMaxLaneWindows) return -3;
    // A lane walks R consecutive reads, so that a tile holds about as many windows as a tile of the
    // sequence mode (fewer tiles: less look-back and copy-out overhead per window).  MM_READS_PER_LANE
    // overrides (experiments).
    auto cap_for = [&](uint32_t r) {
        const uint32_t c = (uint32_t)(1.3 * emit_density(a.w, a.mode) * S * r) + 8u;
        return c > (S + a.w) * r ? (S + a.w) * r : c;
    };
    // as many reads per lane (up to 4) as keep the lists near 40 KB, i.e. 4 workgroups per CU
    uint32_t R = 1;
    // ... and as keep the spans of the resident lanes in the L2 (about 100 bytes of sequence per lane, see
    // default_cap_limit)
    const uint32_t stride_like = a.read_starts ? a.read_len : a.read_stride;  // (back-to-back reads: about their length apart)
    const uint32_t r_cache = stride_like >= 400u ? 1u : 400u / (stride_like ? stride_like : 1u);
    while (R < 4u && R < r_cache && cap_for(R + 1) * kListStride <= 40u * 1024u) ++R;
    if (const char *e = mm_env("MM_READS_PER_LANE")) R = (uint32_t)atoi(e);
    if (R < 1u) R = 1u;
    if (R > 4u) R = 4u;
    const uint32_t cap = cap_for(R);
    if (cap > 65535u) return -3;  // 16-bit per-read counts
    const uint32_t lds_bytes = cap * kListStride;
    if (lds_bytes > 159u * 1024u) return -3;
    const uint64_t per_tile = (uint64_t)kFusedThreads * R;
    const uint64_t nblocks = (a.n_reads + per_tile - 1) / per_tile;
    const bool sk = a.out.sk != nullptr && a.mode == 0;
    if (sk) {  // packed (window, offset) list entries bound the read length (kSkShift, mm_fused_impl.h)
        uint32_t sh = 1;
        while ((1u << sh) <= a.w) ++sh;
        if (((uint64_t)S << sh) > 65536u) return -3;
    }
    const KernelRef kr = resolve_reads_kernel(a);
    if (!kr) return -2;

    FusedParams p;
    p.seq = a.seq;
    p.ht = a.ht;
    p.k = a.k;
    p.nblk = nblk;
    p.win_begin = p.win_end = 0;
    p.list_cap = cap;
    p.use_ticket = a.use_ticket ? 1u : 0u;
    p.debug = 0;
    p.epoch = a.status_epoch;
    p.append = 0;
    p.taper_first = 0xffffffffu;
    p.taper_per_level = 1;
    p.taper_min_nblk = 0;
    p.taper_start = 0;
    p.n_reads = (uint32_t)a.n_reads;
    p.reads_per_lane = R;
    p.read_stride = a.read_stride;
    p.read_len = a.read_len;
    p.read_lens = a.read_lens;
    p.read_starts = a.read_starts;
    p.lane_segs = nullptr;
    p.seg_tile_origin = nullptr;
    p.read_offsets = a.read_offsets;
    p.wamb = a.wamb;
    p.wamb_dwords = a.wamb_dwords;
    const uint32_t land_bytes = ambi_landing(a.w, a.wamb);
    p.land_bytes = land_bytes;
    if (lds_bytes + land_bytes > 159u * 1024u) return -3;
    p.batch_seqs = nullptr;
    p.batch_tile_seq = nullptr;
    p.batch_offsets = nullptr;
    p.batch_n = 0;
    p.trace = nullptr;
    p.dump = nullptr;
    p.dump_stride = 0;
    p.tile_status = nullptr;
    p.redo_list = nullptr;
    p.redo_n = nullptr;
    p.out = a.out;
    if (a.status_epoch == 0 &&
        hipMemsetAsync(a.out.status, 0, sizeof(unsigned long long) * (nblocks + 8) * status_stride_host(), stream) != hipSuccess)
        return -1;
    // (the ticket is only read in ticket mode: one stream operation less per run otherwise)
    if (a.use_ticket && hipMemsetAsync(a.out.ticket, 0, sizeof(uint32_t), stream) != hipSuccess) return -1;
    return launch_kernel(kr, (uint32_t)nblocks, lds_bytes + land_bytes, stream, p, a.timing_start, a.timing_stop);
}

}  // namespace mm
