// mm_lanes.hip — the LANE TABLE of a reads-mode launch, built on the device (round 6).
//
// The reference's operator is `Builder::run` per read / contig (src/lib.rs:378), and its own `short` experiment spans
// sequence lengths 16 .. 16 384 (bench/src/bin/paper.rs:62-115).  The reads-mode kernel of rounds 2-5 gave every read one
// lane (sized for the longest read), so a batch with one read above about 1.5 kbp fell back to one launch per read,
// and the batch mode gave every contig tiles of its own, so a 10 kbp contig filled 33 of a tile's 256 lanes.  With a lane
// table (LaneSeg, mm_common.h) a tile's 256 lanes are ANY 256 consecutive segments: a read of n_w windows takes
// ceil(n_w / S) consecutive lanes of (almost) equal length, a short read one lane, a read without a window one empty
// lane (it stores the read's output offset).  Four small kernels on the run's stream (two for up to 2048 reads), no host round trip:
//   seg_count_kernel   segments per read, summed per block of 2048 reads
//   seg_scan_kernel    exclusive scan of the block sums (one workgroup)
//   seg_first_kernel   first lane of every read (seg_first[r], n_reads + 1 entries)
//   seg_fill_kernel    one thread per lane of the (padded) table: binary search for its read, its share of the read's
//                      windows, and the tile's origin (smallest start of its lanes)
// The grid of the walk is sized from an upper bound of the lane count (n_reads + total_bases / S), so nothing is read
// back; lanes behind the last real one are empty (count 0) and their tiles pass through the look-back with nothing.
#include "mm_common.h"
#include "mm_launch.h"

namespace mm {

namespace {

constexpr uint32_t kSegItems = 8;                               // reads per thread of the counting kernels
constexpr uint32_t kSegBlock = kBlockThreads * kSegItems;       // reads per workgroup

struct SegGeom {
    SegSource src;
    uint32_t n_reads;
    uint32_t l;  // k + w - 1: bases of a window
    uint32_t S;  // windows per lane at most
};

__device__ __forceinline__ uint32_t seg_len(const SegGeom &g, uint32_t r) {
    unsigned long long len;
    if (g.src.lens) len = g.src.lens[r];
    else if (g.src.starts) {
        const unsigned long long s0 = g.src.starts[r], s1 = g.src.starts[r + 1];
        len = s1 > s0 ? s1 - s0 : 0ull;
    } else len = g.src.max_len;
    return len < (unsigned long long)g.src.max_len ? (uint32_t)len : g.src.max_len;
}
__device__ __forceinline__ uint32_t seg_windows(const SegGeom &g, uint32_t r) {
    const uint32_t len = seg_len(g, r);
    return len >= g.l ? len - g.l + 1u : 0u;
}
// lanes of a read: every read owns at least one (a read without a window: an empty lane that stores its offset)
__device__ __forceinline__ uint32_t seg_lanes(const SegGeom &g, uint32_t r) {
    const uint32_t nw = seg_windows(g, r);
    return nw ? (nw + g.S - 1u) / g.S : 1u;
}
__device__ __forceinline__ unsigned long long seg_start(const SegGeom &g, uint32_t r) {
    return g.src.starts ? g.src.starts[r] : (unsigned long long)r * g.src.stride;
}

// sum over the workgroup (every thread gets it); `lds` holds kWavesPerBlock words
__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t *lds) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) lds[threadIdx.x / kWave] = v;
    __syncthreads();
    uint32_t t = 0;
#pragma unroll
    for (int i = 0; i < kWavesPerBlock; ++i) t += lds[i];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(kBlockThreads) void seg_count_kernel(const SegGeom g, uint32_t *blk_sums) {
    __shared__ uint32_t lds[kWavesPerBlock];
    const uint32_t r0 = (blockIdx.x * kBlockThreads + threadIdx.x) * kSegItems;
    uint32_t v = 0;
#pragma unroll
    for (uint32_t i = 0; i < kSegItems; ++i)
        if (r0 + i < g.n_reads) v += seg_lanes(g, r0 + i);
    const uint32_t t = block_sum(v, lds);
    if (threadIdx.x == 0) blk_sums[blockIdx.x] = t;
}

// exclusive scan of blk_sums[0 .. n) in place; blk_sums[n] receives the total
__global__ __launch_bounds__(1024) void seg_scan_kernel(uint32_t *blk_sums, uint32_t n) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024u) {
        const uint32_t i = base + tid;
        const uint32_t v = i < n ? blk_sums[i] : 0u;
        const uint32_t incl = wave_inclusive_sum(v);
        if (lane == kWave - 1) wsum[wave] = incl;
        __syncthreads();
        uint32_t before = carry_s;
        for (uint32_t q = 0; q < wave; ++q) before += wsum[q];
        if (i < n) blk_sums[i] = before + incl - v;
        __syncthreads();
        if (tid == 1023u) carry_s = before + incl;
        __syncthreads();
    }
    if (tid == 0) blk_sums[n] = carry_s;
}

// (also: tile_r0[b] = the read that owns lane 256 b, the first lane of tile b - written by the read's own thread, so that
// seg_fill_kernel starts without a search of the whole array; a read that spans many tiles loops over them)
__global__ __launch_bounds__(kBlockThreads) void seg_first_kernel(const SegGeom g, const uint32_t *blk_sums, uint32_t n_blocks,
                                                                  uint32_t *seg_first, uint32_t *tile_r0, uint32_t n_tiles) {
    __shared__ uint32_t wsum[kWavesPerBlock];
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const uint32_t r0 = (blockIdx.x * kBlockThreads + tid) * kSegItems;
    uint32_t c[kSegItems], v = 0;
#pragma unroll
    for (uint32_t i = 0; i < kSegItems; ++i) {
        c[i] = r0 + i < g.n_reads ? seg_lanes(g, r0 + i) : 0u;
        v += c[i];
    }
    const uint32_t incl = wave_inclusive_sum(v);
    if (lane == kWave - 1) wsum[wave] = incl;
    __syncthreads();
    // (n_blocks == 1 - up to 2048 reads: this workgroup is the whole scan, the counting and scanning kernels are not launched -
    // two launches less on a small batch's stream, 84 -> 78 us for 100 reads of 10 kbp, tools/gpu_lanes_small.py)
    uint32_t at = (n_blocks == 1u ? 0u : blk_sums[blockIdx.x]) + incl - v;
    for (uint32_t q = 0; q < wave; ++q) at += wsum[q];
#pragma unroll
    for (uint32_t i = 0; i < kSegItems; ++i) {
        if (r0 + i < g.n_reads) {
            seg_first[r0 + i] = at;
            for (uint32_t b = (at + kFusedThreads - 1u) / kFusedThreads; b < n_tiles && b * kFusedThreads < at + c[i]; ++b) tile_r0[b] = r0 + i;
        }
        at += c[i];
    }
    if (blockIdx.x == 0 && tid == 0) {
        uint32_t total = 0;
        if (n_blocks == 1u) {
#pragma unroll
            for (int q = 0; q < kWavesPerBlock; ++q) total += wsum[q];
        } else {
            total = blk_sums[n_blocks];
        }
        seg_first[g.n_reads] = total;
    }
}

__global__ __launch_bounds__(kFusedThreads) void seg_fill_kernel(const SegGeom g, const uint32_t *seg_first, LaneSeg *table,
                                                                 uint32_t *tile_origin, uint32_t *error) {
    __shared__ uint32_t wmin[kFusedWaves];
    const uint32_t lane_id = blockIdx.x * kFusedThreads + threadIdx.x;
    const uint32_t total = seg_first[g.n_reads];
    // more lanes than the grid was sized for: the caller's total_bases understates its reads (the bound is n_reads +
    // total_bases / S).  The run fails with error code 5 instead of dropping the reads behind the table.
    if (lane_id == 0 && total > gridDim.x * kFusedThreads) flag_error(error, 5u);
    LaneSeg sg{0u, 1u, 0u, 0xffffffffu};  // (behind the table: no window, not a read's first lane)
    // The read a lane belongs to is the last r with seg_first[r] <= lane_id (strictly increasing: every read owns a lane).
    // The read r0 of the tile's first lane comes from seg_first_kernel (tile_origin[b] holds it until this kernel stores the
    // origin there); the tile's other lanes belong to reads r0 .. r0 + 255 at most, whose entries go to LDS in one coalesced
    // load and are searched there.  (A search of the whole array per lane - or per tile, by one thread - is a chain of 18
    // dependent loads per workgroup: 100 us per 8 M lanes, 7 % of the long-read row, profiles/r06_f_LONGREADS.txt.)
    __shared__ uint32_t s_first[kFusedThreads + 1];
    const uint32_t lane0 = blockIdx.x * kFusedThreads;
    uint32_t r0 = lane0 < total ? tile_origin[blockIdx.x] : 0u;
    r0 = r0 < g.n_reads ? r0 : g.n_reads - 1u;
    {
        const uint32_t i0 = r0 + threadIdx.x;
        s_first[threadIdx.x] = seg_first[i0 < g.n_reads ? i0 : g.n_reads];  // (behind the last read: the total, above every lane)
        if (threadIdx.x == 0) s_first[kFusedThreads] = seg_first[r0 + kFusedThreads < g.n_reads ? r0 + kFusedThreads : g.n_reads];
    }
    __syncthreads();
    if (lane_id < total) {
        uint32_t lo = 0, hi = kFusedThreads;
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo + 1u) / 2u;
            if (s_first[mid] <= lane_id) lo = mid;
            else hi = mid - 1u;
        }
        const uint32_t r = r0 + lo, f = s_first[lo], ns = s_first[lo + 1 <= kFusedThreads ? lo + 1 : kFusedThreads] - f, j = lane_id - f;
        const uint32_t nw = seg_windows(g, r);
        // the read's windows in ns shares of (almost) equal length: share j = base + (j < rem) windows
        const uint32_t base = nw / ns, rem = nw % ns;
        sg.win0 = j * base + (j < rem ? j : rem);
        sg.count = base + (j < rem ? 1u : 0u);
        sg.start = (uint32_t)(seg_start(g, r) + sg.win0);
        sg.read = r;
    }
    table[lane_id] = sg;
    uint32_t m = sg.count ? sg.start : 0xffffffffu;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, d, kWave));
    if ((threadIdx.x & (kWave - 1)) == 0) wmin[threadIdx.x / kWave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0xffffffffu;
#pragma unroll
        for (int i = 0; i < kFusedWaves; ++i) t = min(t, wmin[i]);
        tile_origin[blockIdx.x] = t == 0xffffffffu ? 0u : t;
    }
}

}  // namespace

uint64_t lane_table_blocks(uint64_t n_reads) { return (n_reads + kSegBlock - 1) / kSegBlock; }

// Queues the four kernels.  b.blk_sums holds lane_table_blocks(n_reads) + 1 words, b.seg_first n_reads + 1, b.table
// plan.tiles * 256 entries, b.tile_origin plan.tiles.  Returns 0 or -1.
int launch_lane_table(const SegSource &src, uint64_t n_reads, uint32_t l, const SegPlan &plan, const SegBuffers &b,
                      uint32_t *error, hipStream_t stream) {
    if (n_reads == 0 || n_reads >= (1ull << 32) || plan.tiles == 0) return -1;
    SegGeom g;
    g.src = src;
    g.n_reads = (uint32_t)n_reads;
    g.l = l;
    g.S = plan.S;
    const uint32_t nb = (uint32_t)lane_table_blocks(n_reads);
    if (nb > 1) {
        hipLaunchKernelGGL(seg_count_kernel, dim3(nb), dim3(kBlockThreads), 0, stream, g, b.blk_sums);
        hipLaunchKernelGGL(seg_scan_kernel, dim3(1), dim3(1024), 0, stream, b.blk_sums, nb);
    }
    hipLaunchKernelGGL(seg_first_kernel, dim3(nb), dim3(kBlockThreads), 0, stream, g, b.blk_sums, nb, b.seg_first, b.tile_origin,
                       (uint32_t)plan.tiles);
    hipLaunchKernelGGL(seg_fill_kernel, dim3((uint32_t)plan.tiles), dim3(kFusedThreads), 0, stream, g, b.seg_first, b.table,
                       b.tile_origin, error);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace mm
