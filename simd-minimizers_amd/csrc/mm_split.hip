// mm_split.hip — the expander of the split path (round 3).
//
// walk_kernel (mm_fused_impl.h) leaves, per tile, a slot in HBM with the 256 lane counts and the raw list rows
// ([entry][lane], 8- or 16-bit entries) and ONE status word {valid, overflow, rows, count}.  The expander is a
// small persistent grid on a second stream that runs beside the walk: workgroup g takes tiles g, g + E, g + 2E ...
// For each it sums the counts of the tiles since its previous one (a coalesced read of at most E status words:
// no chain of dependent hops, and the only party that ever waits is this small grid, never a walking
// workgroup), loads the rows back into LDS with 16-byte loads and runs the fused kernel's own copy-out
// (copy_out_wave): positions in window order, bounds-checked against the caller's capacity.  Tiles whose lists
// overflowed are only accounted for; their first output slot goes to the redo list.
// Output semantics as in fused_kernel: src/collect.rs:252-272 (window order), src/syncmers.rs:166-169.
#include "mm_split.h"

#ifndef MM_SPLIT_SLEEP
#define MM_SPLIT_SLEEP 16
#endif

namespace mm {
namespace {

constexpr uint32_t kSplitMaxSpins = 1u << 21;  // x ~0.5 us: about a second, then the run is reported as failed

template <bool E8, bool SK>
__global__ __launch_bounds__(kFusedThreads) void expand_kernel(const ExpandParams p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // list rows of the tile being expanded
    __shared__ uint32_t s_sum[kFusedWaves];
    __shared__ uint32_t s_wave_tot[kFusedWaves];
    __shared__ unsigned long long s_mine;
    constexpr uint32_t kStride = list_stride(E8);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    unsigned long long prefix = *p.carry;  // outputs before tile prev + 1
    long long prev = -1;
    for (uint32_t t = blockIdx.x; t < p.n_tiles; t += gridDim.x) {
        // ---- counts of the tiles (prev, t]: every one of them has to be published
        uint32_t sum = 0;
        for (long long i = prev + 1 + tid; i <= (long long)t; i += kFusedThreads) {
            unsigned long long s = ld_status(&p.tile_status[i]);
            for (uint32_t spins = 0; !(s & kSplitValid); ++spins) {
                if (spins > kSplitMaxSpins) {  // the walk never published this tile: report it, do not hang
                    flag_error(p.out.error, 1u);
                    break;
                }
                __builtin_amdgcn_s_sleep(MM_SPLIT_SLEEP);
                s = ld_status(&p.tile_status[i]);
            }
            sum += (uint32_t)s;
            if (i == (long long)t) s_mine = s;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, kWave);
        if (lane == 0) s_sum[wave] = sum;
        __syncthreads();  // (also: every wave is done with the LDS rows of the previous tile)
        uint32_t interval = 0;
#pragma unroll
        for (int v = 0; v < kFusedWaves; ++v) interval += s_sum[v];
        const unsigned long long mine = s_mine;
        const uint32_t my_total = (uint32_t)mine;
        const uint32_t rows = (uint32_t)(mine >> 32) & 0xffffu;
        const bool overflow = (mine & kSplitOverflow) != 0;
        const unsigned long long tile_prefix = prefix + interval - my_total;
        prefix += interval;
        prev = t;
        // ---- where the tile's windows start
        const uint32_t bw0 = p.win_begin + t * p.NB;
        if (t == p.n_tiles - 1 && tid == 0) *p.out.total = tile_prefix + my_total;
        if (overflow) {
            if (tid == 0) {
                const uint32_t idx = atomicAdd(p.redo_n, 1u);
                p.redo_list[idx].tile = t;
                p.redo_list[idx].pad = 0;
                p.redo_list[idx].prefix = tile_prefix;
            }
            continue;
        }
        // ---- lane counts and list rows (sc1 loads: the walk stored them write-through)
        const uint8_t *slot = p.dump + (size_t)t * p.dump_stride;
        const __amdgpu_buffer_rsrc_t dr =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(slot), 0, (int)p.dump_stride, 0x00020000);
        const uint32_t cw = __builtin_amdgcn_raw_buffer_load_b32(dr, ((uint32_t)tid >> 1) * 4u, 0, 16);
        const uint32_t cnt = (cw >> (16u * ((uint32_t)tid & 1u))) & 0xffffu;
        const uint32_t pieces = (rows * kStride + 15u) / 16u;
        for (uint32_t i0 = 0; i0 < pieces; i0 += 4u * kFusedThreads) {
            u32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = i0 + (uint32_t)u * kFusedThreads + (uint32_t)tid;
                // (pieces past the end read as zero: the bounds check of the slot)
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(dr, i < pieces ? kSplitHeader + 16u * i : 0xfffffff0u, 0, 16);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = i0 + (uint32_t)u * kFusedThreads + (uint32_t)tid;
                if (i < pieces) *reinterpret_cast<u32x4 *>(smem + 16u * i) = v[u];
            }
        }
        const uint32_t incl = wave_scan_dpp(cnt);
        const uint32_t wave_total = __builtin_amdgcn_readlane(incl, kWave - 1);
        if (lane == 0) s_wave_tot[wave] = wave_total;
        __syncthreads();
        uint32_t wave_base = 0;
#pragma unroll
        for (int v = 0; v < kFusedWaves; ++v)
            if (v < wave) wave_base += s_wave_tot[v];
        copy_out_wave<E8, SK, false>(smem, p.out, p.debug, wave, lane, bw0 - p.mode_sub, p.S, p.sk_shift,
                                     tile_prefix + wave_base, wave_total, cnt, incl - cnt);
    }
}

}  // namespace

int launch_expand(const ExpandParams &p, bool e8, bool sk, uint32_t workgroups, uint32_t lds_bytes, hipStream_t stream) {
    using Fn = void (*)(const ExpandParams);
    const Fn fn = e8 ? (Fn)expand_kernel<true, false> : (sk ? (Fn)expand_kernel<false, true> : (Fn)expand_kernel<false, false>);
    if (lds_bytes > 64u * 1024u &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_bytes) != hipSuccess)
        return -1;
    hipLaunchKernelGGL(fn, dim3(workgroups), dim3(kFusedThreads), lds_bytes, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace mm
