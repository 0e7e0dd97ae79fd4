// mm_generic.hip — generic kernel family: any k, any w < 2^15, any hasher tables.
//
// Three plain kernels through HBM scratch (hash -> per-window position -> ordered
// compaction).  It exists so that every (k, w) the reference accepts has a HIP path
// (the fused kernel is specialised per w) and as an independent on-device cross-check
// of the fused kernel.  It is a definition-level formulation: per-window argmin over
// w keys (src/minimizers.rs:22-28), strand vote by popcount (src/canonical.rs:18-29),
// adjacent dedup / syncmer predicate (src/collect.rs:15-76, src/syncmers.rs:19-48).
#include "mm_common.h"
#include "mm_launch.h"

namespace mm {

// ------------------------------------------------------------------ hashes
// Each lane rolls over kKmersPerLane consecutive k-mers after a k-base warm-up.
constexpr uint32_t kKmersPerLane = 64;

__global__ __launch_bounds__(kBlockThreads) void generic_hash_kernel(
    SeqView seq, HashTables ht, uint32_t k, uint64_t km_begin, uint64_t km_end,
    uint32_t *__restrict__ hash /* [km_end - km_begin] */) {
    uint64_t first = km_begin + ((uint64_t)blockIdx.x * kBlockThreads + threadIdx.x) * kKmersPerLane;
    if (first >= km_end) return;
    uint64_t last = first + kKmersPerLane;
    if (last > km_end) last = km_end;
    long long p = (long long)seq.base0 + (long long)first;  // first base of k-mer `first`
    uint32_t fw = ht.fw0, rc = ht.rc0;  // (the hasher's constant XOR terms; 0 for NtHasher)
    for (uint32_t j = 0; j < k; ++j) {
        uint32_t a = base_at(seq, p + j);
        uint2 t = ht.t_in[a];
        fw = rotl32(fw, ht.rot) ^ t.x;
        rc = rotr32(rc, ht.rot) ^ t.y;
    }
    for (uint64_t e = first;; ++e) {
        hash[e - km_begin] = ht.canonical ? fw + rc : fw;
        if (e + 1 >= last) break;
        uint32_t out = base_at(seq, p);
        uint32_t in = base_at(seq, p + k);
        uint2 t = ht.t_in_out[(out << 2) | in];
        fw = rotl32(fw, ht.rot) ^ t.x;
        rc = rotr32(rc, ht.rot) ^ t.y;
        ++p;
    }
}

constexpr uint32_t kSkipped = 0xFFFFFFFEu;

// ------------------------------------------------------ per-window position
// One window per lane: leftmost / rightmost argmin of (hash & 0xffff0000) over w k-mers.
__global__ __launch_bounds__(kBlockThreads) void generic_window_kernel(
    SeqView seq, uint32_t k, uint32_t w, int canonical_windows, uint64_t km_begin,
    const uint32_t *__restrict__ hash, uint64_t win_first, uint64_t win_end,
    uint32_t *__restrict__ winpos /* [win_end - win_first] */,
    const uint32_t *__restrict__ wamb /* skip-ambiguous: bit i = window i is skipped; or null */) {
    uint64_t i = win_first + (uint64_t)blockIdx.x * kBlockThreads + threadIdx.x;
    if (i >= win_end) return;
    if (wamb && ((wamb[i >> 5] >> (i & 31)) & 1u)) {
        winpos[i - win_first] = kSkipped;  // SKIPPED, src/minimizers.rs:18
        return;
    }
    const uint32_t *h = hash + (i - km_begin);
    uint32_t best = h[0] & 0xffff0000u;
    uint32_t left = 0, right = 0;
    for (uint32_t j = 1; j < w; ++j) {
        uint32_t v = h[j] & 0xffff0000u;
        if (v < best) {
            best = v;
            left = right = j;
        } else if (v == best) {
            right = j;
        }
    }
    uint32_t sel = left;
    if (canonical_windows) {
        // #(T|G) = number of set high bits of the 2-bit codes in the l-base window
        const uint64_t l = (uint64_t)k + w - 1;
        long long p = (long long)seq.base0 + (long long)i;
        long long pe = p + (long long)l;  // exclusive
        uint64_t tg = 0;
        for (long long q = p >> 4; q <= (pe - 1) >> 4; ++q) {
            uint32_t wd = load_dword_clamped(seq, q) & 0xAAAAAAAAu;
            long long lo = q << 4;
            if (lo < p) wd &= ~0u << (2u * (uint32_t)(p - lo));
            if (lo + 16 > pe) wd &= ~0u >> (2u * (uint32_t)(lo + 16 - pe));
            tg += __popc(wd);
        }
        sel = (2 * tg > l) ? left : right;
    }
    winpos[i - win_first] = (uint32_t)(i + sel);
}

// ----------------------------------------------------- ordered compaction
// Block = 256 lanes x 16 consecutive windows. Flags, block scan, decoupled look-back, emit.
constexpr uint32_t kWinPerLaneC = 16;
constexpr uint32_t kWinPerBlockC = kBlockThreads * kWinPerLaneC;

__global__ __launch_bounds__(kBlockThreads) void generic_compact_kernel(
    uint32_t w, uint32_t mode, uint64_t win_first /* window index of winpos[0] */,
    uint64_t win_begin /* first window to emit */, uint64_t win_end,
    const uint32_t *__restrict__ winpos, OutParams out, int skip /* drop SKIPPED windows */) {
    __shared__ uint32_t s_bid;
    __shared__ uint32_t s_wave_tot[kWavesPerBlock];
    __shared__ unsigned long long s_excl;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    if (tid == 0) s_bid = atomicAdd(out.ticket, 1u);
    __syncthreads();
    const uint32_t bid = s_bid;
    const uint64_t i0 = win_begin + (uint64_t)bid * kWinPerBlockC + (uint64_t)tid * kWinPerLaneC;

    uint32_t flags = 0;
    uint32_t prev = 0;
    bool have_prev = false;
    if (i0 < win_end && i0 > win_first) {
        prev = winpos[i0 - 1 - win_first];
        have_prev = true;
    }
    uint32_t vals[kWinPerLaneC];
#pragma unroll
    for (uint32_t j = 0; j < kWinPerLaneC; ++j) {
        uint64_t i = i0 + j;
        uint32_t p = (i < win_end) ? winpos[i - win_first] : 0u;
        bool f = false;
        if (i < win_end) {
            if (mode == 0) f = !have_prev || p != prev;
            else if (mode == 1) f = (p == (uint32_t)i) || (p == (uint32_t)i + w - 1);
            else f = (p == (uint32_t)i + w / 2);
            // src/intrinsics/dedup.rs:147-155: differs from its predecessor and is not SKIPPED
            if (skip && p == kSkipped) f = false;
        }
        vals[j] = (mode == 0) ? p : (uint32_t)i;
        flags |= (uint32_t)f << j;
        prev = p;
        have_prev = true;
    }
    uint32_t cnt = __popc(flags);
    uint32_t incl = wave_inclusive_sum(cnt);
    if (lane == kWave - 1) s_wave_tot[wave] = incl;
    __syncthreads();
    uint32_t wave_base = 0, block_total = 0;
#pragma unroll
    for (int v = 0; v < kWavesPerBlock; ++v) {
        uint32_t t = s_wave_tot[v];
        if (v < wave) wave_base += t;
        block_total += t;
    }
    if (wave == 0) {
        unsigned long long carry = (bid == 0) ? *out.total : 0ull;
        unsigned long long e = lookback_exclusive(out.status, bid, block_total, carry, out.error);
        if (lane == 0) s_excl = e;
    }
    __syncthreads();
    unsigned long long dst = s_excl + wave_base + (incl - cnt);
#pragma unroll
    for (uint32_t j = 0; j < kWinPerLaneC; ++j) {
        if (flags & (1u << j)) {
            if (dst < out.cap) {
                out.pos[dst] = vals[j];
                if (out.sk) out.sk[dst] = (uint32_t)(i0 + j);
            }
            ++dst;
        }
    }
    // the last block (in ticket order) publishes the new running total
    if (tid == 0 && bid == gridDim.x - 1) *out.total = s_excl + block_total;
}

// --------------------------------------------------------------- launcher
int launch_generic(const RunArgs &a, hipStream_t stream) {
    // a.scratch holds: hash (u32 per k-mer of a round) then winpos (u32 per window of a round)
    const uint64_t round = a.generic_round_windows;
    uint32_t *hash = reinterpret_cast<uint32_t *>(a.scratch);
    uint32_t *winpos = hash + (round + a.w + 1);
    if (a.timing_start) hipEventRecord(a.timing_start, stream);
    for (uint64_t wb = a.win_begin; wb < a.win_end; wb += round) {
        uint64_t we = wb + round < a.win_end ? wb + round : a.win_end;
        uint64_t win_first = wb > 0 ? wb - 1 : 0;  // one extra window in front for the dedup seam
        uint64_t km_begin = win_first, km_end = we + a.w - 1;
        uint64_t n_km = km_end - km_begin;
        uint32_t g1 = (uint32_t)((n_km + (uint64_t)kBlockThreads * kKmersPerLane - 1) /
                                 ((uint64_t)kBlockThreads * kKmersPerLane));
        hipLaunchKernelGGL(generic_hash_kernel, dim3(g1), dim3(kBlockThreads), 0, stream, a.seq,
                           a.ht, a.k, km_begin, km_end, hash);
        uint64_t n_w = we - win_first;
        uint32_t g2 = (uint32_t)((n_w + kBlockThreads - 1) / kBlockThreads);
        hipLaunchKernelGGL(generic_window_kernel, dim3(g2), dim3(kBlockThreads), 0, stream, a.seq,
                           a.k, a.w, a.canonical_windows, km_begin, hash, win_first, we, winpos, a.wamb);
        uint32_t g3 = (uint32_t)((we - wb + kWinPerBlockC - 1) / kWinPerBlockC);
        if (hipMemsetAsync(a.out.status, 0, sizeof(unsigned long long) * g3, stream) != hipSuccess)
            return -1;
        if (hipMemsetAsync(a.out.ticket, 0, sizeof(uint32_t), stream) != hipSuccess) return -1;
        hipLaunchKernelGGL(generic_compact_kernel, dim3(g3), dim3(kBlockThreads), 0, stream, a.w,
                           a.mode, win_first, wb, we, winpos, a.out, a.wamb ? 1 : 0);
    }
    if (a.timing_stop) hipEventRecord(a.timing_stop, stream);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

uint64_t generic_scratch_bytes(uint64_t round_windows, uint32_t w) {
    return sizeof(uint32_t) * ((round_windows + w + 1) + (round_windows + 2));
}
uint64_t generic_status_words(uint64_t round_windows) {
    return (round_windows + kWinPerBlockC - 1) / kWinPerBlockC + 1;
}

}  // namespace mm
