// Do two kernels on two streams share the chip the way a walk / expand split needs?  (round 3)
//
//   walker   : ntiles workgroups x 256 threads, `lds` bytes of dynamic LDS, WALK_VGPR registers, a VALU-bound spin
//              of about one tile's walk, then a write-through dump of `dump_vec` x 16 bytes per thread, a drained
//              flag store (count), exit.  Never waits for anything.
//   expander : E persistent workgroups x ETHREADS threads on a second stream; workgroup g takes tiles g, g + E, ...
//              in order, waits until every tile up to its own has published (coalesced poll of the E flags since its
//              previous tile), reads the dump with sc1 loads, writes `out_dw` dwords per thread of output.
// Reports: walker alone, walker + expander overlapped (walker launched first), expander alone (serial), how long
// after the first tile finished the expanders started, and the lag between a tile's flag and its expansion.
//
// build: hipcc -O3 --offload-arch=gfx950 -DWALK_VGPR=127 concurrent_kernels.hip -o concurrent_kernels.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#ifndef WALK_VGPR
#define WALK_VGPR 127  // index of the last VGPR the walker claims
#endif
#ifndef ETHREADS
#define ETHREADS 256
#endif
#define STR2(x) #x
#define STR(x) STR2(x)

typedef unsigned long long u64;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct P {
    u64 *flags;      // one per tile, 64 bytes apart
    u32x4 *dump;     // ntiles x 256 x dump_vec
    uint32_t *out;   // ntiles x 256 x out_dw
    u64 *ts;         // [3 * tile]: walker done, expander saw it, expander done
    u64 *es;         // expander start per workgroup
    int ntiles, spin, dump_vec, out_dw, E;
};

__global__ __launch_bounds__(256) void walker(const P p) {
    extern __shared__ unsigned char smem[];
    const int tid = threadIdx.x, tile = blockIdx.x;
    // force the register allocation of a real walk
    asm volatile("v_mov_b32 v" STR(WALK_VGPR) ", 0" ::: "v" STR(WALK_VGPR));
    uint32_t a = tid, b = tid * 3u, c = tid * 5u, d = tid * 7u;
    for (int i = 0; i < p.spin; ++i) {  // 8 VALU per iteration, four independent chains
        a = (a ^ b) + 0x9e3779b9u;
        b = (b ^ c) + 0x7f4a7c15u;
        c = (c ^ d) + 0x94d049bbu;
        d = (d ^ a) + 0xbf58476du;
    }
    smem[tid] = (unsigned char)(a + b + c + d);
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
        p.dump + (size_t)tile * 256 * p.dump_vec, 0, 256 * p.dump_vec * 16, 0x00020000);
    u32x4 v = {a, b, c, d};
    for (int i = 0; i < p.dump_vec; ++i)
        __builtin_amdgcn_raw_buffer_store_b128(v, r, (uint32_t)(i * 256 + tid) * 16u, 0, 16 /* sc1: write-through */);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        p.ts[3 * (size_t)tile] = wall_clock64();
        __hip_atomic_store(&p.flags[(size_t)tile * 8], 1ull + smem[0] % 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ __launch_bounds__(ETHREADS) void expander(const P p) {
    const int tid = threadIdx.x, g = blockIdx.x;
    __shared__ u64 s_sum;
    if (tid == 0) p.es[g] = wall_clock64();
    u64 prefix = 0;
    long long prev = -1;
    for (int t = g; t < p.ntiles; t += p.E) {
        // every tile in (prev, t] must have published
        if (tid == 0) s_sum = 0;
        __syncthreads();
        u64 mine = 0;
        for (long long i = prev + 1 + tid; i <= t; i += ETHREADS) {
            u64 f;
            while ((f = __hip_atomic_load(&p.flags[(size_t)i * 8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0)
                __builtin_amdgcn_s_sleep(32);
            mine += f;
        }
        atomicAdd(&s_sum, mine);
        __syncthreads();
        prefix += s_sum;
        if (tid == 0) p.ts[3 * (size_t)t + 1] = wall_clock64();
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
            p.dump + (size_t)t * 256 * p.dump_vec, 0, 256 * p.dump_vec * 16, 0x00020000);
        uint32_t acc = (uint32_t)prefix;
        for (int i = tid; i < 256 * p.dump_vec; i += ETHREADS) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (uint32_t)i * 16u, 0, 16 /* sc1 */);
            acc += v.x + v.y + v.z + v.w;
        }
        uint32_t *o = p.out + (size_t)t * 256 * p.out_dw;
        for (int i = tid; i < 256 * p.out_dw; i += ETHREADS) __builtin_nontemporal_store(acc + (uint32_t)i, &o[i]);
        __syncthreads();
        if (tid == 0) p.ts[3 * (size_t)t + 2] = wall_clock64();
        prev = t;
    }
}

#define CK(x)                                                                \
    do {                                                                     \
        hipError_t e_ = (x);                                                 \
        if (e_ != hipSuccess) {                                              \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                   \
            return 1;                                                        \
        }                                                                    \
    } while (0)

int main(int argc, char **argv) {
    P p;
    p.ntiles = argc > 1 ? atoi(argv[1]) : 20000;
    p.spin = argc > 2 ? atoi(argv[2]) : 6000;
    int lds = argc > 3 ? atoi(argv[3]) : 39408;
    p.dump_vec = argc > 4 ? atoi(argv[4]) : 8;   // x 4 KB per tile
    p.out_dw = argc > 5 ? atoi(argv[5]) : 52;    // x 1 KB per tile
    p.E = argc > 6 ? atoi(argv[6]) : 256;
    const int prio = argc > 7 ? atoi(argv[7]) : 1;
    CK(hipMalloc(&p.flags, (size_t)p.ntiles * 64));
    CK(hipMalloc(&p.dump, (size_t)p.ntiles * 256 * p.dump_vec * 16));
    CK(hipMalloc(&p.out, (size_t)p.ntiles * 256 * p.out_dw * 4));
    CK(hipMalloc(&p.ts, (size_t)p.ntiles * 24));
    CK(hipMalloc(&p.es, (size_t)p.E * 8));
    hipStream_t sa, sb;
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, lo));
    CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, prio ? hi : lo));
    hipEvent_t e0, e1, e2, e3;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreate(&e3));
    if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void *)walker, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    int occ = 0, occe = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)walker, 256, lds));
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occe, (const void *)expander, ETHREADS, 0));
    hipFuncAttributes fa{};
    CK(hipFuncGetAttributes(&fa, (const void *)walker));
    printf("tiles %d spin %d lds %d dump %d KB out %d KB per tile, E %d x %d threads, prio %d (range %d..%d); walker %d regs "
           "%d/CU, expander %d/CU\n",
           p.ntiles, p.spin, lds, p.dump_vec * 4, p.out_dw, p.E, ETHREADS, prio, lo, hi, fa.numRegs, occ, occe);
    std::vector<u64> ts((size_t)p.ntiles * 3), es(p.E);
    for (int rep = 0; rep < 3; ++rep) {
        // (1) walker alone
        CK(hipMemsetAsync(p.flags, 0, (size_t)p.ntiles * 64, sa));
        CK(hipEventRecord(e0, sa));
        hipLaunchKernelGGL(walker, dim3(p.ntiles), dim3(256), lds, sa, p);
        CK(hipEventRecord(e1, sa));
        CK(hipStreamSynchronize(sa));
        float ms_walk = 0;
        CK(hipEventElapsedTime(&ms_walk, e0, e1));
        // (2) expander alone (all flags are set)
        CK(hipEventRecord(e2, sb));
        hipLaunchKernelGGL(expander, dim3(p.E), dim3(ETHREADS), 0, sb, p);
        CK(hipEventRecord(e3, sb));
        CK(hipStreamSynchronize(sb));
        float ms_exp = 0;
        CK(hipEventElapsedTime(&ms_exp, e2, e3));
        // (3) both: walker first, expander on the other stream right behind it
        CK(hipMemsetAsync(p.flags, 0, (size_t)p.ntiles * 64, sa));
        CK(hipMemsetAsync(p.ts, 0, (size_t)p.ntiles * 24, sa));
        CK(hipStreamSynchronize(sa));
        CK(hipEventRecord(e0, sa));
        hipLaunchKernelGGL(walker, dim3(p.ntiles), dim3(256), lds, sa, p);
        CK(hipEventRecord(e1, sa));
        CK(hipEventRecord(e2, sb));
        hipLaunchKernelGGL(expander, dim3(p.E), dim3(ETHREADS), 0, sb, p);
        CK(hipEventRecord(e3, sb));
        CK(hipStreamSynchronize(sa));
        CK(hipStreamSynchronize(sb));
        float ms_w2 = 0, ms_all = 0, ms_e2 = 0;
        CK(hipEventElapsedTime(&ms_w2, e0, e1));
        CK(hipEventElapsedTime(&ms_all, e0, e3));
        CK(hipEventElapsedTime(&ms_e2, e2, e3));
        CK(hipMemcpy(ts.data(), p.ts, ts.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(es.data(), p.es, es.size() * 8, hipMemcpyDeviceToHost));
        u64 first_done = ~0ull, last_done = 0, last_exp = 0;
        for (int t = 0; t < p.ntiles; ++t) {
            first_done = std::min(first_done, ts[3 * t]);
            last_done = std::max(last_done, ts[3 * t]);
            last_exp = std::max(last_exp, ts[3 * t + 2]);
        }
        std::vector<double> lag, start;
        u64 run_max = 0;  // a tile can be expanded once all earlier tiles are done
        for (int t = 0; t < p.ntiles; ++t) {
            run_max = std::max(run_max, ts[3 * t]);
            lag.push_back(((double)ts[3 * t + 1] - (double)run_max) / 100.0);
        }
        for (int g = 0; g < p.E; ++g) start.push_back(((double)es[g] - (double)first_done) / 100.0);
        std::sort(lag.begin(), lag.end());
        std::sort(start.begin(), start.end());
        printf("rep %d: walker alone %.3f ms | expander alone %.3f ms | together: walker %.3f, expander %.3f, all %.3f ms; "
               "expanders start %.1f .. %.1f .. %.1f us after the first tile is done (walk spans %.1f us); lag behind the "
               "frontier p50 %.1f p90 %.1f p99 %.1f max %.1f us; tail after the last tile %.1f us\n",
               rep, ms_walk, ms_exp, ms_w2, ms_e2, ms_all, start.front(), start[start.size() / 2], start.back(),
               ((double)last_done - (double)first_done) / 100.0, lag[lag.size() / 2], lag[lag.size() * 9 / 10],
               lag[lag.size() * 99 / 100], lag.back(), ((double)last_exp - (double)last_done) / 100.0);
    }
    return 0;
}
