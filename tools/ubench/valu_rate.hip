// Micro-benchmark: issue rate of the VALU ops the minimizer kernel is made of (gfx950).
// Each kernel runs a long unrolled chain mix of ONE instruction on 8 independent registers.
// Round 3: the rate is reported in SHADER CYCLES, not at an assumed clock: lane 0 of every workgroup reads
// s_memtime (the shader clock counter) and s_memrealtime (100 MHz) around its loop; eight workgroups per CU are
// resident together (2048 in all), so a SIMD issues 8 x iters x REP wave-instructions during one workgroup's
// span.  The shader clock under each loop is printed beside it (d s_memtime / d s_memrealtime x 100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 256
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned seed, int iters, unsigned long long *ts) {
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    unsigned b = seed * 2654435761u + threadIdx.x, c = b ^ 0x55aa55aa;
    unsigned long long m64 = 0x5555aaaa5555aaaaull * seed;
    unsigned long long c0, r0, c1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
#define STEP(x)                                                                                   \
    if (OP == 0) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(x) : "v"(b));                         \
    if (OP == 1) asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(x));                          \
    if (OP == 2) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));            \
    if (OP == 3) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));          \
    if (OP == 4) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x26" : "+v"(x) : "v"(b), "v"(c)); \
    if (OP == 5) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x) : "v"(b));                         \
    if (OP == 6) asm volatile("v_min_u32 %0, %1, %0" : "+v"(x) : "v"(b));                         \
    if (OP == 7) asm volatile("v_bfe_u32 %0, %0, 3, 2" : "+v"(x));                                \
    if (OP == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b));                \
    if (OP == 9) asm volatile("v_cmp_lt_i32 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc");             \
    if (OP == 10) asm volatile("v_alignbit_b32 %0, %0, %0, %1" : "+v"(x) : "s"(seed));            \
    if (OP == 11) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "s"(seed));      \
    if (OP == 12) asm volatile("v_lshrrev_b32 %0, 5, %0" : "+v"(x));                              \
    if (OP == 13) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(b));                     \
    if (OP == 14) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(x));                             \
    if (OP == 15) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "s"(m64));      \
    if (OP == 16) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x) : "v"(b));                \
    if (OP == 17) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));        \
    if (OP == 18) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(x) : "s"(seed));                     \
    if (OP == 19) asm volatile("v_max_u32 %0, %1, %0" : "+v"(x) : "v"(b));                        \
    if (OP == 20) asm volatile("v_and_b32 %0, 0x78, %0" : "+v"(x));                               \
    if (OP == 21) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(x) : "v"(b));                 \
    if (OP == 22) asm volatile("v_min_i32 %0, %1, %0" : "+v"(x) : "v"(b));                        \
    if (OP == 23) asm volatile("v_min_f32 %0, %1, %0" : "+v"(x) : "v"(b));                        \
    if (OP == 24) asm volatile("v_pk_min_u16 %0, %1, %0" : "+v"(x) : "v"(b));                     \
    if (OP == 25) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));           \
    if (OP == 26) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));            \
    if (OP == 27) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));                \
    if (OP == 28) asm volatile("v_mov_b32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "+v"(x)); \
    if (OP == 29) asm volatile("v_and_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "+v"(x) : "v"(b)); \
    if (OP == 30) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(x) : "v"(b)); \
    if (OP == 31) asm volatile("v_cmp_ne_u32_sdwa vcc, %0, %1 src0_sel:WORD_0 src1_sel:WORD_0" : : "v"(x), "v"(b) : "vcc"); \
    if (OP == 32) asm volatile("v_bfe_i32 %0, %0, 4, 2" : "+v"(x));                               \
    if (OP == 33) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x) : "v"(b), "v"(c));            \
    if (OP == 34) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(x));                              \
    if (OP == 35) asm volatile("v_or_b32 %0, %1, %0" : "+v"(x) : "v"(b));                         \
    if (OP == 36) asm volatile("v_add_u32_sdwa %0, %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(x) : "v"(b)); \
    if (OP == 37) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(b));                        \
    if (OP == 38) asm volatile("v_subrev_u32 %0, %1, %0" : "+v"(x) : "v"(b));                     \
    if (OP == 39) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xea" : "+v"(x) : "v"(b), "s"(seed)); \
    if (OP == 40) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x) : "s"(seed));                     \
    if (OP == 41) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(x) : "s"(seed));                     \
    if (OP == 42) asm volatile("v_add_u32 %0, 0x204, %0" : "+v"(x));                              \
    if (OP == 43) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(x) : "s"(seed));                 \
    if (OP == 44) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x) : "s"(seed));                     \
    if (OP == 45) asm volatile("v_bitop3_b32 %0, %1, %0, %2 bitop3:0xea" : "+v"(x) : "s"(seed), "v"(b)); \
    if (OP == 46) asm volatile("v_bitop3_b32 %0, %0, %1, 21 bitop3:0xea" : "+v"(x) : "v"(b));     \
    if (OP == 47) asm volatile("v_pack_b32_f16 %0, %1, %0 op_sel:[0,1]" : "+v"(x) : "v"(b));     \
    if (OP == 48) asm volatile("v_pack_b32_f16 %0, 21, %0 op_sel:[0,1]" : "+v"(x));              \
    if (OP == 49) asm volatile("v_and_or_b32 %0, %0, %1, 21" : "+v"(x) : "v"(b));                \
    if (OP == 50) asm volatile("v_readlane_b32 s20, %0, 3\n\tv_xor_b32 %0, %1, %0" : "+v"(x) : "v"(b) : "s20"); \
    if (OP == 51) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(b));
            STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1));
    if (threadIdx.x == 0 && ts) {
        ts[2 * blockIdx.x] = c1 - c0;
        ts[2 * blockIdx.x + 1] = r1 - r0;
    }
}

static unsigned long long *g_ts;

template <int OP>
void run(const char *name, unsigned *d) {
    const int blocks = 256 * 8, iters = 200;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u, 2, (unsigned long long *)nullptr);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 7u, iters, g_ts);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), g_ts, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (int b = 0; b < blocks; ++b) { cyc += (double)h[2 * b]; real += (double)h[2 * b + 1]; }
    cyc /= blocks; real /= blocks;
    const double per_simd = 8.0 * iters * REP;  // wave-instructions a SIMD issues during one workgroup's span
    const double ghz = cyc / real * 0.1;          // s_memrealtime ticks at 100 MHz
    double winst = (double)blocks * 4 * iters * REP;
    printf("%-28s %8.3f ms  %6.3f shader cycles per wave-instr  (shader clock %.3f GHz; by wall time at that clock: %.3f)\n",
           name, ms, cyc / per_simd, ghz, (ms * 1e-3) * ghz * 1e9 * 1024.0 / winst);
}

int main() {
    unsigned *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    hipMalloc(&g_ts, 256 * 8 * 16);
    run<0>("v_xor_b32", d); run<5>("v_add_u32", d); run<6>("v_min_u32", d); run<12>("v_lshrrev_b32 imm", d);
    run<7>("v_bfe_u32 imm", d); run<1>("v_alignbit_b32 imm", d); run<10>("v_alignbit_b32 sgpr", d);
    run<2>("v_min3_u32", d); run<3>("v_and_or_b32 vvv", d); run<11>("v_and_or_b32 vvs", d);
    run<4>("v_bitop3_b32", d); run<8>("v_cndmask_b32 vcc", d); run<9>("v_cmp_lt_i32", d); run<13>("v_mul_lo_u32", d);
    run<14>("v_ashrrev_i32", d); run<15>("v_cndmask_b32 sgpr-mask", d); run<16>("v_lshl_add_u32", d); run<17>("v_mad_u32_u24", d);
    run<18>("v_sub_u32", d); run<19>("v_max_u32", d); run<20>("v_and_b32 lit", d); run<21>("v_lshl_or_b32", d); run<22>("v_min_i32", d);
    run<23>("v_min_f32", d); run<24>("v_pk_min_u16", d); run<25>("v_add3_u32", d); run<26>("v_xad_u32", d); run<27>("v_perm_b32", d);
    run<28>("v_mov_b32_sdwa byte", d); run<29>("v_and_b32_sdwa byte", d); run<30>("v_add_u32_sdwa byte", d);
    run<31>("v_cmp_ne_u32_sdwa", d); run<32>("v_bfe_i32", d); run<33>("v_bfi_b32", d); run<34>("v_lshlrev_b32", d);
    run<35>("v_or_b32", d); run<36>("v_add_u32_sdwa sext byte", d); run<37>("v_sub_u32 vv", d); run<38>("v_subrev_u32", d);
    run<39>("v_bitop3_b32 vvs", d); run<40>("v_add_u32 sv", d); run<41>("v_xor_b32 sv", d); run<42>("v_add_u32 literal", d);
    run<43>("v_lshrrev_b32 sv", d); run<44>("v_and_b32 sv", d); run<45>("v_bitop3_b32 svv", d);
    run<46>("v_bitop3_b32 vv inline", d); run<47>("v_pack_b32_f16 vv op_sel", d); run<48>("v_pack_b32_f16 inline,v", d);
    run<49>("v_and_or_b32 vv inline", d); run<50>("v_readlane + v_xor (pair)", d); run<51>("v_mov_b32 vv", d);
    return 0;
}
