// Which CU / XCD does workgroup b land on, and when?  (dispatch order of a 256-thread, 39 KB-LDS grid)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned long long *out, int spin) {
    extern __shared__ unsigned char smem[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long rt = wall_clock64();
    // busy work so that several rounds of workgroups exist
    unsigned x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1664525u + 1013904223u;
    smem[threadIdx.x] = (unsigned char)x;
    if (threadIdx.x == 0) {
        out[3 * blockIdx.x + 0] = ((unsigned long long)xcc << 32) | hw;
        out[3 * blockIdx.x + 1] = rt;
        out[3 * blockIdx.x + 2] = t0 + (x & 1);
    }
}
int main() {
    const int blocks = 4096;
    unsigned long long *d; hipMalloc(&d, blocks * 24);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 39408, 0, d, 20000);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 3);
    hipMemcpy(h.data(), d, blocks * 24, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull;
    for (int b = 0; b < blocks; ++b) if (h[3 * b + 1] < tmin) tmin = h[3 * b + 1];
    for (int b = 0; b < blocks; ++b) {
        unsigned hw = (unsigned)h[3 * b], xcc = (unsigned)(h[3 * b] >> 32);
        // gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
        if (b < 48 || (b % 256) < 4 || b % 509 == 0)
            printf("b=%4d xcc=%u se=%u sh=%u cu=%2u simd=%u wave=%u t=%llu\n", b, xcc & 15, (hw >> 13) & 7, (hw >> 12) & 1,
                   (hw >> 8) & 15, (hw >> 4) & 3, hw & 15, (unsigned long long)(h[3 * b + 1] - tmin));
    }
    return 0;
}
