// Cost of the walk's per-lane sequence loads: 64 lanes `stride` bytes apart, 8 or 16 bytes per lane, at dword-aligned
// or odd byte offsets; every lane advances `adv` bytes per load (L1/L2-resident working set, like the walk's).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <int BYTES>
__global__ __launch_bounds__(256) void k(const uint8_t *src, uint32_t span, int stride, int adv, int mis, int iters, uint32_t *out) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src + (size_t)blockIdx.x * span), 0, (int)span, 0x00020000);
    uint32_t off = threadIdx.x * stride + mis, acc = 0;
    for (int i = 0; i < iters; i += 4) {
        if (BYTES == 16) {
            u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
            u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(r, off + adv, 0, 0);
            u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(r, off + 2 * adv, 0, 0);
            u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(r, off + 3 * adv, 0, 0);
            acc += a.x + a.w + b.x + b.w + c.x + c.w + d.x + d.w;
        } else {
            u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
            u32x2 b = __builtin_amdgcn_raw_buffer_load_b64(r, off + adv, 0, 0);
            u32x2 c = __builtin_amdgcn_raw_buffer_load_b64(r, off + 2 * adv, 0, 0);
            u32x2 d = __builtin_amdgcn_raw_buffer_load_b64(r, off + 3 * adv, 0, 0);
            acc += a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
        }
        off += 4 * adv;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
    const int blocks = 256 * 6, iters = 2048;
    const uint32_t span = 256 * 512 + 2048 * 16 + 64;
    uint8_t *d; uint32_t *o;
    hipMalloc(&d, (size_t)blocks * span); hipMalloc(&o, blocks * 256 * 4);
    hipMemset(d, 1, (size_t)blocks * span);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, int bytes, int stride, int adv, int mis) {
        float best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            if (bytes == 16) hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(256), 0, 0, d, span, stride, adv, mis, iters, o);
            else hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), 0, 0, d, span, stride, adv, mis, iters, o);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double winst = (double)blocks * 4 * iters;  // wave-level load instructions
        printf("%-52s %7.3f ms  %6.1f ns per wave-load per CU  (%5.1f clk at 2.1 GHz)\n", name, best, best * 1e6 / (winst / 256.0),
               best * 1e6 / (winst / 256.0) * 2.1);
    };
    for (int stride : {60, 64, 76, 96, 128, 192, 256, 320, 352, 356, 368, 372, 380, 384, 400, 448, 512}) {
        char name[96];
        snprintf(name, sizeof name, "16 B aligned, lanes %d B apart, +12 B per load", stride);
        run(name, 16, stride, 12, 0);
    }
    return 0;
}
