// Do raw buffer loads of 16 bytes work at any BYTE offset on gfx950, and is the range check per dword?
// (the walk's wide loads rely on both: round 3)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const uint8_t *src, int nbytes, u32x4 *out) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src), 0, nbytes, 0x00020000);
    out[threadIdx.x] = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x, 0, 0);  // byte offset = thread id
}
int main() {
    const int n = 100;  // bytes in range; threads 0..127 load 16 bytes from offsets 0..127
    std::vector<uint8_t> h(256);
    for (int i = 0; i < 256; ++i) h[i] = (uint8_t)(i * 7 + 3);
    uint8_t *d; u32x4 *o;
    hipMalloc(&d, 256); hipMalloc(&o, 128 * 16);
    hipMemcpy(d, h.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, d, n, o);
    std::vector<uint8_t> g(128 * 16);
    hipMemcpy(g.data(), o, g.size(), hipMemcpyDeviceToHost);
    int bad_bytewise = 0, bad_dwordwise = 0, bad_wholeload = 0;
    for (int t = 0; t < 128; ++t)
        for (int j = 0; j < 16; ++j) {
            const int a = t + j;
            const uint8_t got = g[t * 16 + j];
            // three candidate rules for bytes past num_records
            const uint8_t byte_rule = a < n ? h[a] : 0;
            const int dw0 = t + (j / 4) * 4;  // first byte of the dword this byte belongs to
            const uint8_t dword_rule = (dw0 + 4 <= n) ? h[a] : 0;
            const uint8_t whole_rule = (t + 16 <= n) ? h[a] : 0;
            bad_bytewise += got != byte_rule;
            bad_dwordwise += got != dword_rule;
            bad_wholeload += got != whole_rule;
        }
    printf("unaligned 16-byte raw buffer loads: mismatches against the per-byte rule %d, per-dword rule %d, whole-load rule %d\n",
           bad_bytewise, bad_dwordwise, bad_wholeload);
    for (int t : {0, 1, 2, 3, 5, 83, 84, 85, 86, 90, 97, 99, 100}) {
        printf("offset %3d:", t);
        for (int j = 0; j < 16; ++j) printf(" %02x%s", g[t * 16 + j], (t + j < 256 && g[t * 16 + j] == h[t + j]) ? "" : "*");
        printf("\n");
    }
    return 0;
}
