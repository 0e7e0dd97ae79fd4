// Where does a direct-to-LDS buffer load (buffer_load_dword / _dwordx4 ... lds, gfx950) put each lane's data?
// Every lane loads src[lane-specific offset]; the wave's M0 base is wave-uniform.  Prints the layout the hardware used:
// for each lane, the LDS dword index (relative to the wave's base) at which its first dword landed.
//   hipcc --offload-arch=gfx950 -O3 -o lds_dma.bin lds_dma.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int SIZE>
__global__ void k(const uint32_t *src, uint32_t *dump, int n_dwords) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(src), 0, n_dwords * 4, 0x00020000);
    const uint32_t wave = threadIdx.x / 64, lane = threadIdx.x & 63;
    for (uint32_t i = threadIdx.x; i < 4096 / 4 * 4; i += blockDim.x) reinterpret_cast<uint32_t *>(smem)[i] = 0xdeadbeefu;
    __syncthreads();
    const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>(smem) + wave * 4096u);
    // lane L loads SIZE bytes from byte offset 64 * L (so that every lane's data is recognisable: src[i] = i)
    if (SIZE == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)(uintptr_t)base, 4, lane * 64u + wave * 8192u, 0, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)(uintptr_t)base, 16, lane * 64u + wave * 8192u, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (uint32_t i = lane; i < 1024; i += 64) dump[wave * 1024 + i] = reinterpret_cast<const uint32_t *>(smem + wave * 4096u)[i];
}
int main() {
    const int n = 1 << 16;
    std::vector<uint32_t> h(n);
    for (int i = 0; i < n; ++i) h[i] = i;
    uint32_t *d_src, *d_dump;
    hipMalloc(&d_src, n * 4); hipMalloc(&d_dump, 4 * 1024 * 4);
    hipMemcpy(d_src, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int size : {4, 16}) {
        if (size == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(256), 16384, 0, d_src, d_dump, n);
        else hipLaunchKernelGGL(k<16>, dim3(1), dim3(256), 16384, 0, d_src, d_dump, n);
        std::vector<uint32_t> out(4096);
        hipMemcpy(out.data(), d_dump, 4096 * 4, hipMemcpyDeviceToHost);
        printf("size %d bytes per lane (%s):\n", size, hipGetErrorString(hipGetLastError()));
        for (int wave = 0; wave < 2; ++wave) {
            printf("  wave %d: ", wave);
            for (int lane : {0, 1, 2, 3, 31, 32, 63}) {
                const uint32_t want = (lane * 64u + wave * 8192u) / 4;  // the lane's first dword value
                int at = -1;
                for (int i = 0; i < 1024; ++i) if (out[wave * 1024 + i] == want) { at = i; break; }
                printf("lane %d -> dword %d%s; ", lane, at, (size == 16 && at >= 0 && out[wave * 1024 + at + 1] == want + 1 && out[wave * 1024 + at + 3] == want + 3) ? " (+3 contiguous)" : "");
            }
            int written = 0; for (int i = 0; i < 1024; ++i) written += out[wave * 1024 + i] != 0xdeadbeefu;
            printf("%d dwords written\n", written);
        }
    }
    return 0;
}
