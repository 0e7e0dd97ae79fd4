// Host <-> device copy rates on the box: pageable vs pinned vs registered-in-place, and the cost of registering.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t n = 512ull << 20;
    void *d; hipMalloc(&d, n);
    char *pageable = (char *)aligned_alloc(4096, n); memset(pageable, 1, n);
    char *pinned; hipHostMalloc((void **)&pinned, n, hipHostMallocDefault); memset(pinned, 1, n);
    hipStream_t s; hipStreamCreate(&s);
    auto rate = [&](const char *what, void *dst, const void *src, hipMemcpyKind k) {
        hipMemcpyAsync(dst, src, n, k, s); hipStreamSynchronize(s);
        double t0 = now();
        for (int i = 0; i < 3; ++i) hipMemcpyAsync(dst, src, n, k, s);
        hipStreamSynchronize(s);
        printf("%-28s %6.1f GB/s\n", what, 3.0 * n / (now() - t0) / 1e9);
    };
    rate("H2D pageable", d, pageable, hipMemcpyHostToDevice);
    rate("D2H pageable", pageable, d, hipMemcpyDeviceToHost);
    rate("H2D pinned", d, pinned, hipMemcpyHostToDevice);
    rate("D2H pinned", pinned, d, hipMemcpyDeviceToHost);
    double t0 = now();
    hipError_t e = hipHostRegister(pageable, n, hipHostRegisterDefault);
    double tr = now() - t0;
    printf("hipHostRegister 512 MiB: %s in %.1f ms (%.1f GB/s)\n", hipGetErrorString(e), tr * 1e3, n / tr / 1e9);
    if (e == hipSuccess) {
        rate("H2D registered", d, pageable, hipMemcpyHostToDevice);
        rate("D2H registered", pageable, d, hipMemcpyDeviceToHost);
        t0 = now(); hipHostUnregister(pageable); printf("hipHostUnregister: %.1f ms\n", (now() - t0) * 1e3);
    }
    // both directions at once on two streams
    {
        void *d2; hipMalloc(&d2, n);
        char *pinned2; hipHostMalloc((void **)&pinned2, n, hipHostMallocDefault); memset(pinned2, 2, n);
        char *pageable2 = (char *)aligned_alloc(4096, n); memset(pageable2, 3, n);
        hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
        hipStream_t s1; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
        auto both = [&](const char *what, void *hsrc, void *hdst) {
            hipMemcpyAsync(d, hsrc, n, hipMemcpyHostToDevice, s1); hipMemcpyAsync(hdst, d2, n, hipMemcpyDeviceToHost, s2);
            hipStreamSynchronize(s1); hipStreamSynchronize(s2);
            double t0 = now();
            for (int i = 0; i < 3; ++i) {
                hipMemcpyAsync(d, hsrc, n, hipMemcpyHostToDevice, s1);
                hipMemcpyAsync(hdst, d2, n, hipMemcpyDeviceToHost, s2);
            }
            hipStreamSynchronize(s1); hipStreamSynchronize(s2);
            printf("%-28s %6.1f GB/s aggregate\n", what, 6.0 * n / (now() - t0) / 1e9);
        };
        both("H2D + D2H pinned", pinned, pinned2);
        both("H2D + D2H pageable", pageable, pageable2);
    }
    t0 = now(); memcpy(pinned, pageable, n); printf("host memcpy 1 thread: %.1f GB/s\n", n / (now() - t0) / 1e9);
    return 0;
}
