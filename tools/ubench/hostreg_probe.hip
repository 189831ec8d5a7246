// hostreg_probe.hip — page-locking a big host buffer: hipHostMalloc against mmap + MADV_HUGEPAGE + hipHostRegister in
// chunks, and the H2D / D2H rates out of / into each.   hipcc -O2 --offload-arch=gfx950 hostreg_probe.hip -o hostreg_probe
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t n = (size_t)6 << 30, chunk = (size_t)256 << 20;
    void* d;
    CK(hipMalloc(&d, (size_t)2 << 30));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (int rep = 0; rep < 2; ++rep) {
        void* h;
        double t0 = now();
        CK(hipHostMalloc(&h, n, hipHostMallocDefault));
        printf("hipHostMalloc 6 GiB: %.0f ms\n", (now() - t0) * 1e3);
        t0 = now();
        CK(hipMemcpyAsync(d, h, (size_t)2 << 30, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        printf("  H2D 2 GiB from it: %.1f GB/s\n", 2.147 / (now() - t0));
        t0 = now();
        CK(hipHostFree(h));
        printf("  hipHostFree: %.0f ms\n", (now() - t0) * 1e3);
    }
    for (int huge = 1; huge >= 0; --huge) {
        double t0 = now();
        char* m = (char*)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) { printf("mmap failed\n"); return 1; }
        if (huge) madvise(m, n, MADV_HUGEPAGE);
        printf("mmap 6 GiB%s: %.1f ms\n", huge ? " + MADV_HUGEPAGE" : "", (now() - t0) * 1e3);
        t0 = now();
        for (size_t off = 0; off < n; off += chunk) CK(hipHostRegister(m + off, chunk, hipHostRegisterDefault));
        printf("  hipHostRegister in 256 MiB chunks: %.0f ms (%.1f GB/s)\n", (now() - t0) * 1e3, 6.44 / (now() - t0));
        bool zero = true;
        for (size_t k = 0; k < n; k += 4096) zero &= m[k] == 0;
        printf("  zero-filled: %d\n", (int)zero);
        for (int r = 0; r < 2; ++r) {
            t0 = now();
            // (a copy that spans two separately registered chunks is refused: "invalid argument")
            for (int c = 0; c < 8; ++c) CK(hipMemcpyAsync((char*)d + c * chunk, m + (c + 3) * chunk, chunk, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            printf("  H2D 2 GiB from it (8 chunk-sized copies): %.1f GB/s\n", 2.147 / (now() - t0));
        }
        t0 = now();
        for (int c = 0; c < 8; ++c) CK(hipMemcpyAsync(m + (c + 3) * chunk, (char*)d + c * chunk, chunk, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        printf("  D2H 2 GiB into it: %.1f GB/s\n", 2.147 / (now() - t0));
        t0 = now();
        for (size_t off = 0; off < n; off += chunk) CK(hipHostUnregister(m + off));
        munmap(m, n);
        printf("  unregister + munmap: %.0f ms\n", (now() - t0) * 1e3);
    }
    return 0;
}
