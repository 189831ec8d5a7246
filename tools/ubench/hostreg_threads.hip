// hostreg_threads.hip — does hipHostRegister of 64 MiB chunks of an untouched huge-page mapping scale with threads? (round 6:
// the staging slabs are page-locked ahead of the frame loop; one thread registers 23 GB/s)
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

int main() {
    const size_t chunk = (size_t)64 << 20, n = 64;  // 4 GB
    hipSetDevice(0);
    hipFree(nullptr);
    for (int nt : {1, 2, 3, 4, 8}) {
        char* m = (char*)mmap(nullptr, chunk * n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        madvise(m, chunk * n, MADV_HUGEPAGE);
        std::atomic<size_t> next{0};
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t)
            th.emplace_back([&] {
                hipSetDevice(0);
                for (;;) {
                    const size_t k = next.fetch_add(1);
                    if (k >= n) break;
                    if (hipHostRegister(m + k * chunk, chunk, hipHostRegisterPortable) != hipSuccess) printf("register failed\n");
                }
            });
        for (auto& t : th) t.join();
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%d thread(s): %zu x 64 MiB registered in %.3f s = %.1f GB/s\n", nt, n, s, chunk * n / s / 1e9);
        for (size_t k = 0; k < n; ++k) hipHostUnregister(m + k * chunk);
        munmap(m, chunk * n);
    }
    return 0;
}
