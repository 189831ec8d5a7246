// hostreg_verify.hip — does an H2D copy out of a chunk-wise hipHostRegister'ed anonymous mapping deliver the bytes the CPU
// wrote BEFORE the registration?  (adjacent chunks registered separately, several mappings, copies issued from a second thread)
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static int run(int n_maps, bool huge, bool threaded) {
    const size_t chunk = (size_t)64 << 20, len = 2 * chunk, used = 3 << 20;  // 3 MiB of data at the start of each mapping
    std::vector<char*> maps;
    for (int i = 0; i < n_maps; ++i) {
        char* m = (char*)mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) return 1;
        if (huge) madvise(m, len, MADV_HUGEPAGE);
        for (size_t k = 0; k < used; ++k) m[k] = (char)(k * 7 + i * 31 + 1);
        maps.push_back(m);
    }
    int bad = 0;
    auto work = [&]() -> int {
        (void)hipSetDevice(0);
        hipStream_t st;
        CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        void* d;
        CK(hipMalloc(&d, used));
        std::vector<char> back(used);
        for (int i = 0; i < n_maps; ++i) {
            CK(hipHostRegister(maps[i], chunk, hipHostRegisterPortable));
            CK(hipMemcpyAsync(d, maps[i], used, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(back.data(), d, used, hipMemcpyDeviceToHost));
            size_t wrong = 0;
            for (size_t k = 0; k < used; ++k) wrong += back[k] != (char)(k * 7 + i * 31 + 1);
            if (wrong) printf("  mapping %d (%p): %zu of %zu bytes wrong, first bytes %d %d %d\n", i, (void*)maps[i], wrong, used, back[0], back[1], back[2]), ++bad;
        }
        return 0;
    };
    if (threaded) { std::thread t(work); t.join(); } else work();
    printf("n_maps=%d huge=%d threaded=%d: %s\n", n_maps, (int)huge, (int)threaded, bad ? "WRONG DATA" : "ok");
    for (char* m : maps) { (void)hipHostUnregister(m); munmap(m, len); }
    return bad;
}
int main() {
    int bad = 0;
    for (int huge = 0; huge < 2; ++huge)
        for (int th = 0; th < 2; ++th) bad += run(3, huge, th);
    return bad ? 1 : 0;
}
