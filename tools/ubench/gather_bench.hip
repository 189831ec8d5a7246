// gather_bench.hip — microbenchmark of the strided column gather (diagnostic tool).
// Each workgroup reads, for its column group, W bytes per row from T rows (row stride = ld bytes);
// lanes take consecutive rows.  Reports useful GB/s and lane-requests/clk/CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int NT, int UNROLL, int WIDTH>  // WIDTH = doubles per lane per row (2, 4, 8, 16)
__global__ void __launch_bounds__(NT) k_gather(const double* __restrict__ base, long ld, int T,
                                               long n_groups, double* __restrict__ out) {
    double acc = 0.0;
    const int tid = threadIdx.x;
    // XCD-aware: blocks sharing an XCD take adjacent column groups
    const int nwg = gridDim.x;
    long slot = blockIdx.x;
    if (nwg % 8 == 0) slot = (long)(blockIdx.x % 8) * (nwg / 8) + blockIdx.x / 8;
    for (long g = slot; g < n_groups; g += nwg) {
        const double* col = base + g * WIDTH;
        for (int t0 = 0; t0 < T; t0 += NT * UNROLL) {
            double2 v[UNROLL][WIDTH / 2];
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) {
                int t = t0 + tid + k * NT;
                t = t < T ? t : T - 1;
                const double2* p = reinterpret_cast<const double2*>(col + (long)t * ld);
#pragma unroll
                for (int w = 0; w < WIDTH / 2; ++w) v[k][w] = p[w];
            }
#pragma unroll
            for (int k = 0; k < UNROLL; ++k)
#pragma unroll
                for (int w = 0; w < WIDTH / 2; ++w) acc += v[k][w].x + v[k][w].y;
        }
    }
    out[(long)blockIdx.x * NT + tid] = acc;
}

// Same with non-temporal loads (nt cache policy).
template <int NT, int UNROLL, int WIDTH>  // WIDTH = doubles per lane per row (2, 4, 8, 16)
__global__ void __launch_bounds__(NT) k_gather_nt(const double* __restrict__ base, long ld, int T,
                                               long n_groups, double* __restrict__ out) {
    double acc = 0.0;
    const int tid = threadIdx.x;
    // XCD-aware: blocks sharing an XCD take adjacent column groups
    const int nwg = gridDim.x;
    long slot = blockIdx.x;
    if (nwg % 8 == 0) slot = (long)(blockIdx.x % 8) * (nwg / 8) + blockIdx.x / 8;
    for (long g = slot; g < n_groups; g += nwg) {
        const double* col = base + g * WIDTH;
        for (int t0 = 0; t0 < T; t0 += NT * UNROLL) {
            double2 v[UNROLL][WIDTH / 2];
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) {
                int t = t0 + tid + k * NT;
                t = t < T ? t : T - 1;
                const double2* p = reinterpret_cast<const double2*>(col + (long)t * ld);
#pragma unroll
                for (int w = 0; w < WIDTH / 2; ++w) { typedef double d2v __attribute__((ext_vector_type(2))); const d2v t = __builtin_nontemporal_load(reinterpret_cast<const d2v*>(&p[w])); v[k][w].x = t.x; v[k][w].y = t.y; }
            }
#pragma unroll
            for (int k = 0; k < UNROLL; ++k)
#pragma unroll
                for (int w = 0; w < WIDTH / 2; ++w) acc += v[k][w].x + v[k][w].y;
        }
    }
    out[(long)blockIdx.x * NT + tid] = acc;
}

// Same gather with a soft grid-wide barrier every `every` column groups: all workgroups then
// fetch the same rows at about the same time (DRAM row-buffer locality across workgroups).
// The spin has an iteration cap, so a missing workgroup can delay but never hang the grid.
template <int NT, int UNROLL, int WIDTH>
__global__ void __launch_bounds__(NT) k_gather_sync(const double* __restrict__ base, long ld, int T,
                                                    long n_groups, double* __restrict__ out,
                                                    unsigned* __restrict__ counter, int every) {
    double acc = 0.0;
    const int tid = threadIdx.x;
    const int nwg = gridDim.x;
    long slot = blockIdx.x;
    if (nwg % 8 == 0) slot = (long)(blockIdx.x % 8) * (nwg / 8) + blockIdx.x / 8;
    unsigned round = 0;
    for (long g = slot; g < n_groups; g += nwg) {
        if (every > 0 && (round % every) == 0) {
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned want = (round / every + 1) * (unsigned)nwg;
                for (int spin = 0; spin < 200000; ++spin) {
                    if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) break;
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            __syncthreads();
        }
        ++round;
        const double* col = base + g * WIDTH;
        for (int t0 = 0; t0 < T; t0 += NT * UNROLL) {
            double2 v[UNROLL][WIDTH / 2];
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) {
                int t = t0 + tid + k * NT;
                t = t < T ? t : T - 1;
                const double2* p = reinterpret_cast<const double2*>(col + (long)t * ld);
#pragma unroll
                for (int w = 0; w < WIDTH / 2; ++w) v[k][w] = p[w];
            }
#pragma unroll
            for (int k = 0; k < UNROLL; ++k)
#pragma unroll
                for (int w = 0; w < WIDTH / 2; ++w) acc += v[k][w].x + v[k][w].y;
        }
    }
    out[(long)blockIdx.x * NT + tid] = acc;
}

template <int NT, int UNROLL, int WIDTH>
void run_sync(const double* d, long ld, int T, long C, int wg_per_cu, double* out, int every) {
    const long n_groups = C / WIDTH;
    int nwg = 256 * wg_per_cu;
    unsigned* counter; CK(hipMalloc(&counter, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        CK(hipMemset(counter, 0, 64));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_gather_sync<NT, UNROLL, WIDTH>), dim3(nwg), dim3(NT), 0, 0, d, ld, T, n_groups, out, counter, every);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double bytes = (double)T * C * 8;
    printf("SYNC every=%d NT=%4d unroll=%2d width=%3dB wg/cu=%d : %8.3f ms  %7.1f GB/s useful\n", every, NT, UNROLL,
           WIDTH * 8, wg_per_cu, best, bytes / best / 1e6);
    CK(hipFree(counter));
}

template <int NT, int UNROLL, int WIDTH>
void run(const double* d, long ld, int T, long C, int wg_per_cu, double* out) {
    const long n_groups = C / WIDTH;
    int nwg = 256 * wg_per_cu;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_gather<NT, UNROLL, WIDTH>), dim3(nwg), dim3(NT), 0, 0, d, ld, T, n_groups, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double bytes = (double)T * C * 8;
    const double reqs = (double)T * n_groups;  // lane-row requests
    printf("NT=%4d unroll=%2d width=%3dB wg/cu=%d : %8.3f ms  %7.1f GB/s useful  %.3f lane-req/clk/CU (2.1GHz)\n",
           NT, UNROLL, WIDTH * 8, wg_per_cu, best, bytes / best / 1e6, reqs / (best * 1e-3 * 2.1e9 * 256));
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 10000;
    const long A = argc > 2 ? atol(argv[2]) : 20000;
    const long C = A * 3;
    double* d; CK(hipMalloc(&d, sizeof(double) * T * C));
    CK(hipMemset(d, 0, sizeof(double) * T * C));
    double* out; CK(hipMalloc(&out, sizeof(double) * 256 * 8 * 1024));
    printf("T=%d C=%ld (%.2f GB)\n", T, C, T * C * 8 / 1e9);
    {
        const long n_groups = C / 2;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e9;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((k_gather_nt<256, 20, 2>), dim3(256), dim3(256), 0, 0, d, C, T, n_groups, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("NT loads NT=256 unroll=20 width=16B : %8.3f ms  %7.1f GB/s useful\n", best, (double)T * C * 8 / best / 1e6);
    }
    run_sync<256, 20, 2>(d, C, T, C, 1, out, 0);
    run_sync<256, 20, 2>(d, C, T, C, 1, out, 1);
    run_sync<256, 20, 2>(d, C, T, C, 1, out, 4);
    run_sync<256, 20, 2>(d, C, T, C, 1, out, 16);
    run_sync<512, 20, 2>(d, C, T, C, 1, out, 1);
    run_sync<1024, 10, 2>(d, C, T, C, 1, out, 1);
    run<256, 8, 2>(d, C, T, C, 1, out);
    run<256, 20, 2>(d, C, T, C, 1, out);
    run<256, 40, 2>(d, C, T, C, 1, out);
    run<512, 20, 2>(d, C, T, C, 1, out);
    run<1024, 10, 2>(d, C, T, C, 1, out);
    run<256, 10, 2>(d, C, T, C, 4, out);
    run<256, 10, 2>(d, C, T, C, 8, out);
    run<1024, 10, 2>(d, C, T, C, 2, out);
    run<256, 20, 4>(d, C, T, C, 1, out);
    run<256, 10, 4>(d, C, T, C, 4, out);
    run<256, 10, 8>(d, C, T, C, 4, out);
    run<256, 5, 16>(d, C, T, C, 4, out);
    run<1024, 10, 4>(d, C, T, C, 2, out);
    run<1024, 5, 8>(d, C, T, C, 2, out);
    run<1024, 2, 16>(d, C, T, C, 2, out);
    return 0;
}
