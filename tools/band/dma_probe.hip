// dma_probe.hip — what buffer_load_dwordx4 ... lds does on gfx950: where lane L's 16 bytes land, what
// out-of-range lanes write, and the result layout of v_mfma_f32_16x16x4_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_probe(const float* src, unsigned n_bytes, float* out, float* mf) {
    __shared__ float lds[2048];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2048; i += 64) lds[i] = -7.0f;
    __syncthreads();
    const unsigned long long base = (unsigned long long)src;
    const u4 rs = u4{(unsigned)base, (unsigned)(base >> 32) & 0xffffu, n_bytes, 0x00020000u};
    const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lds + 256));  // byte offset 1024 into lds
    // lane L fetches bytes [32 L, 32 L + 16): lanes with 32 L + 16 > n_bytes are out of range
    unsigned voff = 32u * lane;
    if (lane == 5) voff = 0x80000000u;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_waitcnt vmcnt(0)" ::"s"(lds_addr), "v"(voff), "s"(rs)
                 : "memory", "m0");
    __syncthreads();
    for (int i = lane; i < 2048; i += 64) out[i] = lds[i];
    // MFMA layout: A[i][k] = 100 i + k... C = A B with B = identity-like selector: B[k][j] = (j == k) -> C[i][j] = A[i][j] for j < 4
    const float a = 100.0f * (lane & 15) + (lane >> 4);       // A[i = l & 15][k = l >> 4]
    const float b = ((lane & 15) == (lane >> 4)) ? 1.0f : 0.0f;  // B[k = l >> 4][j = l & 15]
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) mf[lane * 4 + r] = c[r];
}

int main() {
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    float *d, *o, *m;
    CK(hipMalloc(&d, 4096 * 4));
    CK(hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&o, 2048 * 4));
    CK(hipMalloc(&m, 256 * 4));
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d, 32u * 40 + 16, o, m);  // lanes 0..40 in range
    CK(hipDeviceSynchronize());
    std::vector<float> r(2048), mf(256);
    CK(hipMemcpy(r.data(), o, 2048 * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(mf.data(), m, 256 * 4, hipMemcpyDeviceToHost));
    printf("LDS words that changed (index relative to the M0 base = word 256), value:\n");
    for (int i = 0; i < 2048; ++i)
        if (r[i] != -7.0f) printf("  [%d] = %.0f%s", i - 256, r[i], (i % 4 == 3) ? "\n" : "");
    printf("\nMFMA f32 16x16x4: lane l register r holds C[row][col]; A[i][k] = 100 i + k, C[i][j] = A[i][j] (j < 4)\n");
    for (int l : {0, 1, 2, 3, 16, 17, 32, 48, 49})
        printf("  lane %2d: %.0f %.0f %.0f %.0f\n", l, mf[l * 4], mf[l * 4 + 1], mf[l * 4 + 2], mf[l * 4 + 3]);
    return 0;
}
