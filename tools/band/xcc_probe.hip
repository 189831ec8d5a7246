#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(512) k(int* out) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(x & 0xf);
    // stay resident a little so that all 256 are co-resident
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
}
int main() {
    int* d; hipMalloc(&d, 4 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 50 * 1024, 0, d);
        hipDeviceSynchronize();
        int h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
        for (int b = 0; b < 32; ++b) printf("%d", h[b]); printf(" ... ");
        int ok = 0; for (int b = 0; b < 256; ++b) ok += (h[b] == h[b % 8]);
        printf("blocks whose XCC equals that of block b%%8: %d / 256\n", ok);
    }
}
