// dev tool: which XCD does workgroup i run on?  (s_getreg_b32 HW_REG_XCC_ID, gfx940+)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}
int main() {
    const int n = 512;
    unsigned* d; (void)hipMalloc(&d, n * 4);
    for (int threads : {256, 512}) {
        hipLaunchKernelGGL(k, dim3(n), dim3(threads), 0, 0, d);
        (void)hipDeviceSynchronize();
        unsigned h[n]; (void)hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
        int ok = 0;
        for (int i = 0; i < n; ++i) ok += ((h[i] & 0xf) == (unsigned)(i % 8));
        printf("threads=%d: workgroups with XCC_ID == blockIdx %% 8: %d / %d;  first 24 ids:", threads, ok, n);
        for (int i = 0; i < 24; ++i) printf(" %u", h[i] & 0xf);
        printf("\n");
    }
    return 0;
}
