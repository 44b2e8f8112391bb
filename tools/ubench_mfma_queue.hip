// dev tool: does a wave stall at MFMA issue while the matrix pipe is busy, or are MFMAs queued?
// Times the ISSUE of N independent v_mfma_f32_32x32x16_bf16 (s_memtime right after the last one), and the same followed
// by a dependent VALU read after enough s_nops.   hipcc -O3 --offload-arch=gfx950 -o /tmp/q tools/ubench_mfma_queue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
template <int N>
__global__ __launch_bounds__(64) void k(unsigned long long* out, float* sink) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x4 a = {1e-3f, 2e-3f, 3e-3f, 4e-3f}, b = {1e-3f, 1e-3f, 2e-3f, 2e-3f};
    unsigned long long t0, t1, t2;
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll
    for (int i = 0; i < N; ++i) MFMA(acc[i], a, b);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    sink[threadIdx.x] = s;
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t1; }
}
template <int N> void run() {
    unsigned long long* d; float* s; hipMalloc(&d, 16); hipMalloc(&s, 256);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, d, s); hipDeviceSynchronize(); }
    unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%d independent MFMAs: issue took %llu ticks; 320 nops took %llu ticks\n", N, h[0], h[1]);
}
int main() { run<1>(); run<2>(); run<4>(); run<8>(); return 0; }
