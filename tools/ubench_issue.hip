// Micro-benchmark of the SIMD issue model on gfx950 (dev tool): how MFMA chains, plain VALU and transcendentals overlap
// inside one wave and between the two waves of a SIMD.  Every stream is inline asm, so the instruction order is exact.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/ubench tools/ubench_issue.hip && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define FMA(x, y) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define NOPS(n) asm volatile("s_nop " #n)

// MODE bits per wave role: what a wave executes per iteration
//  1: 17 dependent MFMAs           2: 96 independent-ish VALU (16 regs x 6 rounds)
//  4: 32 transcendentals            8: interleaved {MFMA, 5 VALU} x 17   16: interleaved {MFMA, 4 VALU, 1..2 EXP} x 17
template <int ROLE>
__device__ __forceinline__ void body(f32x16& acc, f32x4 a, f32x4 b, float (&v)[16], float y) {
    if (ROLE == 1) {
#pragma unroll
        for (int i = 0; i < 17; ++i) MFMA(acc, a, b);
    } else if (ROLE == 2) {
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) FMA(v[i], y);
    } else if (ROLE == 4) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) EXP(v[i]);
    } else if (ROLE == 8) {
#pragma unroll
        for (int i = 0; i < 17; ++i) {
            MFMA(acc, a, b);
            FMA(v[(5 * i) & 15], y); FMA(v[(5 * i + 1) & 15], y); FMA(v[(5 * i + 2) & 15], y); FMA(v[(5 * i + 3) & 15], y);
            FMA(v[(5 * i + 4) & 15], y);
        }
    } else if (ROLE == 16) {
#pragma unroll
        for (int i = 0; i < 17; ++i) {
            MFMA(acc, a, b);
            FMA(v[(4 * i) & 15], y); FMA(v[(4 * i + 1) & 15], y); EXP(v[(4 * i + 2) & 15]); FMA(v[(4 * i + 3) & 15], y);
            EXP(v[(4 * i + 5) & 15]);
        }
    } else if (ROLE == 32) {   // VALU + transcendentals mixed, no MFMA (the softmax burst): 96 VALU + 32 EXP
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) { FMA(v[i], y); EXP(v[(i + 8) & 15]); FMA(v[(i + 3) & 15], y); FMA(v[(i + 5) & 15], y); }
    }
}

template <int ROLE_A, int ROLE_B, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(float* out, int iters, unsigned long long* cyc) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc;
    float v[16];
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; v[i] = 0.001f * (threadIdx.x + i); }
    f32x4 a = {1e-3f, 2e-3f, 3e-3f, 4e-3f}, b = {1e-3f, 1e-3f, 2e-3f, 2e-3f};
    const float y = 0.5f;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const bool second = (WAVES == 8) && wave >= 4;
    for (int it = 0; it < iters; ++it) {
        if (!second) body<ROLE_A>(acc, a, b, v, y);
        else body<ROLE_B>(acc, a, b, v, y);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i] + v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * WAVES + wave] = t1 - t0;
}

template <int A, int B, int W>
void run(const char* name) {
    const int blocks = 256, iters = 2000;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * W * 64 * 4); hipMalloc(&cyc, blocks * W * 8);
    hipLaunchKernelGGL((k<A, B, W>), dim3(blocks), dim3(W * 64), 0, 0, out, 10, cyc);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<A, B, W>), dim3(blocks), dim3(W * 64), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * W);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double a = 0, b = 0; int na = 0, nb = 0;
    for (int i = 0; i < blocks * W; ++i) { if (W == 8 && (i % W) >= 4) { b += h[i]; nb++; } else { a += h[i]; na++; } }
    printf("%-58s first-group %7.1f cyc/iter", name, a / na / iters);
    if (nb) printf("   second-group %7.1f cyc/iter", b / nb / iters);
    printf("\n");
    hipFree(out); hipFree(cyc);
}

int main() {
    run<1, 1, 4>("1 wave/SIMD: 17 dependent MFMA");
    run<2, 2, 4>("1 wave/SIMD: 96 VALU");
    run<4, 4, 4>("1 wave/SIMD: 32 EXP");
    run<32, 32, 4>("1 wave/SIMD: 96 VALU + 32 EXP mixed");
    run<8, 8, 4>("1 wave/SIMD: {MFMA,5 VALU}x17 interleaved");
    run<16, 16, 4>("1 wave/SIMD: {MFMA,3 VALU,2 EXP}x17 interleaved");
    run<1, 1, 8>("2 waves/SIMD: both 17 MFMA");
    run<2, 2, 8>("2 waves/SIMD: both 96 VALU");
    run<4, 4, 8>("2 waves/SIMD: both 32 EXP");
    run<32, 32, 8>("2 waves/SIMD: both 96 VALU + 32 EXP");
    run<1, 2, 8>("2 waves/SIMD: A=17 MFMA, B=96 VALU");
    run<1, 32, 8>("2 waves/SIMD: A=17 MFMA, B=96 VALU+32 EXP");
    run<1, 4, 8>("2 waves/SIMD: A=17 MFMA, B=32 EXP");
    run<8, 8, 8>("2 waves/SIMD: both {MFMA,5 VALU}x17");
    run<16, 16, 8>("2 waves/SIMD: both {MFMA,3 VALU,2 EXP}x17");
    return 0;
}
