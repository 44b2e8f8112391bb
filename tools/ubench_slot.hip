// dev tool: what does one v6 "slot" cost?  {MFMA ; softmax stage ops} x 16 per iteration, every stream inline asm, one wave per SIMD.
// Variants isolate which ingredient spoils the MFMA/VALU overlap that {MFMA, 5 plain VALU} shows (tools/ubench_issue.hip).
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/us tools/ubench_slot.hip && /tmp/us
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define MFMA_V(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA_A(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "a"(b))
#define EXP(d, s) asm volatile("v_exp_f32 %0, %1" : "=v"(d) : "v"(s))
#define PKFMA(d, a, b, c) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c))
#define PKADD(d, a) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d) : "v"(a))
#define FMA(d, a, b, c) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c))
#define ADD(d, a) asm volatile("v_add_f32 %0, %0, %1" : "+v"(d) : "v"(a))

// MODE: 0 MFMA only (VGPR operands)   1 MFMA only (AGPR operands)
//       2 {MFMA ; pk_add ; exp ; exp ; pk_fma}, skewed (no intra-slot dependences)
//       3 same with scalar ops {add, add, exp, exp, fma, fma}
//       4 VALU part of mode 2 alone   5 VALU part of mode 3 alone
//       6 mode 2 but exps first: {MFMA ; exp ; exp ; pk_add ; pk_fma}
//       7 {MFMA ; pk_add ; pk_fma ; pk_add ; pk_fma} (no transcendentals)
//       8 mode 2 with the MFMA reading AGPR operands
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* cyc) {
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    f32x4 a = {1e-3f, 2e-3f, 3e-3f, 4e-3f}, b = {1e-3f, 1e-3f, 2e-3f, 2e-3f};
    f32x2 s[8], xn = {0.1f, 0.2f}, e0 = {0.f, 0.f}, e1 = {0.f, 0.f}, l = {0.f, 0.f}, c2 = {1.1f, 1.1f}, nm = {-0.3f, -0.3f};
    for (int i = 0; i < 8; ++i) { s[i][0] = 0.01f * (threadIdx.x + i); s[i][1] = 0.02f * i; }
    asm volatile("" : "+a"(a), "+a"(b));   // park copies in AGPRs for the _A variants
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            if (MODE != 4 && MODE != 5) {
                if (MODE == 1 || MODE == 8) MFMA_A(acc, a, b);
                else MFMA_V(acc, a, b);
            }
            f32x2& ec = (ks & 1) ? e1 : e0;   // written this slot
            f32x2& ep = (ks & 1) ? e0 : e1;   // written last slot
            if (MODE == 2 || MODE == 4 || MODE == 8) {
                PKADD(l, ep);
                EXP(ec[0], xn[0]); EXP(ec[1], xn[1]);
                PKFMA(xn, s[ks & 7], c2, nm);
            } else if (MODE == 3 || MODE == 5) {
                ADD(l[0], ep[0]); ADD(l[1], ep[1]);
                EXP(ec[0], xn[0]); EXP(ec[1], xn[1]);
                FMA(xn[0], s[ks & 7][0], c2[0], nm[0]); FMA(xn[1], s[ks & 7][1], c2[1], nm[1]);
            } else if (MODE == 6) {
                EXP(ec[0], xn[0]); EXP(ec[1], xn[1]);
                PKADD(l, ep);
                PKFMA(xn, s[ks & 7], c2, nm);
            } else if (MODE == 7) {
                PKADD(l, ep); PKFMA(ec, s[ks & 7], c2, nm); PKADD(e0, s[(ks + 1) & 7]); PKFMA(xn, s[(ks + 3) & 7], c2, nm);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = l[0] + l[1] + xn[0] + e0[0] + e1[1];
    for (int i = 0; i < 16; ++i) r += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char* name) {
    const int blocks = 256, iters = 2000;
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, blocks * 256 * 4); (void)hipMalloc(&cyc, blocks * 4 * 8);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, 10, cyc);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double a = 0;
    for (auto v : h) a += v;
    printf("%-70s %7.1f cycles per slot\n", name, a / h.size() / iters / 16);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    run<0>("MFMA only (VGPR operands)");
    run<1>("MFMA only (AGPR operands)");
    run<4>("VALU only: pk_add, exp, exp, pk_fma");
    run<5>("VALU only: add, add, exp, exp, fma, fma");
    run<7>("{MFMA ; pk_add ; pk_fma ; pk_add ; pk_fma}  (no transcendentals)");
    run<2>("{MFMA ; pk_add ; exp ; exp ; pk_fma}");
    run<6>("{MFMA ; exp ; exp ; pk_add ; pk_fma}");
    run<3>("{MFMA ; add ; add ; exp ; exp ; fma ; fma}");
    run<8>("{MFMA(AGPR) ; pk_add ; exp ; exp ; pk_fma}");
    return 0;
}
