// Micro-benchmark (dev tool): the dense kernel's tile step as an instruction skeleton, in two shapes -
//   X  the shipped shape: 8 waves (2 per SIMD), each 32 target columns: per step 16 x {score MFMA ; ds_read_b128 ; fma ; exp ;
//      max3|cvt_pk} + 2 label MFMAs + ONE barrier (waves 4-7 take theirs after gap 7);
//   W  the "wide" shape: 4 waves (1 per SIMD), each 64 target columns: per step 16 x {2 score MFMAs on two accumulators ;
//      ds_read_b128 ; 2 x (fma ; exp ; max3|cvt_pk)} + 4 label MFMAs + ONE barrier.
// Both do the same work per workgroup and step (32 score MFMAs per SIMD); W reads half the LDS bytes.  Every stream is inline asm,
// so the instruction order is exact.   hipcc -O3 --offload-arch=gfx950 -o /tmp/ubw tools/ubench_wide.hip && /tmp/ubw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define FMA(d, x, y) asm volatile("v_fma_f32 %0, %1, %2, %2" : "=v"(d) : "v"(x), "v"(y))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define MAX3(d, x, y) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(d) : "v"(x), "v"(y))
#define CVT(d, x, y) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define DSREAD(d, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(d) : "v"(addr))

// one row of the softmax of the previous tile: e = fma(Sp, c, LM); q = exp2(e); every second row a max3 (alarm) and a pack
#define ROW(Sp, r, pk)                                          \
    do {                                                        \
        FMA(q[(r) & 1], Sp[r], y);                              \
        EXP(q[(r) & 1]);                                        \
        if ((r) & 1) {                                          \
            MAX3(mx, Sp[(r) - 1], Sp[r]);                       \
            CVT(pk[(r) >> 1], q[0], q[1]);                      \
        }                                                       \
    } while (0)

// F bits (shape X only): 1 = waves 0-3 stage five 1-KiB LDS-DMA pieces per step (three steps ahead, `s_waitcnt vmcnt(5)` before
// their barrier), 2 = the label fragments come from LDS (2 ds_read_b128 at gap 10), 4 = ~40 dependent scalar instructions of cursor
// arithmetic + two v_readlane per step, 8 = the alarm: v_cmp + s_cbranch_vccz in front of the label MFMAs
template <int SHAPE, int LAB_IN_CHAIN, int F = 0, int NS = 10, int INGAP = 0>
__global__ __launch_bounds__(SHAPE == 0 ? 512 : 256) void k(float* out, int iters, unsigned long long* cyc, const unsigned char* ring, unsigned ring_bytes) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[6 * 22528];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 6 * 22528 / 4; i += blockDim.x) ((float*)smem)[i] = 1e-3f * (i & 255);
    __syncthreads();
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const unsigned base = (unsigned)(size_t)(lds_ptr)smem + (lane & 31) * 528 + (lane >> 5) * 16;
    f32x16 S0, S1, P0, P1, Y0, Y1;
    for (int i = 0; i < 16; ++i) { S0[i] = S1[i] = Y0[i] = Y1[i] = 0.f; P0[i] = 1e-3f * (i + lane); P1[i] = 2e-3f * (i + lane); }
    f32x4 a[8], b = {1e-3f, 1e-3f, 2e-3f, 2e-3f}, b2 = {2e-3f, 1e-3f, 2e-3f, 1e-3f}, lab = {1.f, 0.f, 1.f, 0.f}, lab2 = {0.f, 1.f, 0.f, 1.f};
    for (int i = 0; i < 8; ++i) a[i] = f32x4{1e-3f, 2e-3f, 3e-3f, 4e-3f};
    float q[2] = {0.f, 0.f}, mx = 0.f;
    const float y = 0.5f;
    f32x4 pkA, pkB, pkC, pkD;
    float pk0[8], pk1[8];
    for (int i = 0; i < 8; ++i) pk0[i] = pk1[i] = 0.f;
    const bool second = SHAPE == 0 && wave >= 4;
    unsigned slot = 0, stg = 3 * 22528;
    const unsigned smem_base = (unsigned)(size_t)(lds_ptr)smem;
    unsigned src_off = (blockIdx.x * 184u * 16384u) % (ring_bytes - 65536u);      // this workgroup's place in the reference stream
    const unsigned lane_off = lane * 16;
    int cur_a = 0, cur_b = 0, cur_c = blockIdx.x;      // scalar cursor state
    float lanev = (float)lane;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    auto step = [&](f32x16& S0, f32x16& S1, const f32x16& P0, const f32x16& P1) __attribute__((always_inline)) {
        const unsigned ad = base + slot;
        const unsigned prv = slot == 0 ? 5 * 22528 : slot - 22528;
        slot = slot == 5 * 22528 ? 0 : slot + 22528;
        if (SHAPE == 0) {
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                MFMA(S0, a[ks & 7], b);
                if (ks == 0) DSREAD(a[0], ad, 0); if (ks == 1) DSREAD(a[1], ad, 32); if (ks == 2) DSREAD(a[2], ad, 64);
                if (ks == 3) DSREAD(a[3], ad, 96); if (ks == 4) DSREAD(a[4], ad, 128); if (ks == 5) DSREAD(a[5], ad, 160);
                if (ks == 6) DSREAD(a[6], ad, 192); if (ks == 7) DSREAD(a[7], ad, 224); if (ks == 8) DSREAD(a[0], ad, 256);
                if (ks == 9) DSREAD(a[1], ad, 288); if (ks == 10) DSREAD(a[2], ad, 320); if (ks == 11) DSREAD(a[3], ad, 352);
                if (ks == 12) DSREAD(a[4], ad, 384); if (ks == 13) DSREAD(a[5], ad, 416); if (ks == 14) DSREAD(a[6], ad, 448);
                if (ks == 15) DSREAD(a[7], ad, 480);
                ROW(P0, ks, pk0);
                if ((F & 1) && !second && ks % 3 == 1) {      // pieces 0..4 in gaps 1, 4, 7, 10, 13
                    unsigned vtmp;
                    asm volatile("s_add_u32 m0, %3, %4\n\tv_add_u32 %0, %1, %2\n\tglobal_load_lds_dwordx4 %0, %5"
                                 : "=&v"(vtmp)
                                 : "v"(lane_off), "s"(src_off + (unsigned)(ks / 3) * 4096u), "s"(smem_base + stg), "s"((unsigned)(wave + 4 * (ks / 3)) * 1024u), "s"(ring)
                                 : "memory", "scc");
                }
                if ((F & 4) && INGAP && ks < NS) {      // the same scalar groups, one per gap
                    asm volatile("s_add_i32 %0, %0, 1\n\ts_cmp_eq_u32 %0, %2\n\ts_cselect_b32 %0, 0, %0\n\ts_addc_u32 %1, %1, 0"
                                 : "+s"(cur_a), "+s"(cur_b) : "s"(9 + ks) : "scc");
                }
                if ((F & 2) && ks == 10) {
                    asm volatile("ds_read_b128 %0, %2 offset:17408\n\tds_read_b128 %1, %2 offset:18432" : "=v"(lab), "=v"(lab2) : "v"(smem_base + prv + lane * 16));
                }
                if (second && ks == 7) asm volatile("s_barrier" ::: "memory");
                if (ks >= 8) asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
            }
            pkA = f32x4{pk0[0], pk0[1], pk0[2], pk0[3]}; pkB = f32x4{pk0[4], pk0[5], pk0[6], pk0[7]};
            if (F & 8) {
                if (__builtin_expect(__any(mx > 1.0e30f), 0)) {      // never true; the branch is what is timed
#pragma unroll
                    for (int i = 0; i < 16; ++i) Y0[i] *= 0.5f;
                }
            }
            if (F & 2) asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory");
            MFMA(Y0, lab, pkA);
            MFMA(Y0, (F & 2) ? lab2 : lab, pkB);
            if (F & 4) {      // the cursor: dependent scalar chains and two lane reads, as the tile loop has them
#pragma unroll
                for (int i = 0; i < (INGAP ? 0 : NS); ++i) {
                    asm volatile("s_add_i32 %0, %0, 1\n\ts_cmp_eq_u32 %0, %2\n\ts_cselect_b32 %0, 0, %0\n\ts_addc_u32 %1, %1, 0"
                                 : "+s"(cur_a), "+s"(cur_b) : "s"(9 + i) : "scc");
                }
                if (F & 16) {
                    const int rl = __builtin_amdgcn_readlane(__float_as_int(lanev), cur_a & 63);
                    const int rl2 = __builtin_amdgcn_readlane(__float_as_int(lanev), cur_b & 63);
                    cur_c += (rl ^ rl2) & 1;
                }
            }
            if (F & 1) {
                src_off += 16384u;
                if (src_off + 32768u > ring_bytes) src_off = 0;
                stg = stg == 5 * 22528 ? 0 : stg + 22528;
                if (!second) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            }
            if (!second) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                MFMA(S0, a[ks & 7], b);
                if (ks == 0) DSREAD(a[0], ad, 0); if (ks == 1) DSREAD(a[1], ad, 32); if (ks == 2) DSREAD(a[2], ad, 64);
                if (ks == 3) DSREAD(a[3], ad, 96); if (ks == 4) DSREAD(a[4], ad, 128); if (ks == 5) DSREAD(a[5], ad, 160);
                if (ks == 6) DSREAD(a[6], ad, 192); if (ks == 7) DSREAD(a[7], ad, 224); if (ks == 8) DSREAD(a[0], ad, 256);
                if (ks == 9) DSREAD(a[1], ad, 288); if (ks == 10) DSREAD(a[2], ad, 320); if (ks == 11) DSREAD(a[3], ad, 352);
                if (ks == 12) DSREAD(a[4], ad, 384); if (ks == 13) DSREAD(a[5], ad, 416); if (ks == 14) DSREAD(a[6], ad, 448);
                if (ks == 15) DSREAD(a[7], ad, 480);
                ROW(P0, ks, pk0);
                MFMA(S1, a[(ks + 1) & 7], b2);      // (the same fragment in the real kernel; another register here keeps the asm simple)
                ROW(P1, ks, pk1);
                if ((F & 1) && ks % 3 == 1) {      // all four waves stage: pieces 0..4 in gaps 1, 4, 7, 10, 13
                    unsigned vtmp;
                    asm volatile("s_add_u32 m0, %3, %4\n\tv_add_u32 %0, %1, %2\n\tglobal_load_lds_dwordx4 %0, %5"
                                 : "=&v"(vtmp)
                                 : "v"(lane_off), "s"(src_off + (unsigned)(ks / 3) * 4096u), "s"(smem_base + stg), "s"((unsigned)(wave + 4 * (ks / 3)) * 1024u), "s"(ring)
                                 : "memory", "scc");
                }
                if ((F & 4) && INGAP && ks < NS) {
                    asm volatile("s_add_i32 %0, %0, 1\n\ts_cmp_eq_u32 %0, %2\n\ts_cselect_b32 %0, 0, %0\n\ts_addc_u32 %1, %1, 0"
                                 : "+s"(cur_a), "+s"(cur_b) : "s"(9 + ks) : "scc");
                }
                if ((F & 2) && ks == 10) {
                    asm volatile("ds_read_b128 %0, %2 offset:17408\n\tds_read_b128 %1, %2 offset:18432" : "=v"(lab), "=v"(lab2) : "v"(smem_base + prv + lane * 16));
                }
                if (LAB_IN_CHAIN && ks == 8) { pkA = f32x4{pk0[0], pk0[1], pk0[2], pk0[3]}; pkC = f32x4{pk1[0], pk1[1], pk1[2], pk1[3]}; MFMA(Y0, lab, pkA); }
                if (LAB_IN_CHAIN && ks == 9) MFMA(Y1, lab, pkC);
                if (ks >= 8) asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
            }
            pkB = f32x4{pk0[4], pk0[5], pk0[6], pk0[7]}; pkD = f32x4{pk1[4], pk1[5], pk1[6], pk1[7]};
            if (F & 8) {
                if (__builtin_expect(__any(mx > 1.0e30f), 0)) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) { Y0[i] *= 0.5f; Y1[i] *= 0.5f; }
                }
            }
            if (F & 2) asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory");
            if (!LAB_IN_CHAIN) {
                pkA = f32x4{pk0[0], pk0[1], pk0[2], pk0[3]}; pkC = f32x4{pk1[0], pk1[1], pk1[2], pk1[3]};
                MFMA(Y0, lab, pkA);
                MFMA(Y1, lab, pkC);
            }
            MFMA(Y0, (F & 2) ? lab2 : lab, pkB);
            MFMA(Y1, (F & 2) ? lab2 : lab, pkD);
            if (F & 4) {
#pragma unroll
                for (int i = 0; i < (INGAP ? 0 : NS); ++i) {
                    asm volatile("s_add_i32 %0, %0, 1\n\ts_cmp_eq_u32 %0, %2\n\ts_cselect_b32 %0, 0, %0\n\ts_addc_u32 %1, %1, 0"
                                 : "+s"(cur_a), "+s"(cur_b) : "s"(9 + i) : "scc");
                }
                if (F & 16) {
                    const int rl = __builtin_amdgcn_readlane(__float_as_int(lanev), cur_a & 63);
                    const int rl2 = __builtin_amdgcn_readlane(__float_as_int(lanev), cur_b & 63);
                    cur_c += (rl ^ rl2) & 1;
                }
            }
            if (F & 1) {
                src_off += 16384u;
                if (src_off + 32768u > ring_bytes) src_off = 0;
                stg = stg == 5 * 22528 ? 0 : stg + 22528;
                asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    };
    for (int it = 0; it < iters; it += 2) {
        step(S0, S1, P0, P1);
        step(P0, P1, S0, S1);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = mx + (float)(cur_a + cur_b + cur_c);
    for (int i = 0; i < 16; ++i) s += S0[i] + S1[i] + Y0[i] + Y1[i] + P0[i] + P1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int SHAPE, int LIC, int F = 0, int NS = 10, int INGAP = 0>
void run(const char* name) {
    const int blocks = 256, iters = 2000, W = SHAPE == 0 ? 8 : 4;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * 512 * 4); hipMalloc(&cyc, blocks * 8 * 8);
    static unsigned char* ring = nullptr;
    const unsigned ring_bytes = 9u * 6432u * 512u;      // nine frames of 480p features
    if (!ring) { hipMalloc(&ring, ring_bytes); hipMemset(ring, 0, ring_bytes); }
    hipLaunchKernelGGL((k<SHAPE, LIC, F, NS, INGAP>), dim3(blocks), dim3(W * 64), 0, 0, out, 10, cyc, ring, ring_bytes);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<SHAPE, LIC, F, NS, INGAP>), dim3(blocks), dim3(W * 64), 0, 0, out, iters, cyc, ring, ring_bytes);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double a = 0; int na = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < W; ++w) { a += h[b * 8 + w]; na++; }
    printf("%-70s %7.1f cycles per step of 256 columns x 32 rows\n", name, a / na / iters);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0, 0>("X: 8 waves x 32 columns, barrier skewed (shipped shape)");
    run<1, 0>("W: 4 waves x 64 columns, label MFMAs behind the chain");
    run<1, 1>("W: 4 waves x 64 columns, two of the label MFMAs inside the chain");
    run<0, 0, 1>("X + LDS-DMA staging (5 pieces per step, waves 0-3)");
    run<0, 0, 2>("X + label fragments from LDS");
    run<0, 0, 20>("X + cursor: 40 scalar instructions + 2 readlane behind the chain");
    run<0, 0, 4>("X + cursor: 40 scalar instructions, no readlane");
    run<0, 0, 4, 5>("X + cursor: 20 scalar instructions");
    run<0, 0, 4, 2>("X + cursor: 8 scalar instructions");
    run<0, 0, 20, 0>("X + 2 readlane only");
    run<0, 0, 4, 10, 1>("X + cursor: 40 scalar instructions, 4 per gap in gaps 0-9");
    run<0, 0, 4, 5, 1>("X + cursor: 20 scalar instructions, 4 per gap in gaps 0-4");
    run<0, 0, 8>("X + alarm compare and branch");
    run<0, 0, 31>("X + all");
    run<0, 0, 31, 10, 1>("X + all, the 40 scalar instructions 4 per gap");
    run<0, 0, 31, 5, 0>("X + all, 20 scalar instructions behind the chain");
    run<0, 0, 31, 5, 1>("X + all, 20 scalar instructions 4 per gap");
    run<0, 0, 11>("X + all but the cursor");
    run<0, 0, 30>("X + all but staging");
    run<1, 0, 31>("W + all");
    run<1, 1, 31>("W + all, two label MFMAs in the chain");
    run<1, 0, 31, 10, 1>("W + all, the 40 scalar instructions 4 per gap");
    run<1, 0, 1>("W + staging");
    run<1, 0, 20>("W + cursor + readlane");
    return 0;
}
