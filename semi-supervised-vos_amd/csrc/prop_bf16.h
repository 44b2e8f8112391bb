// Shared pieces of the bf16 propagation kernels (prop_dense.h, prop_mask.h): LDS slot layout, LDS-DMA pieces, half-wave exchanges,
// label fragments and the label product, the prior tile.  The kernels replace reference src/model/predict.py:46-70 (mm,
// *= temperature, softmax(dim=0), spatial weights, label mm) without ever forming the (N HW) x HW affinity:
//   * S^T tile = mfma(A = 32 reference pixels x K, B = K x 32 target pixels): every lane owns ONE target pixel (column) and 16
//     reference pixels (rows) of the 32x32 tile, so the column statistics are in-register; the two half-waves share a column and
//     exchange one value when needed (v_permlane32_swap).
//   * K = 256 feature channels = 16 x v_mfma_f32_32x32x16_bf16 per tile.  The Gaussian spatial prior w[r,t] depends on the PIXEL
//     positions only: the reference stream is walked pixel tile by pixel tile with the N sampled frames INNER, and a wave computes
//     its 32x32 tile of log2 w once per (pixel tile, sigma) - ONE extra 16-deep MFMA whose channels carry the prior
//     [-dist^2/(sigma^2 tau) is bilinear in (reference coords, target coords): dist^2 = Qp + Qt - a_p(2a_t + 2b_t/W)
//     - b_p(2 gamma b_t + 2a_t/W); 3-way bf16 splits of the real-valued factors keep ~24 significant bits].
//   * numerators out[k,t] = sum_r L[k,r] a[r,t] are a second MFMA: the weighted probabilities a (rows in registers / column on the
//     lane) are packed to bf16 and used as the B operand against the label matrix stored in HBM in A-operand order.
//   * one workgroup = 8 waves x 32 target pixels = 256 target pixels (B fragments live in registers); reference tiles (32 pixels x
//     512 B in 528-B padded rows: conflict-free ds_read_b128 at immediate offsets) stream through an LDS ring filled by LDS-DMA.
//   * persistent, XCD-partitioned lockstep grid (Segment table, common.h); partials (m, l, numerators) per segment go to HBM and
//     combine_kernel merges them.
#pragma once
#include "common.h"

namespace vosprop {

constexpr int kRowB = kC * 2 + 16;                // 528: padded LDS row
constexpr int kLdsFeat = 17 * 1024;               // 32 rows x 528 B = 16896, rounded up to whole 1-KiB LDS-DMA pieces
constexpr int kLdsCoord = 2 * 32 * 16;            // 1024
constexpr int kLdsLab = 2 * 64 * 16;              // 2048
constexpr int kOffCoord = kLdsFeat;
constexpr int kOffLabHi = kLdsFeat + kLdsCoord;
constexpr int kOffLabLo = kOffLabHi + kLdsLab;
constexpr int kLdsBuf = kOffLabLo + kLdsLab;      // 22528
constexpr int kGlbFeat = kTileR * kC * 2;         // 16384 bytes of one tile in HBM
constexpr float kRescaleThr = 8.0f;               // defer-max threshold in log2 units (p <= 2^8)
constexpr float kNegBig = -1.0e30f;
constexpr float kSumThrV3 = 256.0f;              // 2^kRescaleThr, threshold on a tile's partial denominator

// One LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to 1 KiB of LDS at the wave-uniform byte address lds_dst.
// Inline asm ON PURPOSE (cdna guide 5.7): hipcc counts a __builtin_amdgcn_global_load_lds in its s_waitcnt bookkeeping and, not
// knowing which LDS bytes it writes, drains it - `s_waitcnt vmcnt(0)` before the next ds_read of the MFMA chain and before every
// s_barrier - which exposes the L2 latency of a piece issued moments earlier inside the burst that was meant to hide it.  The asm
// form is invisible to that pass: the kernel's own counted `s_waitcnt vmcnt(3)` at the end of a step (pieces issued one step
// earlier) + the barrier that follows are what order the pieces before their readers.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// The same with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset (saddr form): no 64-bit vector add per
// piece, no VGPR pair.  M0 (the LDS destination) is set inside the statement and NOT restored: nothing else in the kernels that use
// this form reads M0 (gfx950 LDS instructions do not), and hipcc sets M0 itself before any instruction of its own that needs it.
__device__ __forceinline__ void glds16s(unsigned voff, const void* sbase, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// ... and with the two additions of a piece folded in (the tile loops are issue-bound, every instruction counts): M0 = lds_base +
// lds_off in one scalar add, lane offset + wave-uniform byte offset in one vector add - which also is the wait state the hardware
// wants between the M0 write and the LDS-DMA, so no s_nop.  Three instructions per piece.
__device__ __forceinline__ void glds16s2(unsigned lane_off, unsigned soff, const void* sbase, unsigned lds_base, unsigned lds_off) {
    unsigned vtmp;
    asm volatile("s_add_u32 m0, %3, %4\n\tv_add_u32 %0, %1, %2\n\tglobal_load_lds_dwordx4 %0, %5"
                 : "=&v"(vtmp)
                 : "v"(lane_off), "s"(soff), "s"(lds_base), "s"(lds_off), "s"(sbase)
                 : "memory", "scc");
}

__device__ __forceinline__ float half_max(float x) {   // max(x[lane], x[lane ^ 32]) in every lane
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_sum(float x) {   // x[lane] + x[lane ^ 32]
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

struct ColState {
    float m;     // running max of raw S of this column (shared by both half-waves)
    float l;     // partial denominator over this half-wave's rows: sum 2^((S - m) c)
    f32x16 Y;    // numerators: rows = classes, column = this lane's target pixel
};

// Label fragments (MFMA A operand) of one tile.
template <bool LAB_LO>
struct LabFrag {
    bf16x8 h0, h1, l0, l1;
    __device__ __forceinline__ void load(const unsigned char* lb, int lane) {
        const unsigned char* lh = lb + kOffLabHi + lane * 16;
        h0 = *(const bf16x8*)(lh);
        h1 = *(const bf16x8*)(lh + 1024);
        if (LAB_LO) {
            l0 = *(const bf16x8*)(lh + kLdsLab);
            l1 = *(const bf16x8*)(lh + kLdsLab + 1024);
        }
    }
};

// A-operand fragments of a tile that are fetched ahead of its MFMA chain: channels 0..127 (8 x ds_read_b128).
template <bool PROB>
struct AFrag {
    bf16x8 a[8];
    // two halves so that the second one can be issued after the softmax has released its registers
    __device__ __forceinline__ void prefetch_lo(const unsigned char* lb, int j, int h) {
        const unsigned char* arow = lb + j * kRowB + h * 16;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) a[ks] = *(const bf16x8*)(arow + ks * 32);
    }
    __device__ __forceinline__ void prefetch_hi(const unsigned char* lb, int j, int h) {
        const unsigned char* arow = lb + j * kRowB + h * 16;
#pragma unroll
        for (int ks = 4; ks < 8; ++ks) a[ks] = *(const bf16x8*)(arow + ks * 32);
    }
    __device__ __forceinline__ void prefetch(const unsigned char* lb, int j, int h) {
        prefetch_lo(lb, j, h);
        prefetch_hi(lb, j, h);
    }
};

// The prior tile of one (pixel tile, sigma): LW[r] = log2 w[r, t] = (-dist^2 g) c  [X from one 16-deep MFMA] - g Q_t c.
// Dense mode keeps w = 2^LW (multiplied into the probabilities), the top-k passes keep LW (added to the exponent they rank by).
template <bool KEEP_LOG>
__device__ __forceinline__ void prior_tile(const unsigned char* lb, int j, int h, const bf16x8& Bx, float c, float kq, float (&Wt)[16]) {
    const bf16x8 ax = *(const bf16x8*)(lb + kOffCoord + h * 512 + j * 16);
    f32x16 X;
#pragma unroll
    for (int r = 0; r < 16; ++r) X[r] = 0.0f;
    X = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax, Bx, X, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float lw = __builtin_fmaf(X[r], c, -kq);
        Wt[r] = KEEP_LOG ? lw : __builtin_amdgcn_exp2f(lw);
    }
}

// The label product of a tile: Y[class, t] += L[class, rows] a[rows, t] with a packed to bf16 - the rows-in-registers /
// column-on-lane layout of the accumulator IS the B-operand layout.
template <bool LAB_LO>
__device__ __forceinline__ void label_mfmas(const LabFrag<LAB_LO>& lab, const bf16x8& pk0, const bf16x8& pk1, f32x16& Y) {
    Y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lab.h0, pk0, Y, 0, 0, 0);
    Y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lab.h1, pk1, Y, 0, 0, 0);
    if (LAB_LO) {
        Y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lab.l0, pk0, Y, 0, 0, 0);
        Y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lab.l1, pk1, Y, 0, 0, 0);
    }
}

// small VALU helpers shared with prop_dense.h
__device__ __forceinline__ float vmaxf(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float max16v(const float (&v)[16]) {
    float t0, t1, t2, t3, t4, r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t0) : "v"(v[0]), "v"(v[1]), "v"(v[2]));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t1) : "v"(v[3]), "v"(v[4]), "v"(v[5]));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t2) : "v"(v[6]), "v"(v[7]), "v"(v[8]));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t3) : "v"(v[9]), "v"(v[10]), "v"(v[11]));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t4) : "v"(v[12]), "v"(v[13]), "v"(v[14]));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t0) : "v"(t0), "v"(t1), "v"(t2));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t3) : "v"(t3), "v"(t4), "v"(v[15]));
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(t0), "v"(t3));
    return r;
}

}  // namespace vosprop
