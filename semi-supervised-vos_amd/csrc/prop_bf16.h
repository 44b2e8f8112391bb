// The hot kernel: fused affinity -> online column softmax -> spatial prior -> label product.
//
// Replaces reference src/model/predict.py:46-70 (mm, *=temperature, softmax(dim=0), spatial weights,
// label mm) without ever forming the (N*HW) x HW affinity.
//
// Mapping onto CDNA4 (gfx950):
//   * S^T tile = mfma(A = 32 reference pixels x K, B = K x 32 target pixels): every lane owns ONE target
//     pixel (column) and 16 reference pixels (rows) of the 32x32 tile, so the column softmax statistics are
//     in-register; the two half-waves share a column and exchange one value per tile (v_permlane32_swap).
//   * K = 256 feature channels = 16 x v_mfma_f32_32x32x16_bf16 per tile.  The Gaussian spatial prior w[r,t] depends on the PIXEL
//     positions only, not on the frame: the reference stream is walked pixel tile by pixel tile with the N sampled frames INNER, so
//     a wave computes its 32x32 tile of w once per (pixel tile, sigma) - ONE extra 16-deep MFMA whose channels carry the prior
//     [-dist^2/(sigma^2 tau) is bilinear in (reference coords, target coords): dist^2 = Qp + Qt - a_p(2a_t + 2b_t/W)
//     - b_p(2 gamma b_t + 2a_t/W); 3-way bf16 splits of the real-valued factors keep ~24 significant bits] + 16 exponentials -
//     keeps it in 16 registers, and every frame of that pixel tile then costs ONE exponential and one multiply per score
//     (a = p w) instead of two exponentials: 1 + 2/N transcendentals per score instead of 2 (the kernel is VALU-issue bound).
//   * numerators  out[k,t] = sum_r L[k,r] a[r,t]  are a second MFMA: the weighted probabilities a (already
//     laid out rows-in-registers / column-on-lane) are packed to bf16 and used as the B operand against the
//     label matrix stored in HBM in A-operand order.  Denominators stay f32 on the VALU.
//   * one workgroup = 8 waves x 32 target pixels = 256 target pixels (B fragments live in registers);
//     reference tiles (32 pixels x 512 B in 528-B padded rows: conflict-free ds_read_b128 at immediate
//     offsets) stream through a 3-deep LDS ring, one barrier per tile.
//   * waves 4-7 run half a tile behind waves 0-3: while one wave of a SIMD is in its MFMA burst the other is in
//     its softmax (VALU/transcendental) burst, so the matrix pipe and the vector issue overlap.
//   * persistent stream-K grid (WorkMap in common.h): every CU gets the same number of tiles; the reference
//     stream is partitioned over the 8 XCDs so each L2 keeps 1/8 of the features.
//   * partial (m, l, numerators) per (workgroup, target tile) go to HBM; combine_kernel merges them.
#pragma once
#include "common.h"

#ifdef VOSPROP_STAMP
#define VOSPROP_NSTAMP 16
// diagnostic build: STAMP_AT(k) adds the cycles since the previous stamp to bucket k (wave-uniform scalars)
#define STAMP_DECL unsigned long long tsum[VOSPROP_NSTAMP] = {}; unsigned long long tprev = 0
#define STAMP_ARGS , unsigned long long (&tsum)[VOSPROP_NSTAMP], unsigned long long& tprev
#define STAMP_PASS , tsum, tprev
#define STAMP_AT(k)                                                                        \
    do {                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        unsigned long long tn_;                                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tn_)::"memory");      \
        tsum[k] += tn_ - tprev;                                                            \
        tprev = tn_;                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                 \
    } while (0)
#else
#define STAMP_DECL
#define STAMP_ARGS
#define STAMP_PASS
#define STAMP_AT(k) __builtin_amdgcn_sched_barrier(0)
#endif

#ifndef VOSPROP_ABLATE
#define VOSPROP_ABLATE 0   // timing experiments only (tools/ablate.sh): 1 = no exp, 2 = no score MFMAs, 4 = no staging, 8 = no barriers, 16 = stage one cached tile
#endif

namespace vosprop {

constexpr int kRowB = kC * 2 + 16;                // 528: padded LDS row
constexpr int kLdsFeat = 17 * 1024;               // 32 rows x 528 B = 16896, rounded up to whole 1-KiB LDS-DMA pieces
constexpr int kLdsCoord = 2 * 32 * 16;            // 1024
constexpr int kLdsLab = 2 * 64 * 16;              // 2048
constexpr int kOffCoord = kLdsFeat;
constexpr int kOffLabHi = kLdsFeat + kLdsCoord;
constexpr int kOffLabLo = kOffLabHi + kLdsLab;
constexpr int kLdsBuf = kOffLabLo + kLdsLab;      // 22528
constexpr int kRing4 = 4;                         // shipped kernel: tiles are staged THREE steps ahead (see the main loop)
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

// 16 MFMAs: S = R.T over the 256 channels.  The first 8 fragments were prefetched during the previous softmax burst; each MFMA
// is followed by the ds_read_b128 (immediate offset) that refills its slot with the fragment 8 steps ahead, so the chain never
// waits on LDS.
template <bool PROB, typename Hook>
__device__ __forceinline__ void tile_scores(const unsigned char* lb, int j, int h, const bf16x8 (&Bt)[16], AFrag<PROB>& f,
                                            f32x16& S, Hook&& hook) {
    const unsigned char* arow = lb + j * kRowB + h * 16;
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] = 0.0f;
#if VOSPROP_ABLATE & 2
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        asm volatile("" ::"v"(f.a[ks]), "v"(Bt[ks]));
        f.a[ks] = *(const bf16x8*)(arow + (ks + 8) * 32);
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) asm volatile("" ::"v"(f.a[ks]), "v"(Bt[ks + 8]));
    S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0], Bt[0], S, 0, 0, 0);
#else
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[ks], Bt[ks], S, 0, 0, 0);
        f.a[ks] = *(const bf16x8*)(arow + (ks + 8) * 32);
        hook(ks);
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[ks], Bt[ks + 8], S, 0, 0, 0);
        hook(ks + 8);
    }
#endif
}

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

// Online softmax update of one 32x32 score tile: p[r] = 2^((S[r] - m) c) against the running max, denominators updated.  The
// label product of the tile (weights a = p w, bf16 packing, label MFMAs) is NOT done here: it is deferred to the wave's next
// MFMA burst (label_product below), so this VALU-only burst - the longer of the two bursts of a step - gets shorter and the
// deferred VALU work hides in the shadow of the score MFMAs of the same wave.
template <bool PROB>
__device__ __forceinline__ void tile_softmax(int h, f32x16& S, float (&p)[16], ColState& st, float c, bool tail,
                                             int rows_valid STAMP_ARGS) {
    if (tail) {
        // padded reference rows of a frame's last tile must not enter the softmax (wave-uniform, rare)
        asm volatile("; tail tile" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (acc_row(r, h) >= rows_valid) S[r] = kNegBig;
    }
    // Optimistic pass: exponentiate against the CURRENT running max; the running max is only raised (and this tile
    // redone) when some score exceeds it by more than kRescaleThr - rare after the first tiles.  The test is on the tile's
    // partial denominator (any term above 2^8, or an overflow, pushes the sum of 16 non-negative terms above 2^8; a false
    // alarm only costs a redo) - no per-tile max reduction at all.
    float mc = st.m * c;
    float l0 = 0.0f, l1 = 0.0f;   // two chains: halves the dependent-add latency
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
#if VOSPROP_ABLATE & 1
        const float pa = __builtin_fmaf(S[r], c, -mc), pb = __builtin_fmaf(S[r + 1], c, -mc);
#else
        const float pa = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r], c, -mc));
        const float pb = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r + 1], c, -mc));
#endif
        if (PROB) {
            // probability mode: the label product sees p rounded to bf16; the denominator must see THE SAME numbers, or the
            // columns of the result sum to 1 +- 2^-9 instead of 1 (the reference's softmax columns sum to 1, and the class-axis
            // fusion of its flip strategies breaks ties on exactly that).  Rounding here makes the bf16 conversion
            // below exact.
            p[r] = bf16_round(pa);
            p[r + 1] = bf16_round(pb);
            l0 += p[r];
            l1 += p[r + 1];
        } else {
            p[r] = pa;
            p[r + 1] = pb;
            l0 += pa;
            l1 += pb;
        }
    }
    if (__any(l0 + l1 > kSumThrV3)) {
        // slow path: raise the running max (shared by the two half-waves of a column), rescale what was accumulated
        // against the old one exactly once, and redo this tile against the new one (cdna guide T13 hazard)
        asm volatile("; rescale" ::: "memory");
        float u0, u1, u2, u3, u4, smax;   // max raw score of this column in this tile (from S itself: the exponent loses it when
                                          // the running max is still the -1e30 start value)
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(u0) : "v"(S[0]), "v"(S[1]), "v"(S[2]));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(u1) : "v"(S[3]), "v"(S[4]), "v"(S[5]));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(u2) : "v"(S[6]), "v"(S[7]), "v"(S[8]));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(u3) : "v"(S[9]), "v"(S[10]), "v"(S[11]));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(u4) : "v"(S[12]), "v"(S[13]), "v"(S[14]));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(u0) : "v"(u0), "v"(u1), "v"(u2));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(u3) : "v"(u3), "v"(u4), "v"(S[15]));
        asm("v_max_f32 %0, %1, %2" : "=v"(smax) : "v"(u0), "v"(u3));
        const float tmax = half_max(smax);
        const float mn = fmaxf(st.m, tmax);
        const float sc = __builtin_amdgcn_exp2f((st.m - mn) * c);
        st.l *= sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) st.Y[r] *= sc;
        st.m = mn;
        mc = mn * c;
        l0 = 0.0f;
        l1 = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const float pa = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r], c, -mc));
            const float pb = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r + 1], c, -mc));
            if (PROB) {
                p[r] = bf16_round(pa);
                p[r + 1] = bf16_round(pb);
                l0 += p[r];
                l1 += p[r + 1];
            } else {
                p[r] = pa;
                p[r + 1] = pb;
                l0 += pa;
                l1 += pb;
            }
        }
    }
    st.l += l0 + l1;
}

// The deferred label product of the PREVIOUS tile: Y[class, t] += L[class, rows] a[rows, t] with a = p w (label mode) packed to
// bf16 - the rows-in-registers / column-on-lane layout of the accumulator IS the B-operand layout.  Two halves so that the
// caller can spread the VALU part over the gaps of its score-MFMA chain.
__device__ __forceinline__ bf16x8 pack_half(const float (&a)[16], int half) {
    bf16x8 pk;
#pragma unroll
    for (int e = 0; e < 8; ++e) pk[e] = (bf16_t)a[half * 8 + e];
    return pk;
}
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

// The round-1 two-burst schedule of the dense step (MODE must be 0; the shipped dense kernel and both top-k passes are
// prop_dense.h).  Kept for A/B timing only (VOSPROP_DENSE_TWO_BURST=1): same arithmetic, same results.
template <bool PROB, bool LAB_LO, int MODE>
__global__ __launch_bounds__(kWaves * 64, 2) void prop_bf16_kernel(const PropArgs A) {
    static_assert(MODE == 0, "the top-k passes moved to prop_dense.h (TK = 1 / 2)");
    __shared__ __attribute__((aligned(16))) unsigned char smem[kRing4 * kLdsBuf];
    // per-lane constants of the prior (target-side MFMA fragment and g Q_t c for both sigmas): needed twice per 9 tiles, so they
    // live in LDS (written and read by the same lane), not in 10 registers
    __shared__ __attribute__((aligned(16))) bf16x8 s_bx[2][kWaves * 64];
    __shared__ float s_kq[2][kWaves * 64];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifndef VOSPROP_GROUP_MODE
#define VOSPROP_GROUP_MODE 0
#endif
    // the half-tile-late wave group (experiments: 1 = odd waves, 2 = waves 2,3,6,7, 3 = no skew)
    const bool grpB = VOSPROP_GROUP_MODE == 0 ? wave >= 4 : VOSPROP_GROUP_MODE == 1 ? (wave & 1) != 0
                      : VOSPROP_GROUP_MODE == 2 ? ((wave >> 1) & 1) != 0 : false;
    const int j = lane & 31;       // MFMA row (A operand) / column (B operand, C/D)
    const int h = lane >> 5;       // k-half of the operand fragments / row-half of the accumulator

    const int TPF = A.tiles_per_frame;
    const float c = A.c;
    const bool ragged = A.HW != A.HWp;
    const int rows_last = A.HW - (TPF - 1) * kTileR;   // valid rows of a frame's last tile

    // ---- staging roles (LDS-DMA, global_load_lds_dwordx4: one wave instruction lands 1 KiB = 64 lanes x 16 B at a
    // wave-uniform LDS address, the SOURCE address is per lane).  A tile is 17 feature pieces (the 528-B padded row image is
    // produced by the per-lane source offsets: LDS unit q of 16 B holds chunk q % 33 of row q / 33, chunk 32 = padding),
    // 1 coordinate piece, 2 + 2 label pieces.  Wave w issues feature pieces w and w + 8, and one more: wave 0 feature
    // piece 16, wave 1 coordinates, waves 2-3 label hi, waves 4-5 label lo.
    auto feat_src_off = [&](int piece) -> unsigned {
        int qq = 64 * piece + lane;
        if (qq >= kTileR * 33) qq = 0;   // lanes past the image (piece 16, lanes 32-63): any valid source, lands in slack
        int row = qq / 33, ch = qq - row * 33;
        if (ch == 32) ch = 31;
        return (unsigned)(row * 512 + ch * 16);
    };
    const unsigned src_a = feat_src_off(wave), src_b = feat_src_off(wave + 8), src_c = feat_src_off(16);
    // ---- this workgroup's segments (common.h): runs of reference tiles, each against ONE target tile ----
    const int seg0 = A.seg_off[blockIdx.x], seg1 = A.seg_off[blockIdx.x + 1];
    for (int si = seg0; si < seg1; ++si) {
        const Segment sg = A.segs[si];
        const int tt = __builtin_amdgcn_readfirstlane(sg.tt);
        const int r_lo = __builtin_amdgcn_readfirstlane(sg.r_lo);
        const int n_steps = __builtin_amdgcn_readfirstlane(sg.n_steps);
        const int part_slot = __builtin_amdgcn_readfirstlane(sg.slot);

        // target (B operand) fragments: 32 columns x 256 channels per wave, resident in registers
        const int t = tt * kBT + wave * kColsPerWave + j;
        const int t_ld = t < A.HWp ? t : A.HWp - 1;
        const bf16_t* trow = A.feat_ring + ((size_t)A.target_slot * A.HWp + t_ld) * kC + h * 8;
        bf16x8 Bt[16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) Bt[ks] = *(const bf16x8*)(trow + ks * 16);
        // Consume the fragments HERE: hipcc otherwise sinks its counted waits for these 16 loads into the tile loop - a
        // `s_waitcnt vmcnt(15) ... vmcnt(0)` ladder in front of the MFMAs of EVERY iteration - and the hardware counter those
        // waits read also counts the LDS-DMA pieces in flight, so each score burst drained the pieces it had just issued.
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) asm volatile("" : "+v"(Bt[ks]));

        // target-side spatial channels for both sigmas and the per-column constants g*Q_t*c
        if (!PROB) {
            const int tq = t < A.HW ? t : A.HW - 1;
            const double at = (double)(tq / A.Wd), bt = (double)(tq % A.Wd);
            const double tw = A.two_over_w, gm = A.gamma;
            const double qt = at * at + tw * at * bt + gm * bt * bt;
#pragma unroll
            for (int sg = 0; sg < 2; ++sg) {
                const double g = sg ? A.g2 : A.g1;
                float ah, am, al, bh, bm, bl, kh, km, kl;
                split3((float)(g * (2.0 * at + tw * bt)), ah, am, al);
                split3((float)(g * (2.0 * gm * bt + tw * at)), bh, bm, bl);
                split3((float)(-g), kh, km, kl);
                // channels 0-7 (k-half 0): alpha x3, beta x3, kappa_h, kappa_m; channels 8-15 (k-half 1): kappa_l,
                // kappa_h, kappa_m, kappa_h, 0...  (pairs with the reference-side table of build_coord_table)
                bf16x8 B;
                B[0] = (bf16_t)(h ? kl : ah);
                B[1] = (bf16_t)(h ? kh : am);
                B[2] = (bf16_t)(h ? km : al);
                B[3] = (bf16_t)(h ? kh : bh);
                B[4] = (bf16_t)(h ? 0.0f : bm);
                B[5] = (bf16_t)(h ? 0.0f : bl);
                B[6] = (bf16_t)(h ? 0.0f : kh);
                B[7] = (bf16_t)(h ? 0.0f : km);
                s_bx[sg][tid] = B;
                s_kq[sg][tid] = (float)(g * qt * (double)c);
            }
        }

        ColState st;
        st.m = kNegBig;
        st.l = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) st.Y[r] = 0.0f;

        // Every thread runs the SAME stream, one step per reference tile:
        //     scores(p): stage tile p+3 (LDS-DMA pieces from inside the MFMA burst); label fragments of p; 16 MFMAs        | barrier
        //     softmax(p): prefetch the first fragments of tile p+1; (prior tile when the pixel tile or sigma changes;) exp / sums / label MFMAs | barrier
        // Group B runs one barrier late, so in every barrier interval one wave of a SIMD is in its MFMA burst and the
        // other in its softmax burst.  Tile t sits in ring slot t & 3: written during step t-3, read during steps t-1
        // (prefetch) and t.
        // The reference stream is walked PIXEL TILE BY PIXEL TILE WITH THE N FRAMES INNER: step index r = tile * N + frame.
        const int N = A.n_ref;
        const int slot_v = A.slot[lane];     // ring slot of sampled frame `lane` (kMaxRef = 64 = one per lane): per-tile frame
                                             // switches read it with v_readlane, no memory access
        int sn = 0, stile = 0;               // staging cursor (frame, pixel tile)
        auto stage_seek = [&](int step) {
            stile = (r_lo + step) / N;
            sn = (r_lo + step) - stile * N;
        };
        // Every wave issues exactly THREE pieces per tile (constant s_waitcnt count): two feature pieces and a third one by
        // role - wave 0 feature piece 16, wave 1 coordinates, waves 2-3 label hi, waves 4-5 label lo; waves without a third
        // piece of their own (6, 7, and 1 / 4 / 5 in the modes that have no such data) repeat their first feature piece.
        // The pieces are issued one at a time from inside the MFMA burst (stage_piece), not in a burst of their own: back to
        // back they cost ~150 cycles of issue each (the vector-memory queue fills), spread out ~40.
        typedef __attribute__((address_space(3))) void* lds_ptr;
        typedef const __attribute__((address_space(1))) void* glb_ptr;
        const unsigned smem_base = (unsigned)(size_t)(lds_ptr)smem;   // LDS byte address of the ring
        auto stage_piece = [&](int buf, int i) {   // i = 0, 1, 2
            const unsigned lds = __builtin_amdgcn_readfirstlane(smem_base + (unsigned)(buf * kLdsBuf));
            const int slot = __builtin_amdgcn_readlane(slot_v, sn);
            const unsigned char* f = (const unsigned char*)A.feat_ring + ((size_t)slot * A.HWp + (size_t)stile * kTileR) * (kC * 2);
            if (i == 0) {
                glds16(f + src_a, lds + wave * 1024);
            } else if (i == 1) {
                glds16(f + src_b, lds + (wave + 8) * 1024);
            } else if (wave == 0) {
                glds16(f + src_c, lds + 16 * 1024);
            } else if (wave == 1 && !PROB) {
                glds16((const unsigned char*)A.coord_tab + (size_t)stile * kLdsCoord + lane * 16, lds + kOffCoord);
            } else if (wave == 2 || wave == 3) {
                const unsigned char* lh = (const unsigned char*)A.lab_hi + ((size_t)slot * TPF + stile) * kLdsLab;
                glds16(lh + (wave - 2) * 1024 + lane * 16, lds + kOffLabHi + (wave - 2) * 1024);
            } else if ((wave == 4 || wave == 5) && LAB_LO) {
                const unsigned char* ll = (const unsigned char*)A.lab_lo + ((size_t)slot * TPF + stile) * kLdsLab;
                glds16(ll + (wave - 4) * 1024 + lane * 16, lds + kOffLabLo + (wave - 4) * 1024);
            } else {
                glds16(f + src_a, lds + wave * 1024);
            }
        };
        auto stage_advance = [&]() {   // next tile of the reference stream; stays on the last one at the end of the stream
            int nn = sn + 1, ns = stile;
            if (nn == N) {
                nn = 0;
                ns = stile + 1;
            }
            if (ns < TPF) {
                sn = nn;
                stile = ns;
            }
        };
        auto stage_issue = [&](int buf) {
            stage_piece(buf, 0);
            stage_piece(buf, 1);
            stage_piece(buf, 2);
            stage_advance();
        };

        // prologue: tiles 0, 1, 2 by everyone (past the end of a short segment: tiles nobody reads)
        stage_seek(0);
        stage_issue(0);
        stage_issue(1);
        stage_issue(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (grpB) __syncthreads();

        STAMP_DECL;
#ifdef VOSPROP_STAMP
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
#endif
        // compute cursor (pixel tile, frame) and the prior tile of the wave's 32 x 32 (reference pixels, target pixels) block:
        // w (dense mode) or log2 w (top-k passes) for the sigma of the frames being walked; recomputed when the pixel tile or the
        // sigma class changes (N = 9, frame_idx > 15: twice per 9 tiles)
        int ctile = r_lo / N, cn = r_lo - ctile * N;
        bool sparse = (A.sparse_mask >> cn) & 1ull;
        bool need_w = !PROB;
        float Wt[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) Wt[r] = 0.0f;

        // the deferred label product: probabilities and label fragments of the previous tile (zeros before the first one)
        float pa[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) pa[r] = 0.0f;
        LabFrag<LAB_LO> lab_prev;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            lab_prev.h0[e] = (bf16_t)0.0f; lab_prev.h1[e] = (bf16_t)0.0f;
            lab_prev.l0[e] = (bf16_t)0.0f; lab_prev.l1[e] = (bf16_t)0.0f;
        }

        AFrag<PROB> fr;
        fr.prefetch(smem, j, h);
#ifndef VOSPROP_PRIO_MODE
#define VOSPROP_PRIO_MODE 2   // 2 = score (MFMA) burst at priority 1: -3.5 % measured; 1 = younger wave group at priority 1: null; 3 = softmax burst at priority 1; 0 = off
#endif
        if (VOSPROP_PRIO_MODE == 1 && grpB) __builtin_amdgcn_s_setprio(1);
        for (int p = 0; p < n_steps; ++p) {
            // Tile t sits in ring slot t & 3.  Step p scores tile p, prefetches tile p+1, stages tile p+3 (three pieces per wave,
            // issued inside the MFMA burst) and ends by waiting for ITS pieces of tile p+2 - issued a whole step earlier, so the
            // wait is normally free (with a 3-slot ring and a wait for the pieces issued in the same step it cost ~230 cycles).
            const unsigned char* lb = smem + (p & 3) * kLdsBuf;
            const unsigned char* lbn = smem + ((p + 1) & 3) * kLdsBuf;
            const int b_st = (p + 3) & 3;
            STAMP_AT(0);   // bucket 0: loop overhead (+ time between segments)
            // ---- scores(p) ----
            LabFrag<LAB_LO> lab;
            if (MODE == 0) lab.load(lb, lane);
            STAMP_AT(1);   // 1: label reads
            f32x16 S;
            bf16x8 pk0, pk1;
            if (VOSPROP_PRIO_MODE == 2) __builtin_amdgcn_s_setprio(1);
            tile_scores<PROB>(lb, j, h, Bt, fr, S, [&](int ks) {
                if (!(VOSPROP_ABLATE & 4) && (ks == 2 || ks == 7 || ks == 12)) stage_piece(b_st, ks / 5);
                if (MODE == 0) {
                    // label product of tile p-1, VALU part, in the shadow of this tile's score MFMAs: a = p w, then bf16 packing
                    if (!PROB && ks < 8) {
                        pa[2 * ks] *= Wt[2 * ks];
                        pa[2 * ks + 1] *= Wt[2 * ks + 1];
                    }
                    if (ks == 9) pk0 = pack_half(pa, 0);
                    if (ks == 11) pk1 = pack_half(pa, 1);
                }
            });
            if (MODE == 0) label_mfmas<LAB_LO>(lab_prev, pk0, pk1, st.Y);
            if (!(VOSPROP_ABLATE & 4)) stage_advance();
            if (VOSPROP_PRIO_MODE == 2) __builtin_amdgcn_s_setprio(0);
            STAMP_AT(2);   // 2: MFMA chain
            if (!(VOSPROP_ABLATE & 8)) __syncthreads();
            STAMP_AT(3);   // 3: barrier 1
            // ---- softmax(p) ----
            fr.prefetch_lo(lbn, j, h);   // first fragments of tile p+1 (harmless when p+1 == n_steps)
            STAMP_AT(4);   // 4: prefetch issue   (5: max + rescale decision, 6: exps + sums inside tile_softmax)
            if (!PROB && need_w) {       // wave-uniform, 2 of 9 tiles at N = 9
                asm volatile("; prior tile" ::: "memory");
                prior_tile<false>(lb, j, h, s_bx[sparse ? 1 : 0][tid], c, s_kq[sparse ? 1 : 0][tid], Wt);
                need_w = false;
            }
            const bool tail = ragged && ctile == TPF - 1;
            if (VOSPROP_PRIO_MODE == 3) __builtin_amdgcn_s_setprio(1);
            tile_softmax<PROB>(h, S, pa, st, c, tail, rows_last STAMP_PASS);
            lab_prev = lab;
            if (VOSPROP_PRIO_MODE == 3) __builtin_amdgcn_s_setprio(0);
            STAMP_AT(7);   // 7: pack + label MFMAs
            fr.prefetch_hi(lbn, j, h);
            // this wave's pieces of tile p+2 have landed; the 3 of tile p+3 may stay in flight (loads return in order).  Top-k
            // pass 2 also issues atomics and stores, which share the counter: it waits for everything
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            if (++cn == N) {        // next pixel tile: a new prior tile
                cn = 0;
                ++ctile;
                need_w = !PROB;
            }
            {
                const bool sp = (A.sparse_mask >> cn) & 1ull;     // the next frame's sigma class (changes once per pixel tile)
                if (sp != sparse) need_w = !PROB;
                sparse = sp;
            }
            STAMP_AT(8);   // 8: DMA wait + frame bookkeeping
            // (barrier 2 was also tried BEFORE the tail - prefetch_hi, DMA wait, bookkeeping - to even out the two intervals of a
            // step: 5 % slower; the two wave groups slow each other down, the sum of the work matters more than the longer phase)
            if (!(VOSPROP_ABLATE & 8)) __syncthreads();
            STAMP_AT(9);   // 9: barrier 2
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the look-ahead pieces before the ring is re-staged
        if (!grpB) __syncthreads();
        if (MODE == 0) {   // the last tile's label product
            if (!PROB) {
#pragma unroll
                for (int r = 0; r < 16; ++r) pa[r] *= Wt[r];
            }
            label_mfmas<LAB_LO>(lab_prev, pack_half(pa, 0), pack_half(pa, 1), st.Y);
        }
#ifdef VOSPROP_STAMP
        if (A.dbg && lane == 0)
            for (int k = 0; k < VOSPROP_NSTAMP; ++k)
                atomicAdd(&A.dbg[((size_t)blockIdx.x * kWaves + wave) * VOSPROP_NSTAMP + k], tsum[k]);
#endif

        // ---- this segment's partial: rows (m, l, numerators[d]) x 256 columns ----
        float* part = A.part + ((size_t)part_slot * A.part_rows) * kBT + wave * kColsPerWave + j;
        const float lsum = half_sum(st.l);
        if (h == 0) {
            part[0] = st.m;
            part[kBT] = lsum;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cls = acc_row(r, h);
            if (cls < A.d) part[(size_t)(2 + cls) * kBT] = st.Y[r];
        }
    }
}

}  // namespace vosprop
