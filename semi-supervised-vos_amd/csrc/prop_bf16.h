// The hot kernel: fused affinity -> online column softmax -> spatial prior -> label product.
//
// Replaces reference src/model/predict.py:46-70 (mm, *=temperature, softmax(dim=0), spatial weights,
// label mm) without ever forming the (N*HW) x HW affinity.
//
// Mapping onto CDNA4 (gfx950):
//   * S^T tile = mfma(A = 32 reference pixels x K, B = K x 32 target pixels): every lane owns ONE target
//     pixel (column) and 16 reference pixels (rows) of the 32x32 tile, so the column softmax statistics are
//     in-register; the two half-waves share a column and exchange one value per tile (v_permlane32_swap).
//   * K = 256 feature channels (16 x v_mfma_f32_32x32x16_bf16) + ONE extra 16-deep MFMA whose channels carry
//     the Gaussian spatial prior:  -dist^2/(sigma^2 tau)  is bilinear in (reference coords, target coords)
//     [dist^2 = Qp + Qt - a_p(2a_t + 2b_t/W) - b_p(2 gamma b_t + 2a_t/W)], so S_w = S + X comes out of the
//     matrix core; 3-way bf16 splits of the real-valued factors keep ~24 significant bits.
//   * numerators  out[k,t] = sum_r L[k,r] a[r,t]  are a second MFMA: the weighted probabilities a (already
//     laid out rows-in-registers / column-on-lane) are packed to bf16 and used as the B operand against the
//     label matrix stored in HBM in A-operand order.  Denominators stay f32 on the VALU.
//   * one workgroup = 8 waves x 32 target pixels = 256 target pixels (B fragments live in registers for the
//     whole kernel); reference tiles (32 pixels x 512 B, XOR-swizzled) stream through a double-buffered LDS.
//   * partial (m, l, numerators) per (target tile, reference chunk) go to HBM; combine_kernel merges them.
#pragma once
#include "common.h"

namespace vosprop {

constexpr int kLdsFeat = kTileR * kC * 2;        // 16384
constexpr int kLdsCoord = 2 * 32 * 16;           // 1024
constexpr int kLdsLab = 2 * 64 * 16;             // 2048
constexpr int kLdsBuf = kLdsFeat + kLdsCoord + 2 * kLdsLab;   // 21504
constexpr float kRescaleThr = 8.0f;              // defer-max threshold in log2 units (p <= 2^8)
constexpr float kNegBig = -1.0e30f;

__device__ __forceinline__ float other_half(float x) {
    // value held by lane ^ 32
    return __shfl_xor(x, 32);
}

template <bool PROB, bool LAB_LO>
__global__ __launch_bounds__(kWaves * 64, 2) void prop_bf16_kernel(const PropArgs A) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * kLdsBuf];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;     // MFMA row (A operand) / column (B operand, C/D)
    const int h = lane >> 5;     // k-half of the operand fragments / row-half of the accumulator

    const int U = A.n_ref * A.row_splits;
    const int tt = blockIdx.x / U;
    const int u = blockIdx.x - tt * U;
    const int n = u / A.row_splits;
    const int rs = u - n * A.row_splits;
    const int tile_begin = rs * A.tiles_per_split;
    int tile_end = tile_begin + A.tiles_per_split;
    if (tile_end > A.tiles_per_frame) tile_end = A.tiles_per_frame;

    const int slot = A.slot[n];
    const bool sparse = (A.sparse_mask >> n) & 1ull;
    const size_t frame_tiles = (size_t)slot * A.tiles_per_frame;
    const unsigned char* feat_base = (const unsigned char*)(A.feat_ring + (size_t)slot * A.HWp * kC);
    const unsigned char* coord_base = (const unsigned char*)A.coord_tab;
    const unsigned char* labhi_base = (const unsigned char*)A.lab_hi + frame_tiles * kLdsLab;
    const unsigned char* lablo_base = LAB_LO ? (const unsigned char*)A.lab_lo + frame_tiles * kLdsLab : nullptr;

    // ---- target (B operand) fragments: 32 columns x 256 channels per wave, resident in registers ----
    int t = tt * kBT + wave * kColsPerWave + j;
    const int t_ld = t < A.HWp ? t : A.HWp - 1;
    const bf16_t* trow = A.feat_ring + ((size_t)A.target_slot * A.HWp + t_ld) * kC + h * 8;
    bf16x8 Bt[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) Bt[ks] = *(const bf16x8*)(trow + ks * 16);

    // ---- target-side spatial channels (one fragment for this chunk's sigma) and the per-column constant ----
    bf16x8 Bx;
    float kq = 0.0f;   // g * Q_t * c
    if (!PROB) {
        const double g = sparse ? A.g2 : A.g1;
        const int tq = t < A.HW ? t : A.HW - 1;
        const double at = (double)(tq / A.Wd), bt = (double)(tq % A.Wd);
        const double tw = A.two_over_w, gm = A.gamma;
        const float alpha = (float)(g * (2.0 * at + tw * bt));
        const float beta = (float)(g * (2.0 * gm * bt + tw * at));
        const float kappa = (float)(-g);
        kq = (float)(g * (at * at + tw * at * bt + gm * bt * bt) * (double)A.c);
        float ah, am, al, bh, bm, bl, kh, km, kl;
        split3(alpha, ah, am, al);
        split3(beta, bh, bm, bl);
        split3(kappa, kh, km, kl);
        // channels 0-7 (k-half 0): alpha x3, beta x3, kappa_h, kappa_m ; channels 8-15 (k-half 1): kappa_l, kappa_h,
        // kappa_m, kappa_h, 0...  (pairs with the reference-side table built by build_coord_table)
        Bx[0] = (bf16_t)(h ? kl : ah);
        Bx[1] = (bf16_t)(h ? kh : am);
        Bx[2] = (bf16_t)(h ? km : al);
        Bx[3] = (bf16_t)(h ? kh : bh);
        Bx[4] = (bf16_t)(h ? 0.0f : bm);
        Bx[5] = (bf16_t)(h ? 0.0f : bl);
        Bx[6] = (bf16_t)(h ? 0.0f : kh);
        Bx[7] = (bf16_t)(h ? 0.0f : km);
    }

    // ---- staging: global -> registers -> LDS (XOR-swizzled feature rows) ----
    uint4 g0, g1, g2;
    const int q0 = tid, q1 = tid + 512;
    const int row0 = q0 >> 5, ch0 = q0 & 31, row1 = q1 >> 5, ch1 = q1 & 31;
    const int dst0 = row0 * 512 + ((ch0 ^ (row0 & 15)) << 4);
    const int dst1 = row1 * 512 + ((ch1 ^ (row1 & 15)) << 4);
    // waves 0: coord (64 chunks), 1-2: label hi (128 chunks), 3-4: label lo (128 chunks)
    const int aux_kind = wave == 0 ? 0 : (wave <= 2 ? 1 : (wave <= 4 ? 2 : 3));
    const int aux_idx = aux_kind == 0 ? tid : (aux_kind == 1 ? tid - 64 : tid - 192);
    const int aux_dst = (aux_kind == 0 ? kLdsFeat : (aux_kind == 1 ? kLdsFeat + kLdsCoord : kLdsFeat + kLdsCoord + kLdsLab))
                        + aux_idx * 16;

    auto stage_load = [&](int tile) {
        const unsigned char* f = feat_base + (size_t)tile * kLdsFeat;
        g0 = *(const uint4*)(f + q0 * 16);
        g1 = *(const uint4*)(f + q1 * 16);
        if (aux_kind == 0) {
            if (!PROB) g2 = *(const uint4*)(coord_base + (size_t)tile * kLdsCoord + aux_idx * 16);
        } else if (aux_kind == 1) {
            g2 = *(const uint4*)(labhi_base + (size_t)tile * kLdsLab + aux_idx * 16);
        } else if (aux_kind == 2) {
            if (LAB_LO) g2 = *(const uint4*)(lablo_base + (size_t)tile * kLdsLab + aux_idx * 16);
        }
    };
    auto stage_write = [&](int buf) {
        unsigned char* b = smem + buf * kLdsBuf;
        *(uint4*)(b + dst0) = g0;
        *(uint4*)(b + dst1) = g1;
        if ((aux_kind == 0 && !PROB) || aux_kind == 1 || (aux_kind == 2 && LAB_LO)) *(uint4*)(b + aux_dst) = g2;
    };

    // ---- running statistics (per lane: its column, its half's rows) ----
    float m = kNegBig;   // running max of raw S (shared by both halves of a column)
    float l = 0.0f;      // partial denominator: sum over this half's rows of 2^((S - m) c)
    f32x16 Y;            // numerators: rows = classes, col = this lane's column
#pragma unroll
    for (int r = 0; r < 16; ++r) Y[r] = 0.0f;
    const float c = A.c;

    if (tile_begin < tile_end) {
        stage_load(tile_begin);
        stage_write(0);
    }
    __syncthreads();

    int buf = 0;
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        const bool has_next = tile + 1 < tile_end;
        if (has_next) stage_load(tile + 1);

        const unsigned char* lb = smem + buf * kLdsBuf;
        const unsigned char* arow = lb + j * 512;
        f32x16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const bf16x8 a = *(const bf16x8*)(arow + (((2 * ks + h) ^ (j & 15)) << 4));
            S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, Bt[ks], S, 0, 0, 0);
        }
        f32x16 Sw;
        if (!PROB) {
            const bf16x8 ax = *(const bf16x8*)(lb + kLdsFeat + h * 512 + j * 16);
            Sw = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax, Bx, S, 0, 0, 0);
        }
        // padded reference rows of the frame's last tile must not enter the softmax
        if (tile == A.tiles_per_frame - 1 && A.HW != A.HWp) {
            const int r0 = tile * kTileR;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (r0 + acc_row(r, h) >= A.HW) {
                    S[r] = kNegBig;
                    if (!PROB) Sw[r] = kNegBig;
                }
            }
        }

        // ---- online softmax: tile max, deferred rescale ----
        float tmax = S[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, S[r]);
        tmax = fmaxf(tmax, other_half(tmax));
        if (__any((tmax - m) * c > kRescaleThr)) {
            const float mn = fmaxf(m, tmax);
            const float sc = __builtin_amdgcn_exp2f((m - mn) * c);
            l *= sc;
#pragma unroll
            for (int r = 0; r < 16; ++r) Y[r] *= sc;
            m = mn;
        }
        const float mc = m * c;
        const float mq = mc + kq;
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r], c, -mc));
            l += pe;
            p[r] = PROB ? pe : __builtin_amdgcn_exp2f(__builtin_fmaf(Sw[r], c, -mq));
        }
        bf16x8 pk0, pk1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            pk0[e] = (bf16_t)p[e];
            pk1[e] = (bf16_t)p[8 + e];
        }
        // ---- numerators: Y[class, t] += L[class, rows] . a[rows, t] ----
        const unsigned char* lh = lb + kLdsFeat + kLdsCoord + lane * 16;
        Y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(lh), pk0, Y, 0, 0, 0);
        Y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(lh + 1024), pk1, Y, 0, 0, 0);
        if (LAB_LO) {
            Y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(lh + kLdsLab), pk0, Y, 0, 0, 0);
            Y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(lh + kLdsLab + 1024), pk1, Y, 0, 0, 0);
        }

        if (has_next) stage_write(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    // ---- write this unit's partial: rows (m, l, numerators[d]) x 256 columns ----
    const float lsum = l + other_half(l);
    float* part = A.part + ((size_t)blockIdx.x * (2 + A.d)) * kBT + wave * kColsPerWave + j;
    if (h == 0) {
        part[0] = m;
        part[kBT] = lsum;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int cls = acc_row(r, h);
        if (cls < A.d) part[(size_t)(2 + cls) * kBT] = Y[r];
    }
}

}  // namespace vosprop
