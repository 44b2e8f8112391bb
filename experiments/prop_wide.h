// The dense label-mode propagation kernel with 64 target columns per wave and ONE wave per SIMD (4 waves, 512-register budget).
//
// Same arithmetic, LDS tile image, work decomposition (256 target pixels per workgroup, Segment table) and partial format as
// prop_dense.h - another shape of the same in-wave pipeline.  Why: two waves of a SIMD share its
// vector issue, the 8-wave kernel is issue-bound with the SIMD issuing in 87 % of its cycles and the older wave of every SIMD waiting
// a quarter of each step for the younger one.  Here a wave owns TWO 32-column blocks: one A fragment (one ds_read_b128) feeds two
// independent MFMAs, the per-tile bookkeeping (cursor, staging offsets, barrier, waits) is paid once per 32 MFMAs instead of once
// per 16, and nothing competes for the issue port:
//
//     gap ks of step p:  S0 += A[ks] B0[ks] ;  S1 += A[ks] B1[ks]      (2 x 32 pipe cycles, 2 x 8 issue cycles)
//                        row ks of the softmax of tile p-1 for both column blocks: 2 x {fma, exp, mul} + 2 x 1/2 {max3, cvt_pk}
//                        = ~44 issue cycles, the fragment refill, and in five gaps one LDS-DMA piece of tile p+3
//
// Built for plain label propagation (one-hot labels, no low label part; the two NEED_L forms of prop_dense.h); every other mode keeps
// prop_dense_kernel.  Selected by engine.hip launch_prop_mode (VOSPROP_DENSE_WIDE).
#pragma once
#include "common.h"
#include "prop_bf16.h"
#include "prop_dense.h"

#ifndef VOSPROP_WABLATE
#define VOSPROP_WABLATE 0   // timing experiments only (results wrong): 2 = no LDS fragment refills, 4 = no staging
#endif

namespace vosprop {

constexpr int kWWaves = 4;          // waves per workgroup: one per SIMD
constexpr int kWCols = 64;          // target pixels per wave: two 32-column MFMA blocks
static_assert(kWWaves * kWCols == kBT, "same target tile as the 8-wave kernels");

// One gap of the wide kernel's chain as ONE asm statement (fixed order, compiler-allocated registers): the two score MFMAs of k-slice
// ks and three software-pipelined stages of the previous tile's softmax, for both column blocks -
//     row ks:   e = P c - m c          row ks-1:  q = 2^e          row ks-2:  w = q W
// Each MFMA is followed by the three VALU instructions of ITS column block (24 issue cycles with the MFMA's own 8: inside its 32-cycle
// shadow); two MFMAs back to back would make the in-order wave wait 32 cycles for the pipe with nothing issued meanwhile (measured:
// 96 cycles per pair that way).  One statement so that the order is the source's, not the scheduler's.
template <bool FIRST>
__device__ __forceinline__ void wide_gap(f32x16& S0, f32x16& S1, const bf16x8& a, const bf16x8& b0, const bf16x8& b1, float p0,
                                         float p1, float c, float nmc0, float nmc1, float& e0n, float& e1n, float e0p, float e1p,
                                         float& q0n, float& q1n, float q0p, float q1p, float W0r, float W1r, float& w0, float& w1) {
    if (FIRST) {
        asm volatile(
            "v_mfma_f32_32x32x16_bf16 %0, %8, %9, 0\n\t"
            "v_fma_f32 %2, %11, %13, %14\n\t"
            "v_exp_f32 %4, %16\n\t"
            "v_mul_f32 %6, %18, %20\n\t"
            "v_mfma_f32_32x32x16_bf16 %1, %8, %10, 0\n\t"
            "v_fma_f32 %3, %12, %13, %15\n\t"
            "v_exp_f32 %5, %17\n\t"
            "v_mul_f32 %7, %19, %21"
            : "=&v"(S0), "=&v"(S1), "=&v"(e0n), "=&v"(e1n), "=&v"(q0n), "=&v"(q1n), "=&v"(w0), "=&v"(w1)
            : "v"(a), "a"(b0), "a"(b1), "v"(p0), "v"(p1), "s"(c), "v"(nmc0), "v"(nmc1), "v"(e0p), "v"(e1p), "v"(q0p), "v"(q1p),
              "v"(W0r), "v"(W1r));
    } else {
        asm volatile(
            "v_mfma_f32_32x32x16_bf16 %0, %8, %9, %0\n\t"
            "v_fma_f32 %2, %11, %13, %14\n\t"
            "v_exp_f32 %4, %16\n\t"
            "v_mul_f32 %6, %18, %20\n\t"
            "v_mfma_f32_32x32x16_bf16 %1, %8, %10, %1\n\t"
            "v_fma_f32 %3, %12, %13, %15\n\t"
            "v_exp_f32 %5, %17\n\t"
            "v_mul_f32 %7, %19, %21"
            : "+v"(S0), "+v"(S1), "=&v"(e0n), "=&v"(e1n), "=&v"(q0n), "=&v"(q1n), "=&v"(w0), "=&v"(w1)
            : "v"(a), "a"(b0), "a"(b1), "v"(p0), "v"(p1), "s"(c), "v"(nmc0), "v"(nmc1), "v"(e0p), "v"(e1p), "v"(q0p), "v"(q1p),
              "v"(W0r), "v"(W1r));
    }
}

template <bool NEED_L>
__global__ __launch_bounds__(kWWaves * 64, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void prop_wide_kernel(const PropArgs A) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[kRing5 * kLdsBuf];
    __shared__ __attribute__((aligned(16))) bf16x8 s_bx[2][2][kWWaves * 64];   // [sigma][column block][thread]
    __shared__ float s_kq[2][2][kWWaves * 64];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;
    const int h = lane >> 5;
    const int TPF = A.tiles_per_frame;
    const int N = A.n_ref;
    const float c = A.c;
    const bool ragged = A.HW != A.HWp;
    const int rows_last = A.HW - (TPF - 1) * kTileR;

    // ---- staging roles: 20 pieces per tile (17 feature pieces of the padded 528-B row image, coordinates, 2 label pieces), five
    // per wave: feature pieces w, w+4, w+8, w+12 and a fifth chosen once per wave (wave 0: feature piece 16, 1: coordinates,
    // 2 / 3: the label halves) as (base, per-slot stride, per-tile stride, LDS offset)
    auto feat_src_off = [&](int piece) -> unsigned {
        int qq = 64 * piece + lane;
        if (qq >= kTileR * 33) qq = 0;
        int row = qq / 33, ch = qq - row * 33;
        if (ch == 32) ch = 31;
        return (unsigned)(row * 512 + ch * 16);
    };
    const unsigned src0 = feat_src_off(wave), src1 = feat_src_off(wave + 4), src2 = feat_src_off(wave + 8), src3 = feat_src_off(wave + 12);
    const size_t feat_slot_stride = (size_t)A.HWp * (kC * 2);
    const unsigned char* fifth_base = (const unsigned char*)A.feat_ring;
    size_t fifth_slot_stride = feat_slot_stride;
    unsigned fifth_tile_stride = kGlbFeat, fifth_lane = feat_src_off(16), fifth_lds = 16 * 1024;
    if (wave == 1) {
        fifth_base = (const unsigned char*)A.coord_tab;
        fifth_slot_stride = 0;
        fifth_tile_stride = kLdsCoord;
        fifth_lane = lane * 16;
        fifth_lds = kOffCoord;
    } else if (wave >= 2) {
        fifth_base = (const unsigned char*)A.lab_hi + (wave - 2) * 1024;
        fifth_slot_stride = (size_t)TPF * kLdsLab;
        fifth_tile_stride = kLdsLab;
        fifth_lane = lane * 16;
        fifth_lds = kOffLabHi + (wave - 2) * 1024;
    }
    fifth_tile_stride = __builtin_amdgcn_readfirstlane(fifth_tile_stride);
    fifth_lds = __builtin_amdgcn_readfirstlane(fifth_lds);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const unsigned smem_base = (unsigned)(size_t)(lds_ptr)smem;
    const unsigned my_slot = (unsigned)A.slot[lane];
    const unsigned fo = my_slot * (unsigned)feat_slot_stride;
    const unsigned to = my_slot * (unsigned)fifth_slot_stride;
    const unsigned char* const feat_base = (const unsigned char*)A.feat_ring;

    const int seg0 = A.seg_off[blockIdx.x], seg1 = A.seg_off[blockIdx.x + 1];
    for (int si = seg0; si < seg1; ++si) {
        const Segment sg = A.segs[si];
        const int tt = __builtin_amdgcn_readfirstlane(sg.tt);
        const int r_lo = __builtin_amdgcn_readfirstlane(sg.r_lo);
        const int n_steps = __builtin_amdgcn_readfirstlane(sg.n_steps);
        const int part_slot = __builtin_amdgcn_readfirstlane(sg.slot);

        int tid_l = tid, wd_l = A.Wd;
        asm volatile("" : "+v"(tid_l), "+s"(wd_l));       // segment-start values are not hoisted (and spilled): see prop_dense.h
        const int j_l = tid_l & 31, h_l = (tid_l >> 5) & 1;

        // target (B operand) fragments: 2 blocks x 32 columns x 256 channels per wave, resident
        bf16x8 Bt0[16], Bt1[16];
        {
            const int t0 = tt * kBT + wave * kWCols + j_l, t1 = t0 + 32;
            const int a0 = t0 < A.target_rows ? t0 : A.target_rows - 1, a1 = t1 < A.target_rows ? t1 : A.target_rows - 1;
            const bf16_t* r0 = A.target_feat + (size_t)a0 * kC + h_l * 8;
            const bf16_t* r1 = A.target_feat + (size_t)a1 * kC + h_l * 8;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                Bt0[ks] = *(const bf16x8*)(r0 + ks * 16);
                Bt1[ks] = *(const bf16x8*)(r1 + ks * 16);
            }
        }
        // target-side prior constants for both sigmas and both column blocks -> LDS (own lane writes, own lane reads)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int t = tt * kBT + wave * kWCols + cb * 32 + j_l;
            const int tq = t < A.HW ? t : A.HW - 1;
            const int trow_i = tq / wd_l;
            const double at = (double)trow_i, bt = (double)(tq - trow_i * wd_l);
            const double tw = A.two_over_w, gm = A.gamma;
            const double qt = at * at + tw * at * bt + gm * bt * bt;
#pragma unroll
            for (int sgm = 0; sgm < 2; ++sgm) {
                int sg_o = sgm;
                asm volatile("" : "+s"(sg_o));
                const double g = sg_o ? A.g2 : A.g1;
                float ah, am, al, bh, bm, bl, kh, km, kl;
                split3((float)(g * (2.0 * at + tw * bt)), ah, am, al);
                split3((float)(g * (2.0 * gm * bt + tw * at)), bh, bm, bl);
                split3((float)(-g), kh, km, kl);
                bf16x8 B;
                B[0] = (bf16_t)(h_l ? kl : ah);
                B[1] = (bf16_t)(h_l ? kh : am);
                B[2] = (bf16_t)(h_l ? km : al);
                B[3] = (bf16_t)(h_l ? kh : bh);
                B[4] = (bf16_t)(h_l ? 0.0f : bm);
                B[5] = (bf16_t)(h_l ? 0.0f : bl);
                B[6] = (bf16_t)(h_l ? 0.0f : kh);
                B[7] = (bf16_t)(h_l ? 0.0f : km);
                s_bx[sgm][cb][tid_l] = B;
                s_kq[sgm][cb][tid_l] = (float)(g * qt * (double)c);
            }
        }

        float m0 = kNegBig, m1 = kNegBig, l0 = 0.0f, l1 = 0.0f;
        f32x16 Y0, Y1;
        float W0[16], W1[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            Y0[r] = 0.0f;
            Y1[r] = 0.0f;
            W0[r] = 0.0f;
            W1[r] = 0.0f;
        }

        // ---- staging cursor (frame inner) ----
        int sn = 0, stile = 0;
        unsigned so_feat = 0, so_fifth = 0;
        auto stage_bases = [&]() __attribute__((always_inline)) {
            so_feat = (unsigned)__builtin_amdgcn_readlane((int)fo, sn) + (unsigned)stile * (unsigned)kGlbFeat;
            so_fifth = (unsigned)__builtin_amdgcn_readlane((int)to, sn) + (unsigned)stile * fifth_tile_stride;
        };
        auto stage_piece = [&](unsigned lds, int i) __attribute__((always_inline)) {
            if (i == 0) glds16s2(src0, so_feat, feat_base, lds, (unsigned)wave * 1024);
            else if (i == 1) glds16s2(src1, so_feat, feat_base, lds, ((unsigned)wave + 4) * 1024);
            else if (i == 2) glds16s2(src2, so_feat, feat_base, lds, ((unsigned)wave + 8) * 1024);
            else if (i == 3) glds16s2(src3, so_feat, feat_base, lds, ((unsigned)wave + 12) * 1024);
            else glds16s2(fifth_lane, so_fifth, fifth_base, lds, fifth_lds);
        };
        auto stage_advance = [&]() __attribute__((always_inline)) {
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
        float zf = 0.0f;
        asm volatile("" : "+v"(zf));
        if (tid_l < kLdsLab / 16) *(f32x4*)(smem + kRingLast + kOffLabHi + tid_l * 16) = f32x4{zf, zf, zf, zf};   // "tile -1" labels
        stile = r_lo / N;
        sn = r_lo - stile * N;
        for (int q = 0; q < 3; ++q) {
            stage_bases();
#pragma unroll
            for (int i = 0; i < 5; ++i) stage_piece(smem_base + q * kLdsBuf, i);
            stage_advance();
        }
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) asm volatile("" : "+a"(Bt0[ks]), "+a"(Bt1[ks]));   // the 128 B registers live in AGPRs (MFMA reads them there)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        STAMP_DECL;
#ifdef VOSPROP_STAMP
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
#endif
        int ctile = r_lo / N, cn = r_lo - ctile * N;
        bool sparse = (A.sparse_mask >> cn) & 1ull;
        bool need_w = true;

        AFrag<false> fr;
        fr.prefetch(smem, j, h);

        f32x16 Sa0, Sa1, Sb0, Sb1;   // scores of "this" and "the previous" tile, two column blocks each
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            Sb0[r] = -__builtin_inff();
            Sb1[r] = -__builtin_inff();
        }
        int s_cur = 0, s_nxt = kLdsBuf, s_prv = kRingLast, s_stg = 3 * kLdsBuf;
        auto ring_advance = [&]() __attribute__((always_inline)) {
            s_prv = s_cur;
            s_cur = s_nxt;
            s_nxt = s_nxt == kRingLast ? 0 : s_nxt + kLdsBuf;
            s_stg = s_stg == kRingLast ? 0 : s_stg + kLdsBuf;
        };

        // one column block of a finished tile: rescale (rare, wave-uniform branch taken for both blocks), label MFMAs
        auto rescale_block = [&](const f32x16& Sp, const float (&Wt)[16], float& m, float& l, f32x16& Y, bf16x8& pk0, bf16x8& pk1)
                                 __attribute__((always_inline)) {
            float sv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) sv[r] = Sp[r];
            const float mn = fmaxf(m, half_max(max16v(sv)));
            const float sc = __builtin_amdgcn_exp2f((m - mn) * c);
            l *= sc;
#pragma unroll
            for (int r = 0; r < 16; ++r) Y[r] *= sc;
            m = mn;
            float lt0, lt1;
            softmax_rows<false>(Sp, Wt, c, mn * c, lt0, lt1, pk0, pk1);
            return lt0 + lt1;
        };

        auto step = [&](f32x16& S0, f32x16& S1, const f32x16& P0, const f32x16& P1) __attribute__((always_inline)) {
            const unsigned char* lb = smem + s_cur;
            const unsigned char* lbn = smem + s_nxt;
            const unsigned b_st = smem_base + (unsigned)s_stg;
            stage_bases();
            LabFrag<false> labp;
            const float mc0 = m0 * c, mc1 = m1 * c;
            float a0 = NEED_L ? 0.0f : kNegBig, a1 = NEED_L ? 0.0f : kNegBig;   // alarm accumulators (sum of terms / max of scores)
            float b0 = 0.0f, b1 = 0.0f;                                          // NEED_L: second partial sums
            bf16x8 pk00, pk01, pk10, pk11;
            const unsigned char* arow = lb + j * kRowB + h * 16;
            const unsigned char* nrow = lbn + j * kRowB + h * 16;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                S0[r] = 0.0f;
                S1[r] = 0.0f;
            }
#ifdef VOSPROP_STAMP
            STAMP_AT(0);
#endif
            // The softmax rows of tile p-1 are SOFTWARE-PIPELINED over the gaps (wide_gap): gap ks holds the fma of row ks, the
            // exponential of row ks-1 and the prior multiply of row ks-2; the packing of rows ks-4 / ks-3 and the alarm follow it.
            const float nmc0 = -mc0, nmc1 = -mc1;
            float e0 = 0.0f, e1 = 0.0f, q0 = 0.0f, q1 = 0.0f;      // pipeline registers (row ks-1's exponent, row ks-2's term)
            float w0[16], w1[16];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
#ifdef VOSPROP_STAMP
                if (ks == 8) STAMP_AT(1);
#endif
                float e0n, e1n, q0n, q1n, w0r, w1r;
                const float W0r = ks >= 2 ? W0[ks - 2] : 0.0f, W1r = ks >= 2 ? W1[ks - 2] : 0.0f;
                if (ks == 0) wide_gap<true>(S0, S1, fr.a[0], Bt0[0], Bt1[0], P0[0], P1[0], c, nmc0, nmc1, e0n, e1n, e0, e1, q0n, q1n, q0, q1, W0r, W1r, w0r, w1r);
                else wide_gap<false>(S0, S1, fr.a[ks & 7], Bt0[ks], Bt1[ks], P0[ks], P1[ks], c, nmc0, nmc1, e0n, e1n, e0, e1, q0n, q1n, q0, q1, W0r, W1r, w0r, w1r);
                if (ks >= 2) {
                    w0[ks - 2] = w0r;
                    w1[ks - 2] = w1r;
                    if (NEED_L) {      // q0 / q1 still hold row ks-2's terms here
                        if (ks & 1) { b0 += q0; b1 += q1; }
                        else { a0 += q0; a1 += q1; }
                    }
                }
                e0 = e0n; e1 = e1n;
                q0 = q0n; q1 = q1n;
#if !(VOSPROP_WABLATE & 2)
                if (ks < 8) fr.a[ks] = *(const bf16x8*)(arow + (ks + 8) * 32);
                else fr.a[ks - 8] = *(const bf16x8*)(nrow + (ks - 8) * 32);
#endif
#if !(VOSPROP_WABLATE & 4)
                if (ks % 3 == 1) stage_piece(b_st, ks / 3);       // ks = 1, 4, 7, 10, 13 -> pieces 0..4
#endif
                if (ks == 10) labp.load(smem + s_prv, lane);
                if (!NEED_L && (ks & 1)) {
                    a0 = __builtin_fmaxf(__builtin_fmaxf(a0, P0[ks - 1]), P0[ks]);
                    a1 = __builtin_fmaxf(__builtin_fmaxf(a1, P1[ks - 1]), P1[ks]);
                }
                if (ks >= 4 && !(ks & 1)) {          // rows ks-4 (even), ks-3 (odd)
                    const int r = ks - 4;
                    if (r < 8) {
                        pk00[r] = (bf16_t)w0[r]; pk00[r + 1] = (bf16_t)w0[r + 1];
                        pk10[r] = (bf16_t)w1[r]; pk10[r + 1] = (bf16_t)w1[r + 1];
                    } else {
                        pk01[r - 8] = (bf16_t)w0[r]; pk01[r - 7] = (bf16_t)w0[r + 1];
                        pk11[r - 8] = (bf16_t)w1[r]; pk11[r - 7] = (bf16_t)w1[r + 1];
                    }
                }
            }
            {   // drain the pipeline: exponential of row 15, terms of rows 14 and 15, the last two packings
                const float q0l = __builtin_amdgcn_exp2f(e0), q1l = __builtin_amdgcn_exp2f(e1);
                w0[14] = q0 * W0[14];
                w1[14] = q1 * W1[14];
                w0[15] = q0l * W0[15];
                w1[15] = q1l * W1[15];
                if (NEED_L) {
                    a0 += q0; a1 += q1;
                    b0 += q0l; b1 += q1l;
                }
                pk01[4] = (bf16_t)w0[12]; pk01[5] = (bf16_t)w0[13]; pk01[6] = (bf16_t)w0[14]; pk01[7] = (bf16_t)w0[15];
                pk11[4] = (bf16_t)w1[12]; pk11[5] = (bf16_t)w1[13]; pk11[6] = (bf16_t)w1[14]; pk11[7] = (bf16_t)w1[15];
            }
            asm volatile("" : "+v"(pk00), "+v"(pk01), "+v"(pk10), "+v"(pk11));
#ifdef VOSPROP_STAMP
            STAMP_AT(2);
#endif
            stage_advance();
            // finish tile p-1
            const bool alarm = NEED_L ? (a0 + b0 > kSumThrV3 || a1 + b1 > kSumThrV3)
                                      : (a0 > m0 + kAlarmExp / c || a1 > m1 + kAlarmExp / c);
            float lt0 = a0 + b0, lt1 = a1 + b1;
            if (__any(alarm)) {
                asm volatile("; rescale" ::: "memory");
                lt0 = rescale_block(P0, W0, m0, l0, Y0, pk00, pk01);
                lt1 = rescale_block(P1, W1, m1, l1, Y1, pk10, pk11);
            }
            if (NEED_L) {
                l0 += lt0;
                l1 += lt1;
            }
            Y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(labp.h0, pk00, Y0, 0, 0, 0);
            Y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(labp.h0, pk10, Y1, 0, 0, 0);
            Y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(labp.h1, pk01, Y0, 0, 0, 0);
            Y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(labp.h1, pk11, Y1, 0, 0, 0);
#ifdef VOSPROP_STAMP
            STAMP_AT(3);
#endif
            if (ragged && ctile == TPF - 1) {
                asm volatile("; tail tile" ::: "memory");
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (acc_row(r, h) >= rows_last) {
                        S0[r] = kNegBig;
                        S1[r] = kNegBig;
                    }
            }
            if (need_w) {
                asm volatile("; prior tile" ::: "memory");
                const int sp = sparse ? 1 : 0;
                prior_tile<false>(lb, j, h, s_bx[sp][0][tid], c, s_kq[sp][0][tid], W0);
                prior_tile<false>(lb, j, h, s_bx[sp][1][tid], c, s_kq[sp][1][tid], W1);
                need_w = false;
            }
            if (++cn == N) {
                cn = 0;
                ++ctile;
                need_w = true;
            }
            {
                const bool sp = (A.sparse_mask >> cn) & 1ull;
                if (sp != sparse) need_w = true;
                sparse = sp;
            }
#ifdef VOSPROP_STAMP
            STAMP_AT(4);
#endif
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");   // this wave's pieces of tile p+2 have landed (the 5 of p+3 may fly)
#ifdef VOSPROP_STAMP
            STAMP_AT(5);
#endif
            __syncthreads();
#ifdef VOSPROP_STAMP
            STAMP_AT(6);
#endif
            ring_advance();
        };

        int p = 0;
        for (; p + 1 < n_steps; p += 2) {
            step(Sa0, Sa1, Sb0, Sb1);
            step(Sb0, Sb1, Sa0, Sa1);
        }
        auto drain = [&](const f32x16& P0, const f32x16& P1) __attribute__((always_inline)) {
            LabFrag<false> labp;
            labp.load(smem + s_prv, lane);
            bf16x8 pk00, pk01, pk10, pk11;
            float x0, x1, y0, y1;
            softmax_rows<false>(P0, W0, c, m0 * c, x0, x1, pk00, pk01);
            softmax_rows<false>(P1, W1, c, m1 * c, y0, y1, pk10, pk11);
            float lt0 = x0 + x1, lt1 = y0 + y1;
            float sv0[16], sv1[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sv0[r] = P0[r];
                sv1[r] = P1[r];
            }
            // (both forms: the alarm of the last tile looks at the scores)
            if (__any(max16v(sv0) > m0 + kAlarmExp / c || max16v(sv1) > m1 + kAlarmExp / c)) {
                lt0 = rescale_block(P0, W0, m0, l0, Y0, pk00, pk01);
                lt1 = rescale_block(P1, W1, m1, l1, Y1, pk10, pk11);
            }
            if (NEED_L) {
                l0 += lt0;
                l1 += lt1;
            }
            Y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(labp.h0, pk00, Y0, 0, 0, 0);
            Y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(labp.h0, pk10, Y1, 0, 0, 0);
            Y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(labp.h1, pk01, Y0, 0, 0, 0);
            Y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(labp.h1, pk11, Y1, 0, 0, 0);
        };
        if (p < n_steps) {
            step(Sa0, Sa1, Sb0, Sb1);
            drain(Sa0, Sa1);
        } else {
            drain(Sb0, Sb1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#ifdef VOSPROP_STAMP
        if (A.dbg && lane == 0)
            for (int k = 0; k < VOSPROP_NSTAMP; ++k)
                atomicAdd(&A.dbg[((size_t)blockIdx.x * kWaves + wave) * VOSPROP_NSTAMP + k], tsum[k]);
#endif

        // ---- this segment's partial: rows (m, l, numerators[d]) x 256 columns ----
        int tid_e = tid;
        asm volatile("" : "+v"(tid_e));
        float* part = A.part + ((size_t)part_slot * A.part_rows) * kBT + wave * kWCols + (tid_e & 31);
        int hh = (tid_e >> 5) & 1;
        asm volatile("" : "+v"(part), "+v"(hh));
        const float ls0 = half_sum(l0), ls1 = half_sum(l1);
        if (hh == 0) {
            part[0] = m0;
            part[32] = m1;
            part[kBT] = ls0;
            part[kBT + 32] = ls1;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cls = acc_row(r, hh);
            if (cls < A.d) {
                part[(size_t)(2 + cls) * kBT] = Y0[r];
                part[(size_t)(2 + cls) * kBT + 32] = Y1[r];
            }
        }
    }
}

}  // namespace vosprop
