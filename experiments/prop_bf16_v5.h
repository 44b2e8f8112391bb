// prop_bf16_v5_kernel - the dense label-propagation path (PROB = false, one-hot labels, no top-k) re-built on what
// tools/ubench_issue.hip measured on MI355X (profiles/r01_ubench_simd_issue_model.txt):
//   * a plain VALU op costs ~4.9 issue cycles and a transcendental ~8.9 whatever the number of waves on the SIMD, and two
//     waves per SIMD interfere; but up to ~5 VALU ops per MFMA hide completely under a dependent MFMA chain when they are
//     interleaved IN THE SAME WAVE.
// So: ONE wave per SIMD (4 waves x 64 target pixels per workgroup, the whole 512-register file per wave), and every
// half-step is an explicitly ordered stream  { MFMA of unit u ; LDS refill ; one row of the softmax of unit u-1 } x 16
// (unit = reference tile x 32-pixel column block), pinned with sched_barrier.  The MFMAs are inline asm so that the
// 128 registers of target fragments live in AGPRs ("a" operands, loaded there straight from HBM) while everything the
// vector ALU touches stays in arch VGPRs - the builtin form made hipcc park accumulators in AGPRs and copy them back
// (experiments/prop_bf16_v4_4wave_pipelined.h.txt: 40 % of its VALU instructions were such copies).
// Inline-asm MFMAs get no hazard padding from the compiler (cdna guide section 5.7): the wait states are in the strings.
// Same staging (LDS-DMA, 528-B padded rows, 3-slot ring), work map, partial layout and maths as prop_bf16_kernel.
#pragma once
#include "prop_bf16.h"

namespace vosprop {

struct TagT { static constexpr bool value = true; };
struct TagF { static constexpr bool value = false; };

#ifndef V5_ABLATE
#define V5_ABLATE 0   // timing experiments: 1 no DMA, 2 no barrier, 4 no softmax rows, 8 no LDS fragment refills, 16 no MFMA
#endif

constexpr int kW5 = 4;   // waves per workgroup (one per SIMD); 64 target pixels = two 32-pixel column blocks per wave

// accumulate chain: D = C is the same register tuple -> back-to-back issue is legal
#define V5_MFMA_FIRST(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "a"(b))
#define V5_MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b))
// S_w = S + spatial term: reads the chain's result as C into ANOTHER tuple: 18 wait states after the producing MFMA
#define V5_MFMA_SW(dst, a, b, c)                                                                        \
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(dst) \
                 : "v"(a), "v"(b), "v"(c))
// label product: B operand was just written by v_cvt_pk (VALU -> MFMA operand wait states)
#define V5_MFMA_LAB(acc, a, b) \
    asm volatile("s_nop 4\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

struct V5Unit {
    f32x16 S, Sw;
};

struct V5Soft {          // running results of the row-by-row softmax of one unit
    float p[16];         // weighted probabilities (f32) until packed
    float l0, l1, emax;
    bf16x8 pk0, pk1;
};

// One row r of the softmax of a finished unit: 2 fma + 2 exp + 1 add (+ max / pack every other row).
__device__ __forceinline__ void v5_row(const V5Unit& u, V5Soft& s, int r, float c, float mc, float mq) {
    const float e = __builtin_fmaf(u.S[r], c, -mc);
    const float pe = __builtin_amdgcn_exp2f(e);
    if (r & 1) s.l1 += pe; else s.l0 += pe;
    s.p[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(u.Sw[r], c, -mq));
    s.emax = vmaxf(s.emax, e);
    if (r & 1) {
        if (r < 8) {
            s.pk0[r - 1] = (bf16_t)s.p[r - 1];
            s.pk0[r] = (bf16_t)s.p[r];
        } else {
            s.pk1[r - 9] = (bf16_t)s.p[r - 1];
            s.pk1[r - 8] = (bf16_t)s.p[r];
        }
    }
}

__global__ __launch_bounds__(kW5 * 64, 1) void prop_bf16_v5_kernel(const PropArgs A) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[kRing * kLdsBuf];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;
    const int h = lane >> 5;

    const int TPF = A.tiles_per_frame;
    const float c = A.c;
    const bool ragged = A.HW != A.HWp;
    const int rows_last = A.HW - (TPF - 1) * kTileR;

    // LDS-DMA roles with 4 waves: wave w issues feature pieces w, w+4, w+8, w+12 (wave 0 also piece 16);
    // wave 1 the coordinate piece, waves 2 / 3 the two label pieces (see prop_bf16.h for the padded image)
    auto feat_src_off = [&](int piece) -> unsigned {
        int qq = 64 * piece + lane;
        if (qq >= kTileR * 33) qq = 0;
        int row = qq / 33, ch = qq - row * 33;
        if (ch == 32) ch = 31;
        return (unsigned)(row * 512 + ch * 16);
    };
    unsigned src_off[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) src_off[i] = feat_src_off(wave + 4 * i < 17 ? wave + 4 * i : 0);

    const int seg0 = A.seg_off[blockIdx.x], seg1 = A.seg_off[blockIdx.x + 1];
    for (int si = seg0; si < seg1; ++si) {   // this workgroup's segments (common.h)
        const Segment sg = A.segs[si];
        const int tt = __builtin_amdgcn_readfirstlane(sg.tt);
        const int r_lo = __builtin_amdgcn_readfirstlane(sg.r_lo);
        const int n_steps = __builtin_amdgcn_readfirstlane(sg.n_steps);
        const int part_slot = __builtin_amdgcn_readfirstlane(sg.slot);

        // ---- target fragments -> AGPRs (straight from HBM), target-side spatial channels -> VGPRs ----
        bf16x8 Bt[2][16];
        bf16x8 Bx1[2], Bx2[2];
        float kq1[2], kq2[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int t = tt * kBT + wave * 64 + cb * 32 + j;
            const int t_ld = t < A.HWp ? t : A.HWp - 1;
            const bf16_t* trow = A.feat_ring + ((size_t)A.target_slot * A.HWp + t_ld) * kC + h * 8;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(Bt[cb][ks]) : "v"(trow + ks * 16) : "memory");
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
                bf16x8 B;
                B[0] = (bf16_t)(h ? kl : ah);
                B[1] = (bf16_t)(h ? kh : am);
                B[2] = (bf16_t)(h ? km : al);
                B[3] = (bf16_t)(h ? kh : bh);
                B[4] = (bf16_t)(h ? 0.0f : bm);
                B[5] = (bf16_t)(h ? 0.0f : bl);
                B[6] = (bf16_t)(h ? 0.0f : kh);
                B[7] = (bf16_t)(h ? 0.0f : km);
                if (sg) { Bx2[cb] = B; kq2[cb] = (float)(g * qt * (double)c); }
                else { Bx1[cb] = B; kq1[cb] = (float)(g * qt * (double)c); }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the AGPR loads above are invisible to hipcc's counters
        asm volatile("" : "+v"(kq1[0]), "+v"(kq1[1]), "+v"(kq2[0]), "+v"(kq2[1]));

        ColState st[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            st[cb].m = kNegBig;
            st[cb].l = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[cb].Y[r] = 0.0f;
        }

        // ---- staging ----
        int sn = r_lo / TPF, stile = r_lo - sn * TPF;
        const unsigned char *f_base = nullptr, *lh_base = nullptr;
        auto stage_frame = [&]() {
            const int slot = A.slot[sn];
            f_base = (const unsigned char*)A.feat_ring + (size_t)slot * A.HWp * kC * 2;
            lh_base = (const unsigned char*)A.lab_hi + (size_t)slot * TPF * kLdsLab;
        };
        auto stage_issue = [&](int buf) {
            typedef __attribute__((address_space(3))) void* lds_ptr;
            typedef const __attribute__((address_space(1))) void* glb_ptr;
            unsigned char* lds = smem + buf * kLdsBuf;
            const unsigned char* f = f_base + (size_t)stile * kGlbFeat;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds((glb_ptr)(f + src_off[i]), (lds_ptr)(lds + (wave + 4 * i) * 1024), 16, 0, 0);
            const unsigned char* lh = lh_base + (size_t)stile * kLdsLab + lane * 16;
            if (wave == 0)
                __builtin_amdgcn_global_load_lds((glb_ptr)(f + src_off[4]), (lds_ptr)(lds + 16 * 1024), 16, 0, 0);
            else if (wave == 1)
                __builtin_amdgcn_global_load_lds((glb_ptr)((const unsigned char*)A.coord_tab + (size_t)stile * kLdsCoord + lane * 16),
                                                 (lds_ptr)(lds + kOffCoord), 16, 0, 0);
            else if (wave == 2)
                __builtin_amdgcn_global_load_lds((glb_ptr)lh, (lds_ptr)(lds + kOffLabHi), 16, 0, 0);
            else
                __builtin_amdgcn_global_load_lds((glb_ptr)(lh + 1024), (lds_ptr)(lds + kOffLabHi + 1024), 16, 0, 0);
            if (++stile == TPF) {
                asm volatile("; next staged frame" ::: "memory");
                stile = 0;
                ++sn;
                if (sn < A.n_ref) stage_frame();
            }
        };
        stage_frame();
        stage_issue(0);
        if (n_steps > 1) stage_issue(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        int cn = r_lo / TPF, ctile = r_lo - cn * TPF;
        bool sparse = (A.sparse_mask >> cn) & 1ull;
        bool sparse_prev = sparse;

        AFrag<false> fr;
        fr.prefetch(smem, j, h);
        V5Unit u0, u1;                    // scores of column block 0 / 1
        LabFrag<false> labA, labB;        // label fragments of even / odd tiles
        int b_cur = 0, b_nxt = 1, b_st = 2;

        // 17 MFMAs of unit (tile in `lb`, column block CB) interleaved row by row with the softmax of `pu` (the previous
        // unit, column block PCB, its labels in `plab`); then that unit's rescale check + label MFMAs.
        auto half_step = [&](const unsigned char* lb, const unsigned char* nb, auto cb_tag, V5Unit& cu, const V5Unit& pu,
                             const LabFrag<false>& plab, bool cu_sparse, bool pu_sparse, auto sm_tag) {
            constexpr int CB = decltype(cb_tag)::value ? 1 : 0, PCB = 1 - CB;
            constexpr bool SM = decltype(sm_tag)::value;
            const unsigned char* arow = lb + j * kRowB + h * 16;
            const unsigned char* nrow = nb + j * kRowB + h * 16;
            ColState& ps = st[PCB];
            const float pkq = pu_sparse ? kq2[PCB] : kq1[PCB];
            const float mc = ps.m * c, mq = mc + pkq;
            V5Soft sf;
            sf.l0 = 0.0f;
            sf.l1 = 0.0f;
            sf.emax = -3.0e38f;
#ifdef V5_SAFE
            asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#endif
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                if (!(V5_ABLATE & 16)) {
                    if (ks == 0) V5_MFMA_FIRST(cu.S, fr.a[0], Bt[CB][0]);
                    else V5_MFMA(cu.S, fr.a[ks & 7], Bt[CB][ks]);
                } else if (ks == 0) {
                    V5_MFMA_FIRST(cu.S, fr.a[0], Bt[CB][0]);
                }
                if (!(V5_ABLATE & 8))
                    fr.a[ks & 7] = ks < 8 ? *(const bf16x8*)(arow + (ks + 8) * 32) : *(const bf16x8*)(nrow + (ks - 8) * 32);
                if (SM && !(V5_ABLATE & 4)) v5_row(pu, sf, ks, c, mc, mq);
                __builtin_amdgcn_sched_barrier(0);
            }
            V5_MFMA_SW(cu.Sw, fr.ax, cu_sparse ? Bx2[CB] : Bx1[CB], cu.S);
            fr.ax = *(const bf16x8*)(nb + kOffCoord + h * 512 + j * 16);
            if (SM) {
                // pin the row results here: hipcc otherwise sinks the exps below the rescale branch, out of the MFMA shadow
                asm volatile("" : "+v"(sf.pk0), "+v"(sf.pk1), "+v"(sf.l0), "+v"(sf.l1), "+v"(sf.emax));
                if (__any(sf.emax > kRescaleThr)) {
                    // rare: raise the running max (shared by the half-waves of a column), rescale (l, Y) once, redo the unit
                    asm volatile("; rescale" ::: "memory");
                    float sr[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) sr[r] = pu.S[r];
                    float mn = fmaxf(ps.m, half_max(max16v(sr)));
                    asm volatile("" : "+v"(mn));
                    const float scl = __builtin_amdgcn_exp2f((ps.m - mn) * c);
                    ps.l *= scl;
#pragma unroll
                    for (int r = 0; r < 16; ++r) ps.Y[r] *= scl;
                    ps.m = mn;
                    const float mc2 = mn * c, mq2 = mc2 + pkq;
                    sf.l0 = 0.0f;
                    sf.l1 = 0.0f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) v5_row(pu, sf, r, c, mc2, mq2);
                }
                ps.l += sf.l0 + sf.l1;
                V5_MFMA_LAB(ps.Y, plab.h0, sf.pk0);
                V5_MFMA_LAB(ps.Y, plab.h1, sf.pk1);
            }
            __builtin_amdgcn_sched_barrier(0);
        };

        auto tile_step = [&](int p, LabFrag<false>& lab_cur, const LabFrag<false>& lab_prev, auto sm_tag) {
            const unsigned char* lb = smem + b_cur * kLdsBuf;
            const unsigned char* lbn = smem + b_nxt * kLdsBuf;
            if (!(V5_ABLATE & 1) && p + 2 < n_steps) stage_issue(b_st);
            lab_cur.load(lb, lane);
            const bool tail = ragged && ctile == TPF - 1;
            half_step(lb, lb, TagF(), u0, u1, lab_prev, sparse, sparse_prev, sm_tag);     // scores(p,0) || softmax(p-1,1)
            if (!decltype(sm_tag)::value)   // no label MFMAs followed the S_w MFMA: give it its 64 cycles before u0 is read
                asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
            if (tail) {
                asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t; tail tile" ::: "memory");   // MFMA results must have landed
                mask_tail_rows(u0.S, u0.Sw, h, rows_last);
            }
            half_step(lb, lbn, TagT(), u1, u0, lab_cur, sparse, sparse, TagT());          // scores(p,1) || softmax(p,0)
            if (tail) {
                asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t; tail tile" ::: "memory");
                mask_tail_rows(u1.S, u1.Sw, h, rows_last);
            }
            sparse_prev = sparse;
            if (++ctile == TPF) {
                asm volatile("; next scored frame" ::: "memory");
                ctile = 0;
                ++cn;
                sparse = (A.sparse_mask >> cn) & 1ull;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile p+2 have landed
            if (!(V5_ABLATE & 2)) __syncthreads();
            const int tmp = b_cur;
            b_cur = b_nxt;
            b_nxt = b_st;
            b_st = tmp;
        };
        tile_step(0, labA, labB, TagF());
        for (int p = 1; p < n_steps; ++p) {   // label fragment sets alternate (even tiles labA, odd tiles labB)
            if (p & 1) tile_step(p, labB, labA, TagT());
            else tile_step(p, labA, labB, TagT());
        }
        // softmax of the very last unit (last tile, block 1): there are no scores left to hide it under
        auto finish = [&](const LabFrag<false>& last_lab) {
            ColState& ps = st[1];
            const float pkq = sparse_prev ? kq2[1] : kq1[1];
            asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
            float sr[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) sr[r] = u1.S[r];
            const float mn = fmaxf(ps.m, half_max(max16v(sr)));
            const float scl = __builtin_amdgcn_exp2f((ps.m - mn) * c);
            ps.l *= scl;
#pragma unroll
            for (int r = 0; r < 16; ++r) ps.Y[r] *= scl;
            ps.m = mn;
            V5Soft sf;
            sf.l0 = 0.0f;
            sf.l1 = 0.0f;
            sf.emax = -3.0e38f;
#pragma unroll
            for (int r = 0; r < 16; ++r) v5_row(u1, sf, r, c, mn * c, mn * c + pkq);
            ps.l += sf.l0 + sf.l1;
            V5_MFMA_LAB(ps.Y, last_lab.h0, sf.pk0);
            V5_MFMA_LAB(ps.Y, last_lab.h1, sf.pk1);
            asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        };
        if ((n_steps - 1) & 1) finish(labB);
        else finish(labA);

        // ---- this segment's partial ----
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const float lsum = half_sum(st[cb].l);
            float* part = A.part + ((size_t)part_slot * A.part_rows) * kBT
                          + wave * 64 + cb * 32 + j;
            if (h == 0) {
                part[0] = st[cb].m;
                part[kBT] = lsum;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cls = acc_row(r, h);
                if (cls < A.d) part[(size_t)(2 + cls) * kBT] = st[cb].Y[r];
            }
        }
    }
}

}  // namespace vosprop
