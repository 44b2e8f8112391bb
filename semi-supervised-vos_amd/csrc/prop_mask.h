// The mask-only dense propagation kernel (label mode, prediction not requested: the step of the frame loop, bench.py's timed
// kernel).  Same arithmetic as prop_dense_kernel - reference src/model/predict.py:46-70 followed by the arg-max of
// src/utils/inference_utils.py:70 - same feature ring, segment table and partial format; the TILE LOOP is a hand-ordered instruction
// stream on v_mfma_f32_16x16x32_bf16 (tools/gen_mask_loop.py -> prop_mask_loop.inc), entered from this file as ONE asm statement per
// segment with a fixed register map.  What differs from the hipcc-scheduled loop of prop_dense.h, and why
// (profiles/r04_mask_kernel_ablations.txt):
//
//   * The weighted exponent comes out of the matrix core.  The first score MFMA of an accumulator takes LM as its C operand, where
//       LM[r, t] = log2 w[r, t] - M_t           (w = the spatial prior, M_t = the column's running reference level, log2 units)
//     is itself the output of one MFMA (reference-side coordinates x target-side constants, as in prop_bf16.h, with three more K
//     channels carrying the 3-way bf16 split of -(g Q_t c + M_t) and one that sends the padded rows of a frame's last tile to -1e30),
//     rebuilt only when the pixel tile or the sigma class changes.  The target fragments are pre-multiplied by
//     c = temperature * log2(e), so the accumulator IS  x = s c + log2 w - M  and the weighted probability is one v_exp_f32 of it:
//     no fma, no prior block, no tail-tile masking, no accumulator zeroing in the loop.
//   * No cursor arithmetic.  A per-step control table (one entry per lane, refilled every 64 steps from a table the prologue builds
//     in LDS) holds the LDS-DMA source offsets of tile q+3 and the flags of tile q+1; a step fetches it with two v_readlane.
//   * Every wave stages (2 or 3 pieces per step, scalar address arithmetic only), one barrier per step - at the step end for waves
//     0-3, after gap 7 for waves 4-7, so the two waves of a SIMD sit half a step apart - and nothing behind the chain: the label
//     MFMAs of tile q-2 sit in gaps 1 and 2 of chain q, the overflow alarm of tile q-2 was decided in step q-1 and is ONE branch at
//     the step boundary.
//   * One statement per segment.  The loop leaves its steady state only at step boundaries (alarm, control table block exhausted,
//     segment end); the rescale path and the table refill are straight-line code of the same statement (hipcc could not keep 200
//     hard-bound registers in place across a C++ loop: it parked them in scratch around every re-entry).  The segment's prologue
//     and its last two tiles are C++ below.
//   * 16x16x32, not 32x32x16.  The chip is POWER-bound under this kernel: the bare score-MFMA chain alone takes 144 us of a 194 us
//     launch on random data (115 us on zeros), and the 16x16x32 shape moves half the accumulator bytes per MAC - the same chain as
//     16x16x32 instructions ran in 127 us, the whole kernel 2.3 % (480p) to 6.8 % (720p) faster on one box.
//
// Alarm / reference level: a = 2^x stays finite up to x = 127 and is packed to bf16 (same exponent range), Y accumulates in f32;
// the loop leaves when a weighted exponent exceeds kMaskAlarm = 100 (57 780 terms of 2^100 still fit f32), the column's level M is
// raised by its maximum, Y is rescaled ONCE and the pending tile redone (cdna guide T13: everything still at the old level is
// scaled exactly once - Y, the pending tile's weights, the next tile's accumulator which already holds the old LM, and LM).  The
// first tile of a segment always takes that path (it sets M from data, either sign), so afterwards max x >= 0 by construction:
// nothing that matters can underflow.  tests/test_gpu_parity.py forces the path with peaky features.
//
// Layouts (MFMA 16x16x32: A[row l&15][k = 8 (l>>4) + j], B[k = 8 (l>>4) + j][col l&15], D col = l&15, row = 4 (l>>4) + reg):
//   * a wave owns 32 target columns = two column blocks cb of 16; a lane holds columns 16 cb + (l & 15), cb = 0, 1, and k block
//     kb = l >> 4; target fragments B[cb][ks] = channels 32 ks + 8 kb .. + 7 of its column (64 registers, pre-multiplied by c);
//   * a reference tile = 32 rows = two row blocks rb; A(rb, ks) = one ds_read_b128 at row (16 rb + (l & 15)), channels
//     32 ks + 8 kb, from 544-B padded LDS rows (16-B slot = (2 row + kb) mod 16: conflict-free in every 16-lane group);
//   * S[rb][cb] = 4 registers: rows 16 rb + 4 kb + i of column 16 cb + (l & 15); the lane's 8 values of a column are rows
//     {4 kb + i} and {16 + 4 kb + i}: packed to bf16 they ARE the B operand (k = 8 kb + j) of the label MFMA
//     Y[cb] (16 classes x 16 columns) += L (16 classes x 32 rows) pk[cb], with the label fragment stored in that row order
//     (lab16 ring, aux_kernels.h lab16_row) - ONE 16-cycle MFMA per column block instead of two 32-cycle ones, d <= 16;
//   * the prior tile LM[rb][cb] = coordinates(rb) x target-side constants[cb]: K = 32 with channels 16..31 zero on the target side
//     (k blocks 2, 3), so whatever finite bytes those lanes read on the reference side do not matter.
#pragma once
#include "common.h"
#include "prop_bf16.h"
#ifndef VOSPROP_MASK_LOOP_INC
#define VOSPROP_MASK_LOOP_INC "prop_mask_loop.inc"      // (tools/mask_variants.sh builds timing variants from other generator outputs)
#endif
#include VOSPROP_MASK_LOOP_INC

namespace vosprop {

typedef __attribute__((ext_vector_type(16))) unsigned u32x16;
typedef __attribute__((ext_vector_type(8))) unsigned u32x8;
typedef __attribute__((ext_vector_type(8))) float f32x8;

// Two f32 -> two bf16 (round to nearest even) in one dword.  Plain casts ON PURPOSE: hipcc emits v_cvt_pk_bf16_f32 for them AND pads
// the hazard when an operand comes straight out of a transcendental (v_exp_f32 -> VALU needs wait states on gfx950); an `asm`
// v_cvt_pk right behind a compiler-placed v_exp_f32 read garbage (negative "probabilities" in the segment's last tile).
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const bf16x2 v = {(bf16_t)lo, (bf16_t)hi};
    return __builtin_bit_cast(unsigned, v);
}

constexpr int kMaskTabCap = 2048;                          // entries the LDS control table holds

constexpr int kMaskSlot = VOSPROP_MASK_SLOT;
constexpr int kMaskRing = VOSPROP_MASK_NSLOT;
constexpr int kMaskOffCoord = VOSPROP_MASK_OFF_COORD;
constexpr int kMaskOffLab = VOSPROP_MASK_OFF_LAB;
constexpr int kMaskRowB = VOSPROP_MASK_ROWB;
constexpr float kMaskAlarm = VOSPROP_MASK_ALARM;
constexpr int kMaskAhead = VOSPROP_MASK_AHEAD;
constexpr int kMaskTabBlock = VOSPROP_MASK_TAB_BLOCK;
constexpr int kMaskTabEntry = VOSPROP_MASK_TAB_ENTRY;
constexpr int kMaskMaxSteps = kMaskTabCap - kMaskTabBlock - kMaskAhead;      // (engine.hip falls back to prop_dense_kernel beyond that)
constexpr int kMaskLab = 1024;                               // bytes of one tile's label fragment (16 classes x 32 rows bf16)
constexpr int kMaskMaxClasses = 16;
static_assert(kMaskOffCoord == kTileR * kMaskRowB && kMaskOffLab == kMaskOffCoord + kLdsCoord && kMaskSlot == kMaskOffLab + kMaskLab, "slot layout");

// debug hook (tools/mask_stamps.py): 100 MHz wall clock of a workgroup's phases; a null A.dbg costs one scalar compare per stamp
#define VOSPROP_MASK_STAMP(k)                                                                        \
    do {                                                                                             \
        if (A.dbg && tid == 0) {                                                                     \
            unsigned long long t_;                                                                   \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");         \
            if ((k) < 2 || si == seg0 || (k) >= 6) A.dbg[(size_t)blockIdx.x * 8 + (k)] = t_;       \
        }                                                                                            \
    } while (0)

__global__ __launch_bounds__(kWaves * 64, 2) void prop_mask_kernel(const PropArgs A) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[kMaskRing * kMaskSlot];
    __shared__ unsigned s_off[2 * kMaxRef];      // per sampled frame: byte offset of its slot in the feature ring / lab16 ring
    __shared__ __attribute__((aligned(16))) unsigned s_tab[kMaskTabCap * (kMaskTabEntry / 4)];      // control table (prop_mask.h)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int TPF = A.tiles_per_frame;
    const int N = A.n_ref;
    const float c = A.c;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const unsigned smem_base = (unsigned)(size_t)(lds_ptr)smem;

    if (tid < kMaxRef) {
        const unsigned sl = (unsigned)A.slot[tid < N ? tid : 0];
        s_off[tid] = sl * (unsigned)((size_t)A.HWp * (kC * 2));
        s_off[kMaxRef + tid] = sl * (unsigned)(TPF * kMaskLab);
    }

    // ---- staging roles: every wave stages feature pieces w and w + 8 of the padded 544-B row image (17 KiB = 17 pieces, all of
    // them full); waves 0-3 (role A) a third piece each - feature piece 16, the coordinates, the label fragment (waves 2 AND 3: the
    // same bytes to the same place; the role-A stream has three pieces)
    auto feat_src_off = [&](int piece) -> unsigned {
        const int qq = 64 * piece + lane;
        const int row = qq / 34;
        int ch = qq - row * 34;
        if (ch >= 32) ch = 31;      // the two pad chunks of a row: any valid source
        return (unsigned)(row * 512 + ch * 16);
    };
    const bool role_a = wave < kWaves / 2;
    const unsigned role_b = ((unsigned)wave >> 2) & 1u;      // (kWaves == 8)
    const unsigned src_a = feat_src_off(wave), src_b = feat_src_off(wave + 8);
    const unsigned char* const feat_base = (const unsigned char*)A.feat_ring;
    const unsigned char* third_base = feat_base;
    unsigned src_3 = (unsigned)lane * 16, lds3_off = 16 * 1024;
    int third_col = 0;      // which offset the third piece follows: 0 feature tile, 1 coordinate tile, 2 label tile
    if (wave == 0) {
        src_3 = feat_src_off(16);
    } else if (wave == 1) {
        third_base = (const unsigned char*)A.coord_tab;
        lds3_off = kMaskOffCoord;
        third_col = 1;
    } else if (wave == 2 || wave == 3) {
        third_base = (const unsigned char*)A.lab16;
        lds3_off = kMaskOffLab;
        third_col = 2;
    }
    third_col = __builtin_amdgcn_readfirstlane(third_col);
    lds3_off = (unsigned)__builtin_amdgcn_readfirstlane((int)lds3_off);
    const unsigned fb_lo = (unsigned)(size_t)feat_base, fb_hi = (unsigned)((size_t)feat_base >> 32);
    const unsigned tb_lo = __builtin_amdgcn_readfirstlane((unsigned)(size_t)third_base);
    const unsigned tb_hi = __builtin_amdgcn_readfirstlane((unsigned)((size_t)third_base >> 32));
    const unsigned ldsa = smem_base + (unsigned)wave * 1024;
    const unsigned lds3 = __builtin_amdgcn_readfirstlane(smem_base + lds3_off);
    const unsigned tab_base = (unsigned)(size_t)(lds_ptr)s_tab;
    const unsigned tab_col = third_col == 0 ? 12u : third_col == 1 ? 4u : 8u;      // this wave's TB column inside an entry

    const int seg0 = A.seg_off[blockIdx.x], seg1 = A.seg_off[blockIdx.x + 1];
    Segment sg_next = A.seg_first[blockIdx.x];      // (a copy of segs[seg0]: it loads beside the range, not behind it)
    {
        const int si = seg0;
        VOSPROP_MASK_STAMP(0);      // 0: kernel entry (after the kernel arguments and the segment range)
    }
    __syncthreads();      // s_off
    for (int si = seg0; si < seg1; ++si) {
        const Segment sg = sg_next;
        if (si + 1 < seg1) sg_next = A.segs[si + 1];      // the next record flies under this segment (one memory latency off its start)
        const int tt = __builtin_amdgcn_readfirstlane(sg.tt);
        const int r_lo = __builtin_amdgcn_readfirstlane(sg.r_lo);
        const int n_steps = __builtin_amdgcn_readfirstlane(sg.n_steps);
        const int part_slot = __builtin_amdgcn_readfirstlane(sg.slot);

        VOSPROP_MASK_STAMP(1);      // 1: the first segment's record is here
        int tid_l = tid;
        asm volatile("" : "+v"(tid_l));
        const int lane_l = tid_l & 63, j16 = tid_l & 15, kb = (tid_l >> 4) & 3;

        // ---- target (B operand) fragments [cb][ks]: loads fly under the rest of the prologue
        int tcol[2];
        u32x4 Braw[16];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            tcol[cb] = tt * kBT + wave * kColsPerWave + 16 * cb + j16;
            const int t_ld = tcol[cb] < A.target_rows ? tcol[cb] : A.target_rows - 1;
            const bf16_t* trow = A.target_feat + (size_t)t_ld * kC + kb * 8;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) Braw[cb * 8 + ks] = *(const u32x4*)(trow + ks * 32);
        }
        // target-side constants of the prior MFMA [sigma][cb]: k blocks 0 / 1 hold K channels 0-7 / 8-15 (engine.hip
        // build_target_consts), k blocks 2 / 3 are zero; g Q_t c per (sigma, column)
        u32x4 cbv[2][2];
        float kq[2][2];
#pragma unroll
        for (int sgm = 0; sgm < 2; ++sgm)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const int tq = tcol[cb] < A.HW ? tcol[cb] : A.HW - 1;
                const u32x4 v = ((const u32x4*)A.tc_b)[((size_t)sgm * A.HWp + tq) * 2 + (kb & 1)];
                cbv[sgm][cb] = kb < 2 ? v : u32x4{0u, 0u, 0u, 0u};
                kq[sgm][cb] = A.tc_kq[(size_t)sgm * A.HWp + tq];
            }

        // "tile -1" takes its labels from the last slot: zero them (0 x NaN)
        // (the zero is made HERE, opaquely: as a loop invariant hipcc kept the zero vector in scratch, and its reload's vmcnt(0) held
        // the first tiles' LDS-DMA back until the target fragments had landed)
        float zero = 0.0f;
        asm volatile("" : "+v"(zero));
        if (tid_l < kMaskLab / 16) *(f32x4*)(smem + (kMaskRing - 1) * kMaskSlot + kMaskOffLab + tid_l * 16) = f32x4{zero, zero, zero, zero};

        // ---- tiles 0 .. kMaskAhead-1 -> slots 0 .. kMaskAhead-1, issued BEFORE the control table is built (their source offsets are
        // computed here directly): the table's arithmetic and its barrier then run under the memory latency of these pieces and
        // of the target fragments
#pragma unroll
        for (int i = 0; i < kMaskAhead; ++i) {
            const int r = r_lo + (i < n_steps - 1 ? i : n_steps - 1);
            const int tile = r / N, n = r - tile * N;
            const unsigned ea = (unsigned)__builtin_amdgcn_readfirstlane((int)(s_off[n] + (unsigned)tile * (unsigned)kGlbFeat));
            const unsigned el = (unsigned)__builtin_amdgcn_readfirstlane((int)(s_off[kMaxRef + n] + (unsigned)tile * (unsigned)kMaskLab));
            const unsigned eb = third_col == 0 ? ea : third_col == 1 ? (unsigned)tile * (unsigned)kLdsCoord : el;
            const unsigned lds = smem_base + (unsigned)i * kMaskSlot;
            glds16s2(src_a, ea, feat_base, lds, (unsigned)wave * 1024);
            glds16s2(src_b, ea, feat_base, lds, ((unsigned)wave + 8) * 1024);
            if (role_a) glds16s2(src_3, eb, third_base, lds, lds3_off);
        }

        // ---- control table of the segment -> LDS (prop_mask.h has the entry format)
        for (int p0 = tid_l; p0 < n_steps + kMaskTabBlock + kMaskAhead; p0 += kWaves * 64) {
            const int p = p0 < n_steps - 1 ? p0 : n_steps - 1;
            const int r = r_lo + p;
            const int tile = r / N, n = r - tile * N;
            const unsigned fo = s_off[n] + (unsigned)tile * (unsigned)kGlbFeat;
            int p2 = p0 - (kMaskAhead - 1);
            if (p2 > n_steps - 1) p2 = n_steps - 1;
            unsigned flags = 0;
            if (p2 >= 1) {
                const int r2 = r_lo + p2;
                const int tile2 = r2 / N, n2 = r2 - tile2 * N;
                const unsigned sp = (unsigned)(A.sparse_mask >> n2) & 1u;
                const unsigned sp_prev = n2 > 0 ? (unsigned)(A.sparse_mask >> (n2 - 1)) & 1u : 0u;
                flags = ((n2 == 0 || sp != sp_prev) ? 1u : 0u) | (sp << 1);
            }
            *(u32x4*)(s_tab + p0 * (kMaskTabEntry / 4)) =
                u32x4{fo | flags, (unsigned)tile * (unsigned)kLdsCoord, s_off[kMaxRef + n] + (unsigned)tile * (unsigned)kMaskLab, fo};
        }
        __syncthreads();
        const unsigned tab_a = s_tab[lane_l * (kMaskTabEntry / 4)];
        const unsigned tab_b = s_tab[lane_l * (kMaskTabEntry / 4) + tab_col / 4];

#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(Braw[i]));
        VOSPROP_MASK_STAMP(2);      // 2: everything of the prologue is issued, the control table is built
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (target fragments, constants and the first tiles' pieces)
        // the next segment's record has landed too: into scalar registers now (as four vector registers it lived across the loop
        // statement - in scratch, with a dependent scratch load at every segment start)
        sg_next.tt = __builtin_amdgcn_readfirstlane(sg_next.tt);
        sg_next.r_lo = __builtin_amdgcn_readfirstlane(sg_next.r_lo);
        sg_next.n_steps = __builtin_amdgcn_readfirstlane(sg_next.n_steps);
        sg_next.slot = __builtin_amdgcn_readfirstlane(sg_next.slot);
        VOSPROP_MASK_STAMP(3);      // 3: ... and has landed
        // c folded into the target side: T' = bf16(c T) (f16 features are converted on the way, one rounding); Bq = [cb][ks] x 4
        u32x16 Bq0, Bq1, Bq2, Bq3;
#define VOSPROP_MASK_SET(idx, val)                          \
    do {                                                   \
        if ((idx) < 16) Bq0[(idx) & 15] = (val);           \
        else if ((idx) < 32) Bq1[(idx) & 15] = (val);      \
        else if ((idx) < 48) Bq2[(idx) & 15] = (val);      \
        else Bq3[(idx) & 15] = (val);                      \
    } while (0)
        if (A.target_f16) {
#pragma unroll
            for (int f = 0; f < 16; ++f)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned w = Braw[f][i];
                    const float lo = (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xFFFFu));
                    const float hi = (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16));
                    VOSPROP_MASK_SET(f * 4 + i, cvt_pk_bf16(lo * c, hi * c));
                }
        } else {
#pragma unroll
            for (int f = 0; f < 16; ++f)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned w = Braw[f][i];
                    VOSPROP_MASK_SET(f * 4 + i, cvt_pk_bf16(__uint_as_float(w << 16) * c, __uint_as_float(w & 0xFFFF0000u) * c));
                }
        }
#undef VOSPROP_MASK_SET
        __syncthreads();      // the first tiles and the zeroed label area are visible

        // ---- inputs of the loop statement ----
        u32x16 CTL, CB;
        f32x8 AUX;
#pragma unroll
        for (int r = 0; r < 16; ++r) CTL[r] = 0u;
#pragma unroll
        for (int sgm = 0; sgm < 2; ++sgm)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int i = 0; i < 4; ++i) CB[8 * sgm + 4 * cb + i] = cbv[sgm][cb][i];
        const unsigned row_lo = smem_base + (unsigned)(j16 * kMaskRowB + kb * 16);
        const unsigned coord_lo = smem_base + (unsigned)((kb & 1) * 512 + j16 * 16);
        CTL[VOSPROP_MASK_CTL_SRCA] = src_a;
        CTL[VOSPROP_MASK_CTL_SRCB] = src_b;
        CTL[VOSPROP_MASK_CTL_SRC3] = src_3;
        CTL[VOSPROP_MASK_CTL_ROWLO] = row_lo;
        CTL[VOSPROP_MASK_CTL_ROWHI] = row_lo + 3u * kMaskSlot;
        CTL[VOSPROP_MASK_CTL_LANELO] = smem_base + (unsigned)lane_l * 16;
        CTL[VOSPROP_MASK_CTL_LANEHI] = smem_base + (unsigned)lane_l * 16 + 3u * kMaskSlot;
        CTL[VOSPROP_MASK_CTL_COORDLO] = coord_lo;
        CTL[VOSPROP_MASK_CTL_COORDHI] = coord_lo + 3u * kMaskSlot;
        CTL[VOSPROP_MASK_CTL_TA] = tab_a;
        CTL[VOSPROP_MASK_CTL_TB] = tab_b;
        CTL[VOSPROP_MASK_CTL_TABA] = (unsigned)lane_l * kMaskTabEntry;
        CTL[VOSPROP_MASK_CTL_TABB] = (unsigned)lane_l * kMaskTabEntry + tab_col;
#pragma unroll
        for (int r = 0; r < 8; ++r) AUX[r] = 0.0f;
#pragma unroll
        for (int sgm = 0; sgm < 2; ++sgm)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) AUX[VOSPROP_MASK_AUX_KQ + 2 * sgm + cb] = kq[sgm][cb];
        const int n0 = r_lo - (r_lo / N) * N;
        const unsigned sp0 = (unsigned)__builtin_amdgcn_readfirstlane((int)((A.sparse_mask >> n0) & 1ull));

        // a wave whose 32 target columns all lie beyond the map (last target tile: 236 of 256 columns at 480p, 3.4 % of the launch's
        // MFMA work) runs the STAGING-ONLY form of its role: same LDS-DMA pieces, waits and barriers, nothing else
        const bool pad_wave = tt * kBT + wave * kColsPerWave >= A.HW;
        const unsigned role = role_b | (pad_wave ? 2u : 0u);
        VOSPROP_MASK_STAMP(4);      // 4: target fragments scaled, loop inputs assembled
        // ---- the tile loop, its control and its rare paths: ONE statement per segment ----
        f32x16 S0, S1;
        f32x8 Y;
        u32x16 PK;
        u32x4 LAB;
        {
            const unsigned un = (unsigned)n_steps;
            asm volatile(VOSPROP_MASK_LOOP
                         : "=" VOSPROP_MASK_REG_S0(S0), "=" VOSPROP_MASK_REG_S1(S1), "=" VOSPROP_MASK_REG_Y(Y),
                           "=" VOSPROP_MASK_REG_PK(PK), "=" VOSPROP_MASK_REG_LAB(LAB), "+" VOSPROP_MASK_REG_AUX(AUX),
                           "+" VOSPROP_MASK_REG_CB(CB), "+" VOSPROP_MASK_REG_CTL(CTL)
                         : VOSPROP_MASK_REG_B0(Bq0), VOSPROP_MASK_REG_B1(Bq1), VOSPROP_MASK_REG_B2(Bq2), VOSPROP_MASK_REG_B3(Bq3),
                           [n] "s"(un), [fb_lo] "s"(fb_lo), [fb_hi] "s"(fb_hi), [tb_lo] "s"(tb_lo), [tb_hi] "s"(tb_hi),
                           [ldsa] "s"(ldsa), [lds3] "s"(lds3), [role] "s"(role), [tab] "s"(tab_base), [sp0] "s"(sp0)
                         : VOSPROP_MASK_CLOBBERS);
        }
        // 5: the first segment's tile loop is done.  Taken into scalar registers without a C++ branch and stored with stamp 6: a branch right
        // behind the statement (80 hard-bound output registers live) makes hipcc park 16 of them in scratch - 64 B per lane and
        // segment, +12 MB of HBM writes per 480p launch (seen in the WRITE_SIZE counter, not in the time)
        unsigned long long t_loop_done;
        asm volatile("s_cmp_eq_u64 %1, 0\n\ts_cbranch_scc1 LNOSTAMP%=\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)\nLNOSTAMP%=:"
                     : "=s"(t_loop_done) : "s"(A.dbg) : "scc", "memory");
        const bool centred = n_steps >= 2;      // the first tile went through the rescale path (boundary 2)

        // ---- the segment's last two tiles have no chain to hide under ----
        f32x4 Yc[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) Yc[cb] = f32x4{Y[4 * cb], Y[4 * cb + 1], Y[4 * cb + 2], Y[4 * cb + 3]};
        const bf16x8 labp = __builtin_bit_cast(bf16x8, LAB);
        const bool odd = n_steps & 1;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {      // tile n-2: pk and its label fragment are in registers (a one-step segment finds zeros)
            const int o = 4 * cb;
            const u32x4 p = odd ? u32x4{PK[8 + o], PK[9 + o], PK[10 + o], PK[11 + o]} : u32x4{PK[o], PK[o + 1], PK[o + 2], PK[o + 3]};
            Yc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(labp, __builtin_bit_cast(bf16x8, p), Yc[cb], 0, 0, 0);
        }
        float Mc[2];
        {   // tile n-1: weights from its scores, labels from its ring slot (a staging-only wave: zeros in, finite values out, not stored)
            const int slot = (n_steps - 1) % kMaskRing;
            const bf16x8 lab1 = *(const bf16x8*)(smem + slot * kMaskSlot + kMaskOffLab + lane_l * 16);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                float sv[8];      // rows 4 kb + i, then 16 + 4 kb + i, of column 16 cb + j16
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) sv[4 * rb + i] = odd ? S0[8 * rb + 4 * cb + i] : S1[8 * rb + 4 * cb + i];
                float xm = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
                xm = fmaxf(xm, __shfl_xor(xm, 16));
                xm = fmaxf(xm, __shfl_xor(xm, 32));
                const bool forced = !centred;
                const bool mine = forced || xm > kMaskAlarm;
                const float shift = mine ? xm : 0.0f;
                const float sc = forced ? 1.0f : __builtin_amdgcn_exp2f(-shift);
#pragma unroll
                for (int i = 0; i < 4; ++i) Yc[cb][i] *= sc;
                Mc[cb] = AUX[VOSPROP_MASK_AUX_MC + cb] + shift;
                u32x4 p;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    p[i] = cvt_pk_bf16(__builtin_amdgcn_exp2f(sv[2 * i] - shift), __builtin_amdgcn_exp2f(sv[2 * i + 1] - shift));
                Yc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lab1, __builtin_bit_cast(bf16x8, p), Yc[cb], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // drain the look-ahead pieces before the ring is re-staged
        __syncthreads();

        // ---- this segment's partial: rows (m, l = 0, numerators[d]) x 256 columns (combine_kernel, no_l form) ----
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            float* part = A.part + ((size_t)part_slot * A.part_rows) * kBT + wave * kColsPerWave + 16 * cb + j16;
            if (kb == 0) {
                part[0] = pad_wave ? 0.0f : Mc[cb] / c;
                part[kBT] = 0.0f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int cls = 4 * kb + i;
                if (cls < A.d) part[(size_t)(2 + cls) * kBT] = pad_wave ? 0.0f : Yc[cb][i];
            }
        }
        if (A.dbg && tid == 0 && si == seg0) A.dbg[(size_t)blockIdx.x * 8 + 5] = t_loop_done;
        VOSPROP_MASK_STAMP(6);      // 6: the (last) segment's partial is stored
    }
    {
        const int si = seg0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        VOSPROP_MASK_STAMP(7);      // 7: kernel exit
    }
}

}  // namespace vosprop
