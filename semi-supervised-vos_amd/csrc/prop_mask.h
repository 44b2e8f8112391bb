// The mask-only dense propagation kernel (label mode, prediction not requested: the step of the frame loop, bench.py's timed
// kernel).  Same arithmetic as prop_dense_kernel<false, false, 0, false> - reference src/model/predict.py:46-70 followed by the
// arg-max of src/utils/inference_utils.py:70 - same ring, segment table and partial format; the TILE LOOP is a hand-ordered
// instruction stream (tools/gen_mask_loop.py -> prop_mask_loop.inc), entered from this file as ONE asm statement with a fixed
// register map.  What changed against the hipcc-scheduled loop, and why (profiles/r03_dense_kernel_ablations.txt priced the old
// step: 57 us of a 184 us launch were a synchronised skeleton that computed nothing, 71 us of score MFMAs stood exposed):
//
//   * The weighted exponent comes out of the matrix core.  The first score MFMA of a tile takes LM as its C operand, where
//       LM[r, t] = log2 w[r, t] - M_t           (w = the spatial prior, M_t = the column's running reference level, log2 units)
//     is itself the output of one 16-deep MFMA (reference-side coordinates x target-side constants, as in prop_bf16.h, with three
//     more K channels carrying the 3-way bf16 split of -(g Q_t c + M_t) and one that sends the padded rows of a frame's last tile
//     to -1e30), rebuilt only when the pixel tile or the sigma class changes.  The target fragments are pre-multiplied by
//     c = temperature * log2(e), so the accumulator IS  x = s c + log2 w - M  and the weighted probability is one v_exp_f32 of it:
//     no fma, no prior block, no tail-tile masking, no accumulator zeroing in the loop.
//   * No cursor arithmetic.  A per-step control table (one entry per lane, rebuilt by vector code every 64 steps outside the
//     loop from a table the prologue builds in LDS) holds the LDS-DMA source offsets of tile q+3 and the flags of tile q+1; a
//     step fetches it with two v_readlane.
//   * Every wave stages (2 or 3 pieces per step, scalar address arithmetic only), one barrier per step, nothing behind the chain:
//     the label MFMAs of tile q-2 sit in gaps 1 and 2 of chain q, the overflow alarm of tile q-2 was decided in step q-1 and
//     is ONE branch at the step boundary.
//   * One statement per segment.  The loop leaves its steady state only at step boundaries (alarm, control table block exhausted,
//     segment end); the rescale path and the table refill are straight-line code of the same statement (hipcc could not keep 216
//     hard-bound registers in place across a C++ loop: it parked them in scratch around every re-entry).  The segment's prologue
//     and its last two tiles are C++ below.
//
// Alarm / reference level: a = 2^x stays finite up to x = 127 and is packed to bf16 (same exponent range), Y accumulates in f32;
// the loop leaves when a weighted exponent exceeds kMaskAlarm = 100 (57 780 terms of 2^100 still fit f32), the column's level M is
// raised by its maximum, Y is rescaled ONCE and the pending tile redone (cdna guide T13: everything still at the old level is
// scaled exactly once - Y, the pending tile's weights, the next tile's accumulator which already holds the old LM, and LM).  The
// first tile of a segment always takes that path (it sets M from data, either sign), so afterwards max x >= 0 by construction:
// nothing that matters can underflow.  test_gpu_parity.py forces the path with peaky features.
#pragma once
#include "common.h"
#include "prop_bf16.h"
#ifndef VOSPROP_MASK_LOOP_INC
#define VOSPROP_MASK_LOOP_INC "prop_mask_loop.inc"      // (tools/mask_variants.sh builds timing variants from other generator outputs)
#endif
#include VOSPROP_MASK_LOOP_INC

namespace vosprop {

constexpr int kMaskSlot = VOSPROP_MASK_SLOT;
constexpr int kMaskRing = VOSPROP_MASK_NSLOT;
constexpr int kMaskOffCoord = VOSPROP_MASK_OFF_COORD;
constexpr int kMaskOffLab = VOSPROP_MASK_OFF_LAB;
constexpr float kMaskAlarm = VOSPROP_MASK_ALARM;
constexpr int kMaskTabBlock = VOSPROP_MASK_TAB_BLOCK;      // control-table entries a wave holds at a time (one per lane)
constexpr int kMaskTabEntry = VOSPROP_MASK_TAB_ENTRY;      // bytes per entry in LDS
constexpr int kMaskTabCap = 2048;                          // entries the LDS table holds: a segment may walk kMaskMaxSteps tiles
constexpr int kMaskAhead = VOSPROP_MASK_AHEAD;             // tiles the LDS-DMA staging runs ahead of the scoring
constexpr int kMaskMaxSteps = kMaskTabCap - kMaskTabBlock - kMaskAhead;      // (engine.hip falls back to prop_dense_kernel beyond that)
static_assert(kMaskOffCoord == kLdsFeat && kMaskOffLab == kLdsFeat + kLdsCoord && kMaskSlot == kMaskOffLab + kLdsLab,
              "prop_mask_loop.inc and prop_bf16.h disagree on the slot layout");

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

__global__ __launch_bounds__(kWaves * 64, 2) void prop_mask_kernel(const PropArgs A) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[kMaskRing * kMaskSlot];
    __shared__ unsigned s_off[2 * kMaxRef];      // per sampled frame: byte offset of its slot in the feature ring / label ring
    // control table of the segment being walked, one entry per stream position: TA (feature-ring offset of the tile | flags of the
    // tile kMaskAhead-1 positions earlier), coordinate-table offset, label-ring offset, feature-ring offset
    __shared__ __attribute__((aligned(16))) unsigned s_tab[kMaskTabCap * (kMaskTabEntry / 4)];

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
        s_off[kMaxRef + tid] = sl * (unsigned)(TPF * kLdsLab);
    }

    // ---- staging roles (prop_dense.h has the piece list): every wave stages feature pieces w and w + 8 of the padded 528-B row
    // image; waves 0-3 (role A) a third piece each - feature piece 16, the coordinates, the two label halves
    auto feat_src_off = [&](int piece) -> unsigned {
        int qq = 64 * piece + lane;
        if (qq >= kTileR * 33) qq = 0;      // lanes past the image (piece 16, lanes 32-63): any valid source, lands in slack
        int row = qq / 33, ch = qq - row * 33;
        if (ch == 32) ch = 31;
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
        third_base = (const unsigned char*)A.lab_hi + (wave - 2) * 1024;
        lds3_off = kMaskOffLab + (wave - 2) * 1024;
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
    const unsigned tab_col = third_col == 0 ? 12u : third_col == 1 ? 4u : 8u;      // this wave's TB column inside an entry (12: the feature offset without the flags)

    const int seg0 = A.seg_off[blockIdx.x], seg1 = A.seg_off[blockIdx.x + 1];
    __syncthreads();      // s_off
    for (int si = seg0; si < seg1; ++si) {
        const Segment sg = A.segs[si];
        const int tt = __builtin_amdgcn_readfirstlane(sg.tt);
        const int r_lo = __builtin_amdgcn_readfirstlane(sg.r_lo);
        const int n_steps = __builtin_amdgcn_readfirstlane(sg.n_steps);
        const int part_slot = __builtin_amdgcn_readfirstlane(sg.slot);

        // (per-segment values are derived from an opaque copy of the thread id: hoisted above the segment loop they would be
        // live - or spilled - across the 216-register statement below)
        int tid_l = tid;
        asm volatile("" : "+v"(tid_l));
        const int lane_l = tid_l & 63, j_l = tid_l & 31, h_l = (tid_l >> 5) & 1;

        // ---- target (B operand) fragments: 32 columns x 256 channels per wave, resident; loads fly under the rest of the prologue
        const int t = tt * kBT + wave * kColsPerWave + j_l;
        const int t_ld = t < A.target_rows ? t : A.target_rows - 1;
        const bf16_t* trow = A.target_feat + (size_t)t_ld * kC + h_l * 8;
        u32x4 Braw[16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) Braw[ks] = *(const u32x4*)(trow + ks * 16);
        // target-side constants of the prior MFMA for both sigma classes (engine.hip build_target_consts) and g Q_t c
        const int tq = t < A.HW ? t : A.HW - 1;
        const u32x4 cb1 = ((const u32x4*)A.tc_b)[((size_t)tq) * 2 + h_l];
        const u32x4 cb2 = ((const u32x4*)A.tc_b)[((size_t)A.HWp + tq) * 2 + h_l];
        const float kq1 = A.tc_kq[tq], kq2 = A.tc_kq[A.HWp + tq];

        // ---- control table of the segment -> LDS: entry of stream position p (clamped to the segment's last step):
        //   TA = byte offset of the tile in the feature ring | flags of the tile kMaskAhead-1 positions earlier (bit 0: it opens a pixel tile
        //        or a sigma class - its LM must be built -, bit 1: its sigma class);  coordinate / label offsets of the tile
        for (int p0 = tid_l; p0 < n_steps + kMaskTabBlock + kMaskAhead; p0 += kWaves * 64) {
            const int p = p0 < n_steps - 1 ? p0 : n_steps - 1;
            const int r = r_lo + p;
            const int tile = r / N, n = r - tile * N;
            const unsigned fo = s_off[n] + (unsigned)tile * (unsigned)kGlbFeat;
            int p2 = p0 - (kMaskAhead - 1);      // (step q reads entry q + kMaskAhead and wants the flags of tile q + 1)
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
                u32x4{fo | flags, (unsigned)tile * (unsigned)kLdsCoord, s_off[kMaxRef + n] + (unsigned)tile * (unsigned)kLdsLab, fo};
        }
        __syncthreads();
        const unsigned tab_a = s_tab[lane_l * (kMaskTabEntry / 4)];
        const unsigned tab_b = s_tab[lane_l * (kMaskTabEntry / 4) + tab_col / 4];

        // "tile -1" (the first step's previous tile) has probabilities 0 and takes its labels from the last slot: zero them, or
        // stale LDS bits that happen to spell a NaN would turn 0 x NaN into the accumulators
        if (tid_l < kLdsLab / 16) *(f32x4*)(smem + (kMaskRing - 1) * kMaskSlot + kMaskOffLab + tid_l * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

        // ---- tiles 0 .. kMaskAhead-1 -> slots 0 .. kMaskAhead-1
#pragma unroll
        for (int i = 0; i < kMaskAhead; ++i) {
            const unsigned ea = (unsigned)__builtin_amdgcn_readlane((int)tab_a, i) & ~15u;
            const unsigned eb = (unsigned)__builtin_amdgcn_readlane((int)tab_b, i);
            const unsigned lds = smem_base + (unsigned)i * kMaskSlot;
            glds16s2(src_a, ea, feat_base, lds, (unsigned)wave * 1024);
            glds16s2(src_b, ea, feat_base, lds, ((unsigned)wave + 8) * 1024);
            if (role_a) glds16s2(src_3, eb, third_base, lds, lds3_off);
        }

        // the target fragments are "used" here: one memory latency per segment start
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) asm volatile("" : "+v"(Braw[ks]));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // c folded into the target side: T' = bf16(c T) (f16 features of an f16 encoder are converted on the way, one rounding)
        u32x16 Bq0, Bq1, Bq2, Bq3;
        if (A.target_f16) {
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned w = Braw[ks][i];
                    const float lo = (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xFFFFu));
                    const float hi = (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16));
                    const unsigned pk = cvt_pk_bf16(lo * c, hi * c);
                    if (ks < 4) Bq0[ks * 4 + i] = pk;
                    else if (ks < 8) Bq1[(ks - 4) * 4 + i] = pk;
                    else if (ks < 12) Bq2[(ks - 8) * 4 + i] = pk;
                    else Bq3[(ks - 12) * 4 + i] = pk;
                }
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned w = Braw[ks][i];
                    const unsigned pk = cvt_pk_bf16(__uint_as_float(w << 16) * c, __uint_as_float(w & 0xFFFF0000u) * c);
                    if (ks < 4) Bq0[ks * 4 + i] = pk;
                    else if (ks < 8) Bq1[(ks - 4) * 4 + i] = pk;
                    else if (ks < 12) Bq2[(ks - 8) * 4 + i] = pk;
                    else Bq3[(ks - 12) * 4 + i] = pk;
                }
            }
        }
        __syncthreads();      // tiles 0-2 and the zeroed label area are visible

        // ---- inputs of the loop statement (everything else it needs at its start it sets up itself: zeros, the fragments
        // ks = 0..7 of tile 0 and tile 0's prior tile) ----
        u32x16 CTL;
        u32x8 CB;
        f32x8 AUX;            // g Q_t c of both sigma classes in; the column's reference level M (log2 units) out
#pragma unroll
        for (int r = 0; r < 16; ++r) CTL[r] = 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            CB[i] = cb1[i];
            CB[4 + i] = cb2[i];
        }
        CTL[VOSPROP_MASK_CTL_SRCA] = src_a;
        CTL[VOSPROP_MASK_CTL_SRCB] = src_b;
        CTL[VOSPROP_MASK_CTL_SRC3] = src_3;
        CTL[VOSPROP_MASK_CTL_ROWLO] = smem_base + (unsigned)(j_l * kRowB + h_l * 16);
        CTL[VOSPROP_MASK_CTL_ROWHI] = smem_base + (unsigned)(j_l * kRowB + h_l * 16) + 3u * kMaskSlot;
        CTL[VOSPROP_MASK_CTL_LANELO] = smem_base + (unsigned)lane_l * 16;
        CTL[VOSPROP_MASK_CTL_LANEHI] = smem_base + (unsigned)lane_l * 16 + 3u * kMaskSlot;
        CTL[VOSPROP_MASK_CTL_TA] = tab_a;
        CTL[VOSPROP_MASK_CTL_TB] = tab_b;
        CTL[VOSPROP_MASK_CTL_TABA] = (unsigned)lane_l * kMaskTabEntry;
        CTL[VOSPROP_MASK_CTL_TABB] = (unsigned)lane_l * kMaskTabEntry + tab_col;
#pragma unroll
        for (int r = 0; r < 8; ++r) AUX[r] = 0.0f;
        AUX[VOSPROP_MASK_AUX_KQ1] = kq1;
        AUX[VOSPROP_MASK_AUX_KQ2] = kq2;
        const int n0 = r_lo - (r_lo / N) * N;
        const unsigned sp0 = (unsigned)__builtin_amdgcn_readfirstlane((int)((A.sparse_mask >> n0) & 1ull));      // sigma class of tile 0

        // ---- the tile loop, its control and its rare paths (rescale, table refill): ONE statement per segment ----
        f32x16 S0, S1, Y;     // out: scores of the last two tiles (by parity), numerators
        u32x16 PK;            // out: packed weights, even tile | odd tile
        u32x8 LAB;            // out: label fragments of tile n-2
        {
            const unsigned un = (unsigned)n_steps;
            asm volatile(VOSPROP_MASK_LOOP
                         : "=" VOSPROP_MASK_REG_S0(S0), "=" VOSPROP_MASK_REG_S1(S1), "=" VOSPROP_MASK_REG_Y(Y),
                           "=" VOSPROP_MASK_REG_PK(PK), "=" VOSPROP_MASK_REG_LAB(LAB), "+" VOSPROP_MASK_REG_AUX(AUX),
                           "+" VOSPROP_MASK_REG_CB(CB), "+" VOSPROP_MASK_REG_CTL(CTL)      // (the loop overwrites parts of CTL)
                         : VOSPROP_MASK_REG_B0(Bq0), VOSPROP_MASK_REG_B1(Bq1),
                           VOSPROP_MASK_REG_B2(Bq2), VOSPROP_MASK_REG_B3(Bq3), [n] "s"(un), [fb_lo] "s"(fb_lo), [fb_hi] "s"(fb_hi),
                           [tb_lo] "s"(tb_lo), [tb_hi] "s"(tb_hi), [ldsa] "s"(ldsa), [lds3] "s"(lds3), [role] "s"(role_b),
                           [tab] "s"(tab_base), [sp0] "s"(sp0)
                         : VOSPROP_MASK_CLOBBERS);
        }
        float Mc = AUX[VOSPROP_MASK_AUX_MC];
        const bool centred = n_steps >= 2;      // the first tile went through the rescale path (boundary 2)

        // ---- the segment's last two tiles have no chain to hide under ----
        {   // tile n-2: pk and label fragments are in registers (a one-step segment finds zeros there)
            const bool odd = n_steps & 1;
            const u32x4 p0 = odd ? u32x4{PK[8], PK[9], PK[10], PK[11]} : u32x4{PK[0], PK[1], PK[2], PK[3]};
            const u32x4 p1 = odd ? u32x4{PK[12], PK[13], PK[14], PK[15]} : u32x4{PK[4], PK[5], PK[6], PK[7]};
            const u32x4 l0 = {LAB[0], LAB[1], LAB[2], LAB[3]}, l1 = {LAB[4], LAB[5], LAB[6], LAB[7]};
            Y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, l0), __builtin_bit_cast(bf16x8, p0), Y, 0, 0, 0);
            Y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, l1), __builtin_bit_cast(bf16x8, p1), Y, 0, 0, 0);
        }
        {   // tile n-1: weights from its scores, labels from its ring slot
            float sv[16];
            if (n_steps & 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sv[r] = S0[r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) sv[r] = S1[r];
            }
            const float xm = half_max(max16v(sv));
            const bool forced = !centred;
            const bool mine = forced || xm > kMaskAlarm;
            const float shift = mine ? xm : 0.0f;
            const float sc = forced ? 1.0f : __builtin_amdgcn_exp2f(-shift);
#pragma unroll
            for (int r = 0; r < 16; ++r) Y[r] *= sc;
            Mc += shift;
            u32x4 p0, p1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                p0[i] = cvt_pk_bf16(__builtin_amdgcn_exp2f(sv[2 * i] - shift), __builtin_amdgcn_exp2f(sv[2 * i + 1] - shift));
                p1[i] = cvt_pk_bf16(__builtin_amdgcn_exp2f(sv[8 + 2 * i] - shift), __builtin_amdgcn_exp2f(sv[8 + 2 * i + 1] - shift));
            }
            const int slot = (n_steps - 1) % kMaskRing;
            const unsigned char* lh = smem + slot * kMaskSlot + kMaskOffLab + lane_l * 16;
            const bf16x8 l0 = *(const bf16x8*)lh, l1 = *(const bf16x8*)(lh + 1024);
            Y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(l0, __builtin_bit_cast(bf16x8, p0), Y, 0, 0, 0);
            Y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(l1, __builtin_bit_cast(bf16x8, p1), Y, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // drain the look-ahead pieces before the ring is re-staged
        __syncthreads();

        // ---- this segment's partial: rows (m, l = 0, numerators[d]) x 256 columns (combine_kernel, no_l form) ----
        float* part = A.part + ((size_t)part_slot * A.part_rows) * kBT + wave * kColsPerWave + j_l;
        if (h_l == 0) {
            part[0] = Mc / c;
            part[kBT] = 0.0f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cls = acc_row(r, h_l);
            if (cls < A.d) part[(size_t)(2 + cls) * kBT] = Y[r];
        }
    }
}

}  // namespace vosprop
