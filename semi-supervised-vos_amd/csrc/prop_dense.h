// The dense propagation kernel, software-pipelined INSIDE the wave: every dense form that keeps the softmax denominators (a
// prediction was requested, probability mode), both top-k passes and the materialised-affinity variant.  The plain mask-only step
// of the frame loop - the kernel bench.py times - is prop_mask_kernel (prop_mask.h), a hand-ordered stream of the same arithmetic.
//
// Arithmetic helpers, LDS layout constants and LDS-DMA pieces live in prop_bf16.h.  What the round-2 stamps of the two-burst kernel showed (profiles/r02_*): with a wave in its MFMA burst and its SIMD
// partner in its softmax burst, BOTH bursts take ~900 cycles - the sum of what they take alone (~600 + ~300): two waves of one SIMD
// do not overlap matrix and vector work, while up to five single-issue VALU instructions DO hide in the 32-cycle shadow of an MFMA
// of the SAME wave (tools/ubench_issue.hip; MI355X_MICROARCH "issue cost" rows).  So here every wave runs ONE continuous stream:
//
//     step p:   S(p)  = 16 score MFMAs of tile p, and in the gap after MFMA ks, row ks of the softmax of tile p-1:
//                       e = S(p-1)[ks] c - m c ;  q = 2^e ;  l += q ;  a = q w[ks] ;  (odd ks) pack a pair to bf16
//                       - 4.25 VALU instructions with ONE transcendental per gap (the prior tile w is per pixel tile, prop_bf16.h) -
//                       plus the gap's LDS fragment read (tile p second half / tile p+1 first half) and, in three gaps, one LDS-DMA
//                       piece of tile p+3;
//               then the rare rescale check of tile p-1, its two label MFMAs, the prior tile if the pixel tile / sigma changed,
//               `s_waitcnt vmcnt(3)` and ONE barrier.
//
// Tile t lives in LDS ring slot t % 6: written by DMA during step t-3, first fragments read during step t-1, the rest and the
// coordinates during step t, the LABELS during step t+1 (when its label product runs - so they need no registers in between); the
// slot is re-targeted by the DMA of step t+3 (six slots, not five: the two waves of a SIMD keep their one barrier per step half a
// step apart, see the main loop).  The loop is unrolled by two so that the score accumulators of "this" and "the previous" tile
// swap roles without copies.
#pragma once
#include "common.h"
#include "prop_bf16.h"


#define VOSPROP_UNLIKELY(x) __builtin_expect(!!(x), 0)

namespace vosprop {

constexpr int kRing5 = 6;   // LDS ring slots of the dense kernel (135 168 B + 20 KiB of per-lane constants = 155 648 B <= 160 KiB)
constexpr int kRingLast = (kRing5 - 1) * kLdsBuf;
constexpr float kAlarmExp = 8.0f;   // log2(kSumThrV3)

// rows 0..15 of one tile's softmax against the running max (mc = m c), sequential form (rescale path and the segment's last tile).
// FUSED (the no-denominator form): Wt holds log2 w - m c and the weighted probability is ONE exponential, a = 2^(S c + Wt).
template <bool PROB, bool FUSED = false>
__device__ __forceinline__ void softmax_rows(const f32x16& Sp, const float (&Wt)[16], float c, float mc, float& lt0, float& lt1,
                                             bf16x8& pk0, bf16x8& pk1) {
    lt0 = 0.0f;
    lt1 = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const float qa = __builtin_amdgcn_exp2f(__builtin_fmaf(Sp[r], c, FUSED ? Wt[r] : -mc));
        const float qb = __builtin_amdgcn_exp2f(__builtin_fmaf(Sp[r + 1], c, FUSED ? Wt[r + 1] : -mc));
        bf16_t ha, hb;
        if (FUSED) {
            ha = (bf16_t)qa;
            hb = (bf16_t)qb;
        } else if (PROB) {       // the denominator sees the numbers the label product sees (columns sum to 1, see prop_bf16.h)
            ha = (bf16_t)qa;
            hb = (bf16_t)qb;
            lt0 += (float)ha;
            lt1 += (float)hb;
        } else {
            lt0 += qa;
            lt1 += qb;
            ha = (bf16_t)(qa * Wt[r]);
            hb = (bf16_t)(qb * Wt[r + 1]);
        }
        if (r < 8) { pk0[r] = ha; pk0[r + 1] = hb; }
        else { pk1[r - 8] = ha; pk1[r - 7] = hb; }
    }
}

// MAT: 0 = the fused kernel.  1 / 2 = the two halves of the MATERIALISED-affinity variant (BASELINE.json configs[4], "HBM-bandwidth
// stress"; the shape of the reference's own algorithm, src/model/predict.py:49-55, which writes the (N HW) x HW affinity to memory
// and reads it back): MAT 1 runs only the score MFMAs and stores every 32 x 32 score tile to HBM as bf16 (64 lanes x 32 B,
// accumulator order), MAT 2 runs everything EXCEPT the score MFMAs and takes the score tiles back from HBM (two LDS-DMA pieces per
// wave and step in place of the two feature pieces).  2 x N HW^2 x 2 B of traffic per step: 7.46 GB at 720p.
// NEED_L = false (label propagation with only the class map / mask wanted): the softmax denominators are not accumulated.  They
// scale every class of a column alike, so the arg-max - all that vosprop_step's mask and new label depend on - does not see them;
// the per-tile sums that also served as the overflow alarm of the optimistic softmax become a running max (v_max3: 8 instructions
// per tile instead of 16 adds).  The partial's l row is written as 0 and combine_kernel does not divide (engine.hip propagate).
// [r3] With no denominator there is no use for the un-weighted probability either: the prior tile is kept as LM = log2 w - m c and
// the weighted probability is ONE exponential, a = 2^(S c + LM) - fma + exp per score instead of fma + exp + mul, and no
// exponentials at all in a prior tile (16 fma).  Same value up to rounding: 2^(x) 2^(y) vs 2^(x + y), |x + y| < 2^8 => relative
// difference <= 2^-16, far inside the bf16 packing that follows; underflow happens where the product underflowed (log2 w <= 0).
// When the running max moves (rare) LM is rebuilt from the tile's coordinates (they are still in its ring slot).
//
// [r3] TK = 1 / 2: the two passes of the TOP-K variant (not in the reference; definition in aux_kernels.h)
// on this kernel's pipeline - same staging, ring, barriers and MFMA chain; only what happens in the gaps of the chain differs, and
// neither pass has an exponential, a label product or a running max:
//   TK = 1  every (reference tile, half) GROUP of 16 scores a lane owns gives one number, the maximum of its weighted exponents
//           E = S c + log2 w; the lane keeps the KS largest of them, sorted, each with its stream index r and half h packed into
//           the low mantissa bits (tk_idx_bits; one v_and_or).  16 fma + 8 max3 in gaps 0-7, the KS v_med3 of the insertion in gaps
//           8-15: 24 + KS vector instructions per tile against the dense form's 48.  topk_select2_kernel merges the lists: the k-th
//           largest group maximum G_k bounds the k-th largest element from below, and every element >= G_k lies in one of the
//           (at most k, ties aside) groups whose maximum is >= G_k - whose tiles it marks in a bitmap per target tile.
//   TK = 2  re-scores ONLY the marked tiles (workgroup = one of tk_chunks equal shares of a target tile's marked tiles, listed in
//           LDS by the prologue): the same MFMAs on the same bytes give the same E bit for bit, so a lane recognises its
//           candidate groups by the same packed number (>= tk_thr[t]) and writes the group's 16 exponents + r to a slot of its
//           own: no atomics, no counters shared between lanes.  topk_combine2_kernel takes the k largest exactly.
template <bool PROB, bool LAB_LO, int MAT = 0, bool NEED_L = true, int TK = 0, int KS = 8>
__global__ __launch_bounds__(kWaves * 64, 2) void prop_dense_kernel(const PropArgs A) {
    static_assert(NEED_L || (!PROB && !LAB_LO && MAT == 0), "denominators may only be dropped in plain label mode");
    static_assert(TK == 0 || (!PROB && !LAB_LO && MAT == 0 && !NEED_L), "the top-k passes are label-mode, mask-only forms");
    static_assert(NEED_L || TK != 0, "the plain mask-only step is prop_mask_kernel (prop_mask.h); NEED_L = false remains for the top-k passes");
    __shared__ __attribute__((aligned(16))) unsigned char smem[kRing5 * kLdsBuf];
    __shared__ __attribute__((aligned(16))) bf16x8 s_bx[2][kWaves * 64];   // per-lane prior constants (see prop_bf16.h)
    __shared__ float s_kq[2][kWaves * 64];
    // TK 2: this workgroup's share of its target tile's marked reference tiles, (frame << 16 | pixel tile) per entry, in walk order
    __shared__ unsigned s_list[TK == 2 ? kTkListCap : 1];
    __shared__ int s_scan[kWaves];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;       // MFMA row (A operand) / column (B operand, C/D)
    const int h = lane >> 5;       // k-half of the operand fragments / row-half of the accumulator
    const int TPF = A.tiles_per_frame;
    const int N = A.n_ref;
    const float c = A.c;
    const bool ragged = A.HW != A.HWp;
    const int rows_last = A.HW - (TPF - 1) * kTileR;   // valid rows of a frame's last tile

    // ---- staging roles: every wave issues exactly THREE LDS-DMA pieces per tile (prop_bf16.h: 17 feature pieces in the padded
    // 528-B row image, 1 coordinate piece, 2 + 2 label pieces).  Pieces 1 and 2 are feature pieces w and w + 8; the third piece
    // is chosen ONCE per wave as (base pointer, per-slot stride, per-tile stride, LDS offset) so that the tile loop has no role
    // branches: wave 0 feature piece 16, wave 1 coordinates, waves 2-3 label hi, waves 4-5 label lo, the rest repeat piece 1.
    auto feat_src_off = [&](int piece) -> unsigned {
        int qq = 64 * piece + lane;
        if (qq >= kTileR * 33) qq = 0;   // lanes past the image (piece 16, lanes 32-63): any valid source, lands in slack
        int row = qq / 33, ch = qq - row * 33;
        if (ch == 32) ch = 31;
        return (unsigned)(row * 512 + ch * 16);
    };
    // [r2] kStageA (the variants without a low label part): only the OLDER wave of every SIMD (waves 0-3) stages - five pieces each:
    // feature pieces w, w+4, w+8, w+12 and the fifth by the table above.  The older wave wins every issue arbitration and then waits
    // at the barrier for its younger partner (stamps: ~550 cycles per step); the staging instructions are work that can move from the
    // wave that sets the step time to the wave that has slack.
    constexpr bool kStageA = MAT == 0 && !LAB_LO;
    const unsigned src_a = feat_src_off(wave), src_b = feat_src_off(kStageA ? wave + 4 : wave + 8);
    const unsigned src_c = feat_src_off(wave + 8), src_d = feat_src_off(wave + 12);   // kStageA only
    const size_t feat_slot_stride = (size_t)A.HWp * (kC * 2);
    const unsigned char* third_base = (const unsigned char*)A.feat_ring;
    size_t third_slot_stride = feat_slot_stride;
    unsigned third_tile_stride = kGlbFeat, third_lane = src_a, third_lds = (unsigned)wave * 1024;
    if (wave == 0) {
        third_lane = feat_src_off(16);
        third_lds = 16 * 1024;
    } else if (wave == 1 && !PROB) {
        third_base = (const unsigned char*)A.coord_tab;
        third_slot_stride = 0;
        third_tile_stride = kLdsCoord;
        third_lane = lane * 16;
        third_lds = kOffCoord;
    } else if ((wave == 2 || wave == 3) && TK == 0) {      // (the top-k passes read no labels: these waves repeat piece 1)
        third_base = (const unsigned char*)A.lab_hi + (wave - 2) * 1024;
        third_slot_stride = (size_t)TPF * kLdsLab;
        third_tile_stride = kLdsLab;
        third_lane = lane * 16;
        third_lds = kOffLabHi + (wave - 2) * 1024;
    } else if ((wave == 4 || wave == 5) && LAB_LO) {
        third_base = (const unsigned char*)A.lab_lo + (wave - 4) * 1024;
        third_slot_stride = (size_t)TPF * kLdsLab;
        third_tile_stride = kLdsLab;
        third_lane = lane * 16;
        third_lds = kOffLabLo + (wave - 4) * 1024;
    }
    // MAT 2: the first 16 KiB of a slot hold the eight waves' score tiles, not features: the waves without a third piece of their
    // own (and wave 0's feature piece 16) aim their third piece at the KiB of slack behind them
    if (MAT == 2 && third_base == (const unsigned char*)A.feat_ring) third_lds = 16 * 1024;
    third_tile_stride = __builtin_amdgcn_readfirstlane(third_tile_stride);
    third_lds = __builtin_amdgcn_readfirstlane(third_lds);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const unsigned smem_base = (unsigned)(size_t)(lds_ptr)smem;   // LDS byte address of the ring
    // Per-frame source OFFSETS, one sampled frame per LANE (kMaxRef = 64): lane n holds the 32-bit byte offset of frame n inside the
    // feature ring and inside this wave's third-piece array (both rings are far below 4 GiB).  The bases stay in two SGPR pairs for
    // the whole kernel; a step fetches two offsets with v_readlane and adds the tile offset - no 64-bit arithmetic in the tile loop
    // (the loop is issue-bound: 83 % of the SIMD's cycles issue an instruction, a quarter of them scalar).
    const unsigned my_slot = (unsigned)A.slot[lane];
    const unsigned fo = my_slot * (unsigned)feat_slot_stride;
    const unsigned to = my_slot * (unsigned)third_slot_stride;
    const unsigned char* const feat_base = (const unsigned char*)A.feat_ring;

    // TK 2 has no segment table: workgroup b is share (b % tk_chunks) of target tile (b / tk_chunks)
    const int seg0 = TK == 2 ? 0 : A.seg_off[blockIdx.x], seg1 = TK == 2 ? 1 : A.seg_off[blockIdx.x + 1];
    Segment sg_first = Segment{0, 0, 0, 0};      // (a copy of segs[seg0]: it loads beside the range, not behind it)
    if (TK != 2) sg_first = A.seg_first[blockIdx.x];
    if (TK == 1) {      // pass 1 also clears the bitmaps topk_select2_kernel will mark (it runs after this kernel on the stream)
        for (int i = blockIdx.x * (kWaves * 64) + tid; i < A.tk_bitmap_words; i += gridDim.x * (kWaves * 64)) A.tk_bitmap[i] = 0u;
    }
    for (int si = seg0; si < seg1; ++si) {
        Segment sg;
        if (TK == 2) {
            // ---- the walk list of this workgroup: set bits [lo, hi) of its target tile's bitmap, in stream order ----
            sg.tt = (int)blockIdx.x / A.tk_chunks;
            sg.slot = (int)blockIdx.x % A.tk_chunks;      // (the share; there is no partial slot in this pass)
            const unsigned* bm = A.tk_bitmap + (size_t)sg.tt * A.tk_words;
            const int wpt = (A.tk_words + kWaves * 64 - 1) / (kWaves * 64);      // words per thread, a contiguous run each
            const int w0 = tid * wpt, w1 = w0 + wpt < A.tk_words ? w0 + wpt : A.tk_words;
            int cnt_t = 0;
            for (int wd = w0; wd < w1; ++wd) cnt_t += __builtin_popcount(bm[wd]);
            int incl = cnt_t;      // inclusive scan over the wave, then over the eight wave totals
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int u = __shfl_up(incl, o);
                if (lane >= o) incl += u;
            }
            if (lane == 63) s_scan[wave] = incl;
            __syncthreads();
            int base = 0, total = 0;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) {
                const int tw = s_scan[w];
                if (w < wave) base += tw;
                total += tw;
            }
            const int lo = (int)((long long)total * sg.slot / A.tk_chunks), hi = (int)((long long)total * (sg.slot + 1) / A.tk_chunks);
            int rank = base + incl - cnt_t;      // set bits in front of this thread's words
            for (int wd = w0; wd < w1; ++wd) {
                unsigned bits = bm[wd];
                while (bits) {
                    const int b = __builtin_ctz(bits);
                    bits &= bits - 1;
                    if (rank >= lo && rank < hi && rank - lo < kTkListCap) {
                        const unsigned r = (unsigned)(wd * 32 + b);
                        const unsigned pt = r / (unsigned)A.n_ref;
                        s_list[rank - lo] = ((r - pt * (unsigned)A.n_ref) << 16) | pt;
                    }
                    ++rank;
                }
            }
            __syncthreads();
            sg.r_lo = 0;
            sg.n_steps = hi - lo < kTkListCap ? hi - lo : kTkListCap;
        } else {
            sg = si == seg0 ? sg_first : A.segs[si];
        }
        const int tt = __builtin_amdgcn_readfirstlane(sg.tt);
        const int r_lo = __builtin_amdgcn_readfirstlane(sg.r_lo);
        const int n_steps = __builtin_amdgcn_readfirstlane(sg.n_steps);
        const int part_slot = __builtin_amdgcn_readfirstlane(sg.slot);

        // What a segment start computes from the thread id is computed HERE, from an opaque copy: as loop invariants hoisted above
        // the segment loop these values were spilled around the tile loop (256 registers) and every reload brought an
        // `s_waitcnt vmcnt(0)` into the prologue, in front of the staging
        int tid_l = tid, wd_l = A.Wd;
        asm volatile("" : "+v"(tid_l), "+s"(wd_l));
        const int j_l = tid_l & 31, h_l = (tid_l >> 5) & 1;
        // target (B operand) fragments: 32 columns x 256 channels per wave, resident in registers
        const int t = tt * kBT + wave * kColsPerWave + j_l;
        const int t_ld = t < A.target_rows ? t : A.target_rows - 1;
        const bf16_t* trow = A.target_feat + (size_t)t_ld * kC + h_l * 8;
        if (TK == 2 && n_steps <= 0) {      // nothing marked in this share (wave-uniform)
            A.tk_cnt[((size_t)t * 2 + h_l) * A.tk_chunks + part_slot] = 0u;
            continue;
        }
        bf16x8 Bt[16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) Bt[ks] = *(const bf16x8*)(trow + ks * 16);

        // target-side spatial channels for both sigmas and the per-column constants g Q_t c -> LDS (own lane writes, own lane reads)
        if (!PROB) {
            const int tq = t < A.HW ? t : A.HW - 1;
            const int trow_i = tq / wd_l;
            const double at = (double)trow_i, bt = (double)(tq - trow_i * wd_l);
            const double tw = A.two_over_w, gm = A.gamma;
            const double qt = at * at + tw * at * bt + gm * bt * bt;
#pragma unroll
            for (int sgm = 0; sgm < 2; ++sgm) {
                int sg_o = sgm;
                asm volatile("" : "+s"(sg_o));      // (the splits of -g are per-segment work too, not spilled loop invariants)
                const double g = sg_o ? A.g2 : A.g1;
                float ah, am, al, bh, bm, bl, kh, km, kl;
                split3((float)(g * (2.0 * at + tw * bt)), ah, am, al);
                split3((float)(g * (2.0 * gm * bt + tw * at)), bh, bm, bl);
                split3((float)(-g), kh, km, kl);
                bf16x8 B;   // pairs with the reference-side table of engine.hip build_coord_table (prop_bf16.h has the derivation)
                B[0] = (bf16_t)(h_l ? kl : ah);
                B[1] = (bf16_t)(h_l ? kh : am);
                B[2] = (bf16_t)(h_l ? km : al);
                B[3] = (bf16_t)(h_l ? kh : bh);
                B[4] = (bf16_t)(h_l ? 0.0f : bm);
                B[5] = (bf16_t)(h_l ? 0.0f : bl);
                B[6] = (bf16_t)(h_l ? 0.0f : kh);
                B[7] = (bf16_t)(h_l ? 0.0f : km);
                s_bx[sgm][tid_l] = B;
                s_kq[sgm][tid_l] = (float)(g * qt * (double)c);
            }
        }

        ColState st;
        st.m = kNegBig;
        st.l = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) st.Y[r] = 0.0f;
        float Wt[16];      // prior tile of the tile being finished: w (NEED_L) or LM = log2 w - m c (no-denominator form)
#pragma unroll
        for (int r = 0; r < 16; ++r) Wt[r] = 0.0f;
        constexpr bool FUSED = !NEED_L;
        constexpr bool kRowsEarly = FUSED && TK == 0;
        bool w_sparse = false;      // sigma class Wt was built with (FUSED: the rescale path rebuilds LM against the new max)

        // ---- staging cursor (frame inner) and the three pieces of a tile ----
        int sn = 0, stile = 0;
        int sidx = 0;      // TK 2: position of the staging cursor in the walk list
        auto list_at = [&](int i) __attribute__((always_inline)) -> unsigned {      // entry i of the walk list, clamped to its end
            const int ii = i < n_steps ? i : n_steps - 1;
            return (unsigned)__builtin_amdgcn_readfirstlane((int)s_list[TK == 2 ? ii : 0]);
        };
        auto stage_seek = [&](int step) {
            if (TK == 2) {
                sidx = step;
                const unsigned e = list_at(sidx);
                stile = (int)(e & 0xFFFFu);
                sn = (int)(e >> 16);
                return;
            }
            stile = (r_lo + step) / N;
            sn = (r_lo + step) - stile * N;
        };
        // wave-uniform byte offsets of the staging cursor's tile (set by stage_bases, used by the three pieces of a step)
        unsigned so_feat = 0, so_third = 0;
        auto stage_bases = [&]() __attribute__((always_inline)) {
            so_feat = (unsigned)__builtin_amdgcn_readlane((int)fo, sn) + (unsigned)stile * (unsigned)kGlbFeat;
            so_third = (unsigned)__builtin_amdgcn_readlane((int)to, sn) + (unsigned)stile * third_tile_stride;
        };
        const unsigned lds_a = MAT == 2 ? (unsigned)wave * 2048 : (unsigned)wave * 1024;
        const unsigned lds_b = MAT == 2 ? (unsigned)wave * 2048 + 1024 : ((unsigned)wave + 8) * 1024;
        // MAT 1 / 2: this wave's score tiles in HBM: smat[stream index][column block][lane][16 bf16]
        const size_t mat_cb = (size_t)tt * kWaves + wave;                        // column block of 32 target pixels
        const size_t mat_blocks = (size_t)((A.HWp + kBT - 1) / kBT) * kWaves;    // column blocks per stream index
        auto stage_piece = [&](unsigned lds, int i) __attribute__((always_inline)) {   // lds: LDS byte address of the target slot
            if (MAT == 2 && i < 2) {
                const int rs = stile * N + sn;      // stream index of the staged tile
                const unsigned char* src = (const unsigned char*)A.smat + (((size_t)rs * mat_blocks + mat_cb) * 64 + lane) * 32 + i * 16;
                glds16(src, lds + (i ? lds_b : lds_a));
            } else if (kStageA) {
                if (i == 0) glds16s2(src_a, so_feat, feat_base, lds, (unsigned)wave * 1024);
                else if (i == 1) glds16s2(src_b, so_feat, feat_base, lds, ((unsigned)wave + 4) * 1024);
                else if (i == 2) glds16s2(src_c, so_feat, feat_base, lds, ((unsigned)wave + 8) * 1024);
                else if (i == 3) glds16s2(src_d, so_feat, feat_base, lds, ((unsigned)wave + 12) * 1024);
                else glds16s2(third_lane, so_third, third_base, lds, third_lds);
            } else if (i == 0) glds16s2(src_a, so_feat, feat_base, lds, lds_a);
            else if (i == 1) glds16s2(src_b, so_feat, feat_base, lds, lds_b);
            else glds16s2(third_lane, so_third, third_base, lds, third_lds);
        };
        auto stage_advance = [&]() __attribute__((always_inline)) {   // next tile of the stream; stays on the last one at its end
            if (TK == 2) {
                ++sidx;
                const unsigned e = list_at(sidx);
                stile = (int)(e & 0xFFFFu);
                sn = (int)(e >> 16);
                return;
            }
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
        // "tile -1" (the first step's previous tile) has probabilities 0 and takes its labels from the last slot: zero them, or stale LDS
        // bits that happen to spell a NaN would turn 0 x NaN into the accumulators
        float zf = 0.0f;
        asm volatile("" : "+v"(zf));
        if (tid_l < 2 * kLdsLab / 16) *(f32x4*)(smem + kRingLast + kOffLabHi + tid_l * 16) = f32x4{zf, zf, zf, zf};
        stage_seek(0);
        for (int q = 0; q < 3; ++q) {   // prologue: tiles 0, 1, 2
            stage_bases();
            if (!kStageA || wave < kWaves / 2) {
                stage_piece(smem_base + q * kLdsBuf, 0);
                stage_piece(smem_base + q * kLdsBuf, 1);
                stage_piece(smem_base + q * kLdsBuf, 2);
                if (kStageA) {
                    stage_piece(smem_base + q * kLdsBuf, 3);
                    stage_piece(smem_base + q * kLdsBuf, 4);
                }
            }
            stage_advance();
        }
        // the target fragments are "used" HERE: their loads (issued first) fly under the prior-constant arithmetic and the staging of
        // the first three tiles, one memory latency per segment start instead of two - and hipcc's waits for them stay out of the
        // tile loop
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) asm volatile("" : "+v"(Bt[ks]));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (A.target_f16) {      // an f16 encoder's features, read where it left them: the same round-to-nearest-even conversion the
                                 // push kernel would have done on the way into the ring (wave-uniform; once per segment)
            typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const f16x8 hv = __builtin_bit_cast(f16x8, Bt[ks]);
#pragma unroll
                for (int e = 0; e < 8; ++e) Bt[ks][e] = (bf16_t)(float)hv[e];
            }
        }
        __syncthreads();

        // compute cursor (pixel tile, frame) of the tile whose SCORES are being computed
        int ctile = r_lo / N, cn = r_lo - ctile * N;
        int cidx = 0;      // TK 2: position of the compute cursor in the walk list
        if (TK == 2) {
            const unsigned e = list_at(0);
            ctile = (int)(e & 0xFFFFu);
            cn = (int)(e >> 16);
        }
        bool sparse = (A.sparse_mask >> cn) & 1ull;
        bool need_w = !PROB;
        // ---- top-k state ----
        float tkv[TK == 1 ? KS : 1];      // TK 1: the KS largest packed group maxima this lane has seen, descending
        if (TK == 1) {
#pragma unroll
            for (int i = 0; i < KS; ++i) tkv[i] = kTkDummy;
        }
        const unsigned tk_keep = ~((1u << A.tk_idx_bits) - 1u);      // mantissa bits a packed group maximum keeps
        float tk_thr = 3.0e38f;                                      // TK 2: a group whose packed maximum reaches this is dumped
        int tk_cur = 0;                                              // TK 2: groups this lane has dumped
        if (TK == 2 && t < A.HW) tk_thr = A.tk_thr[t];
        float Ek[TK == 2 ? 16 : 1];                                  // TK 2: the weighted exponents of the tile being finished
        int crs_prev = -1;                                           // stream index of the tile being finished
        unsigned tkid_prev = 0xFFFFu;                                // TK 2: the same as (frame << 16 | pixel tile)

        AFrag<PROB> fr;
        if (MAT != 2) fr.prefetch(smem, j, h);
        int crs = TK == 2 ? ctile * N + cn : r_lo;     // stream index of the tile being scored (MAT 1: where its score tile goes)

        f32x16 S0, S1;
#pragma unroll
        for (int r = 0; r < 16; ++r) S1[r] = -__builtin_inff();   // "tile -1": every probability 0 (whatever labels the last slot holds)
        // ring slots of tile p (cur), p+1 (nxt), p-1 (prv), p+3 (stg): counters modulo 6
        // (kept as byte offsets into the ring)
        int s_cur = 0, s_nxt = kLdsBuf, s_prv = kRingLast, s_stg = 3 * kLdsBuf;
        auto ring_advance = [&]() __attribute__((always_inline)) {
            s_prv = s_cur;
            s_cur = s_nxt;
            s_nxt = s_nxt == kRingLast ? 0 : s_nxt + kLdsBuf;
            s_stg = s_stg == kRingLast ? 0 : s_stg + kLdsBuf;
        };

        // finish tile p-1: rescale check (rare), denominators, label MFMAs
        auto finish_prev = [&](const f32x16& Sp, const LabFrag<LAB_LO>& labp, float lt0, float lt1, bf16x8& pk0, bf16x8& pk1) __attribute__((always_inline)) {
            // alarm: a term of this tile above 2^8 (NEED_L: the tile's partial denominator above 2^8)
            if (__any(NEED_L ? lt0 + lt1 > kSumThrV3 : lt0 > st.m + kAlarmExp / c)) {
                // raise the running max (shared by the two half-waves of a column), rescale what was accumulated against the old
                // one exactly once, redo this tile against the new one (cdna guide T13 hazard; Y does not hold this tile yet)
                asm volatile("; rescale" ::: "memory");
                float sv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) sv[r] = Sp[r];
                const float mn = fmaxf(st.m, half_max(max16v(sv)));
                const float sc = __builtin_amdgcn_exp2f((st.m - mn) * c);
                st.l *= sc;
#pragma unroll
                for (int r = 0; r < 16; ++r) st.Y[r] *= sc;
                st.m = mn;
                // LM of this tile against the new max, from the tile's own coordinates (slot s_prv holds tile p-1 until step p+2)
                if (FUSED) prior_tile<true>(smem + s_prv, j, h, s_bx[w_sparse ? 1 : 0][tid], c, s_kq[w_sparse ? 1 : 0][tid] + mn * c, Wt);
                softmax_rows<PROB, FUSED>(Sp, Wt, c, mn * c, lt0, lt1, pk0, pk1);
            }
            if (NEED_L) st.l += lt0 + lt1;
            label_mfmas<LAB_LO>(labp, pk0, pk1, st.Y);
        };

        // TK 2: tile p-1's group of this lane is a candidate group of its column (its packed maximum x reaches the column's
        // threshold - the very number pass 1 computed, bit for bit): the group's 16 exponents and its stream index go to the
        // lane's next slot.  The stores are asm statements so that their NUMBER is known (5, when any lane of the wave stores):
        // the step's closing s_waitcnt counts them.  Returns whether the wave stored.
        static_assert(TK == 0 || kStageA, "the top-k passes assume the older-wave staging scheme (their vmcnt counts)");
        auto tk_dump = [&](float x) __attribute__((always_inline)) -> bool {
            const bool hit = x >= tk_thr && tk_cur < A.tk_cap;
            if (x >= tk_thr && tk_cur >= A.tk_cap) atomicAdd(A.tk_over, 1u);      // a candidate group is DROPPED: reported, never silent
            const bool any = __any(hit);
            if (any) {
                if (hit) {
                    const size_t slot_i = (((size_t)t * 2 + h) * A.tk_chunks + part_slot) * A.tk_cap + tk_cur;
                    const float* dst = A.tk_dump + slot_i * 16;
                    const unsigned* dstr = A.tk_dump_r + slot_i;
                    // the group's tile as (frame << 16 | pixel tile) - what topk_combine2_kernel looks its rows' classes up with; the
                    // "no group" of a share's first step (crs_prev = -1) gets an out-of-range pixel tile
                    const unsigned tk_prev_id = tkid_prev;
                    const f32x4 v0 = {Ek[0], Ek[1], Ek[2], Ek[3]}, v1 = {Ek[4], Ek[5], Ek[6], Ek[7]};
                    const f32x4 v2 = {Ek[8], Ek[9], Ek[10], Ek[11]}, v3 = {Ek[12], Ek[13], Ek[14], Ek[15]};
                    asm volatile("global_store_dwordx4 %0, %1, off\n\tglobal_store_dwordx4 %0, %2, off offset:16\n\t"
                                 "global_store_dwordx4 %0, %3, off offset:32\n\tglobal_store_dwordx4 %0, %4, off offset:48\n\t"
                                 "global_store_dword %5, %6, off"
                                 :
                                 : "v"(dst), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(dstr), "v"(tk_prev_id)
                                 : "memory");
                    ++tk_cur;
                }
            }
            return any;
        };

        // one step: scores of tile p into S, softmax of tile p-1 (scores Sp, labels labp) in the gaps of the chain
        auto step = [&](auto grp, f32x16& S, const f32x16& Sp) __attribute__((always_inline)) {
            constexpr bool MID_BARRIER = (decltype(grp)::value & 1) != 0;   // second wave of every SIMD: its barrier sits after gap 7
            constexpr bool STAGER = (decltype(grp)::value & 2) == 0;        // this wave issues LDS-DMA pieces
            const unsigned char* lb = smem + s_cur;
            const unsigned char* lbn = smem + s_nxt;
            const unsigned b_st = smem_base + (unsigned)s_stg;
            if (STAGER) stage_bases();
            LabFrag<LAB_LO> labp;
            const float mc = st.m * c;
            float lt0 = TK != 0 ? kTkDummy : NEED_L ? 0.0f : kNegBig, lt1 = 0.0f, qprev = 0.0f;
            bf16x8 pk0, pk1;
            const unsigned char* arow = lb + j * kRowB + h * 16;
            const unsigned char* nrow = lbn + j * kRowB + h * 16;
#pragma unroll
            for (int r = 0; r < 16; ++r) S[r] = 0.0f;
            bf16x8 sm0, sm1;
            if (MAT == 2) {      // this wave's score tile of tile p, landed in the slot's first 16 KiB two steps ago
                sm0 = *(const bf16x8*)(lb + wave * 2048 + lane * 16);
                sm1 = *(const bf16x8*)(lb + wave * 2048 + 1024 + lane * 16);
            }
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                if (MAT != 2) {
                    S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.a[ks & 7], Bt[ks], S, 0, 0, 0);
                    // refill the fragment slot just consumed: second half of this tile, then the first half of the next one
                    if (ks < 8) fr.a[ks] = *(const bf16x8*)(arow + (ks + 8) * 32);
                    else fr.a[ks - 8] = *(const bf16x8*)(nrow + (ks - 8) * 32);
                }
                if (STAGER) {
                    if (kStageA) { if (ks % 3 == 1) stage_piece(b_st, ks / 3); }      // gaps 1, 4, 7, 10, 13: pieces 0..4
                    else if (ks == 2 || ks == 7 || ks == 12) stage_piece(b_st, ks / 5);
                }
                if (MAT == 1) continue;                        // score tiles only
                if (TK != 0) {
                    // ---- top-k passes: weighted exponents of tile p-1 and their group maximum (no exponential) ----
                    if (ks < 8) {
                        const float e0 = __builtin_fmaf(Sp[2 * ks], c, Wt[2 * ks]);
                        const float e1 = __builtin_fmaf(Sp[2 * ks + 1], c, Wt[2 * ks + 1]);
                        if (TK == 2) {
                            Ek[2 * ks] = e0;
                            Ek[2 * ks + 1] = e1;
                        }
                        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(lt0) : "v"(lt0), "v"(e0), "v"(e1));      // lt0 = the group's maximum
                    } else {
                        if (ks == 8)      // the maximum with (stream index, half) in its low mantissa bits: ONE v_and_or_b32
                            lt1 = __uint_as_float((__float_as_uint(lt0) & tk_keep) | ((((unsigned)crs_prev << 1) | (unsigned)h) & ~tk_keep));
                        if (TK == 1) {
                            // insertion into the descending list, bottom up, KS / 8 slots per gap:
                            // new v[i] = med3(v[i-1], v[i], x): v[i] if x is below it, x if it lands here, v[i-1] if it lands above
                            constexpr int per = KS / 8;
#pragma unroll
                            for (int q2 = 0; q2 < per; ++q2) {
                                const int i = KS - 1 - ((ks - 8) * per + q2);
                                if (i >= 1) tkv[i] = __builtin_amdgcn_fmed3f(tkv[i - 1], tkv[i], lt1);
                                else tkv[0] = vmaxf(tkv[0], lt1);
                            }
                        }
                    }
                } else {
                if (ks == 10) labp.load(smem + s_prv, lane);   // labels of tile p-1, for the label MFMAs after the chain
                // rows of the previous tile.  [r3] The mask-only form (3.5 vector instructions per row) does them EARLY - row g in gap
                // g up to gap 11, two rows in gaps 12 and 13, none in gaps 14 and 15 - so that the last exponentials and packings are
                // not left standing behind the 16th MFMA, in front of the label MFMAs that wait for them (hipcc otherwise bunches the
                // last rows there: ~60 exposed cycles per step and wave).
                auto soft_row = [&](int r) __attribute__((always_inline)) {
                    const float q = __builtin_amdgcn_exp2f(__builtin_fmaf(Sp[r], c, FUSED ? Wt[r] : -mc));
                    if (PROB) {
                        if (r & 1) {
                            const bf16_t ha = (bf16_t)qprev, hb = (bf16_t)q;
                            lt0 += (float)ha;
                            lt1 += (float)hb;
                            if (r < 8) { pk0[r - 1] = ha; pk0[r] = hb; }
                            else { pk1[r - 9] = ha; pk1[r - 8] = hb; }
                        } else {
                            qprev = q;
                        }
                    } else {
                        if (NEED_L) {
                            if (r & 1) lt1 += q;
                            else lt0 += q;
                        } else if (r & 1) {      // overflow alarm only, from the raw scores (off the exponential's dependency chain):
                            lt0 = __builtin_fmaxf(__builtin_fmaxf(lt0, Sp[r - 1]), Sp[r]);   // lt0 = max of the tile's scores
                        }
                        const float aq = FUSED ? q : q * Wt[r];
                        if (r & 1) {
                            if (r < 8) { pk0[r - 1] = (bf16_t)qprev; pk0[r] = (bf16_t)aq; }
                            else { pk1[r - 9] = (bf16_t)qprev; pk1[r - 8] = (bf16_t)aq; }
                        } else {
                            qprev = aq;
                        }
                    }
                };
                if (kRowsEarly) {
                    if (ks < 12) soft_row(ks);
                    else if (ks == 12) { soft_row(12); soft_row(13); }
                    else if (ks == 13) { soft_row(14); soft_row(15); }
                } else {
                    soft_row(ks);
                }
                }      // TK == 0
                if (MID_BARRIER && ks == 7) {
                    // this wave's pieces of tile p+2 (issued in step p-1) have landed: only the two of this step may be in flight
                    if (STAGER) asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
                    else asm volatile("s_barrier" ::: "memory");      // nothing of this wave's is in flight
                }
                // pin the interleave (cdna guide T19): after each score MFMA its fragment refill and the 4-5 VALU instructions
                // of one softmax row, in the MFMA's shadow - hipcc otherwise sinks the multiplies and packings below the chain
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
                if (kRowsEarly && (ks == 12 || ks == 13)) __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);   // two rows
                else __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);   // <= 5 VALU (one of them the exponential)
            }
            // the packed weights are "used" here, in the chain's basic block: hipcc otherwise sinks the multiplies and packings of the
            // fast path below the rescale branch, out of the MFMA shadow
            if (MAT != 1 && TK == 0) asm volatile("" : "+v"(pk0), "+v"(pk1));
            if (TK != 0) asm volatile("" : "+v"(lt1));
            if (TK == 1) {      // ... and the list: its insertion belongs in the shadow of gaps 8-15, not in a burst behind the chain
#pragma unroll
                for (int i = 0; i < KS; ++i) asm volatile("" : "+v"(tkv[i]));
            }
            if (TK == 2) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(Ek[i]));
            }
            if (STAGER) stage_advance();
            if (MAT != 1 && TK == 0) finish_prev(Sp, labp, lt0, lt1, pk0, pk1);
            bool tk_stored = false;      // TK 2: this wave issued the five dump stores in this step (they count in vmcnt)
            if (TK == 2) tk_stored = tk_dump(lt1);
            if (MAT == 2) {      // the scores of tile p as they came back from HBM (bf16: the materialised affinity's precision)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    S[e] = (float)sm0[e];
                    S[8 + e] = (float)sm1[e];
                }
            }
            if (MAT == 1) {      // the score tile goes to HBM: 64 lanes x 32 B, accumulator order
                bf16x8 o0, o1;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    o0[e] = (bf16_t)S[e];
                    o1[e] = (bf16_t)S[8 + e];
                }
                bf16_t* dst = A.smat + (((size_t)crs * mat_blocks + mat_cb) * 64 + lane) * 16;
                __builtin_nontemporal_store(o0, (bf16x8*)dst);
                __builtin_nontemporal_store(o1, (bf16x8*)(dst + 8));
            }
            crs_prev = crs;
            if (TK == 2) tkid_prev = ((unsigned)cn << 16) | (unsigned)ctile;
            ++crs;
            // tile p: padded rows of a frame's last tile never enter the softmax (wave-uniform, rare)
            if (ragged && ctile == TPF - 1) {
                asm volatile("; tail tile" ::: "memory");
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (acc_row(r, h) >= rows_last) S[r] = kNegBig;
            }
            // the prior tile of tile p (used from the next step on; tile p-1 is finished)
            if (!PROB && need_w) {
                asm volatile("; prior tile" ::: "memory");
                prior_tile<FUSED>(lb, j, h, s_bx[sparse ? 1 : 0][tid], c,
                                  s_kq[sparse ? 1 : 0][tid] + (FUSED && TK == 0 ? st.m * c : 0.0f), Wt);      // (top-k: log2 w itself)
                w_sparse = sparse;
                need_w = false;
            }
            if (TK == 2) {      // the next marked tile: anywhere further down the stream
                ++cidx;
                const unsigned e = list_at(cidx);
                const int nt = (int)(e & 0xFFFFu);
                if (nt != ctile) need_w = true;
                ctile = nt;
                cn = (int)(e >> 16);
                crs = ctile * N + cn;
            } else if (++cn == N) {
                cn = 0;
                ++ctile;
                need_w = !PROB;
            }
            {
                const bool sp = (A.sparse_mask >> cn) & 1ull;
                if (sp != sparse) need_w = !PROB;
                sparse = sp;
            }
            // this wave's pieces of tile p+2 have landed (the 3 of tile p+3 may stay in flight); the barrier then makes every
            // wave's pieces of p+2 visible and retires the slot of tile p-2 for the DMA of step p+1
            if (MAT == 1) {
                // the score-tile stores are counted by hipcc: a __syncthreads() here would drain them every step.  Own LDS-DMA
                // pieces of tile p+2: all but the 3 pieces and 2 stores issued in this step
                asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            } else {
                if (!MID_BARRIER) {
                    // TK 2: the five dump stores of this step were issued after its five pieces: they are the youngest
                    if (STAGER && kStageA && TK == 2 && tk_stored) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                    else if (STAGER && kStageA) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                    else if (STAGER) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                    __syncthreads();
                }
            }
            ring_advance();
        };

        // ---- ONE barrier per step and wave, at a different place for the two waves of a SIMD (MAT == 0):
        // waves 0-3 at the end of their step p, waves 4-7 after gap 7 of THEIR step p - so the second wave of every SIMD runs half
        // a step behind the first, and its serial tail (rescale check, label MFMAs, prior tile, cursor) sits under its partner's
        // chain instead of under its partner's barrier wait (profiles/r02_dense_kernel_stamps.txt).  With t in local steps:
        // tile t is written at t-3+[.1,.8], read at [t-.5, t+1.6]; barrier k is passed at local time k+1 (first group) / k+.5
        // (second).  Visibility: every wave has waited for its pieces of tile k+2 before barrier k; the earliest read of tile t
        // (local t-.5) is after barrier t-2 (first group) or t-1 (second).  Re-use: the slot of tile t is re-targeted for tile t+6
        // at local t+3.1 or later, i.e. after barrier t+2, which every wave passes at local >= t+2.5 > t+1.6 (with FIVE slots the
        // first group's pieces would overtake the second group's label reads - hence six).
        constexpr bool kSkew = MAT == 0;
        const bool grp_b = kSkew && wave >= kWaves / 2;
        // step forms: bit 0 = barrier after gap 7, bit 1 = this wave does not stage
        typedef std::integral_constant<int, 0> GrpA;                       // waves 0-3 (and every wave of the MAT forms)
        typedef std::integral_constant<int, kStageA ? 3 : 1> GrpB;         // waves 4-7, skewed
        int p = 0;
        if (grp_b) {
            for (; p + 1 < n_steps; p += 2) {
                step(GrpB(), S0, S1);
                step(GrpB(), S1, S0);
            }
        } else {
            for (; p + 1 < n_steps; p += 2) {
                step(GrpA(), S0, S1);
                step(GrpA(), S1, S0);
            }
        }
        // the segment's last tile has no chain to hide under (its labels sit in slot s_prv after the last ring_advance)
        auto drain = [&](const f32x16& Sp) __attribute__((always_inline)) {
            if (TK != 0) {      // the last tile's group: exponents, maximum, packed index; list insertion / dump
                float g = kTkDummy;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_fmaf(Sp[r], c, Wt[r]);
                    if (TK == 2) Ek[r] = e;
                    g = vmaxf(g, e);
                }
                const float x = __uint_as_float((__float_as_uint(g) & tk_keep) | ((((unsigned)crs_prev << 1) | (unsigned)h) & ~tk_keep));
                if (TK == 1) {
#pragma unroll
                    for (int i = KS - 1; i >= 1; --i) tkv[i] = __builtin_amdgcn_fmed3f(tkv[i - 1], tkv[i], x);
                    tkv[0] = vmaxf(tkv[0], x);
                } else {
                    (void)tk_dump(x);
                }
                return;
            }
            float lt0, lt1;
            bf16x8 pk0, pk1;
            LabFrag<LAB_LO> labp;
            labp.load(smem + s_prv, lane);
            softmax_rows<PROB, FUSED>(Sp, Wt, c, st.m * c, lt0, lt1, pk0, pk1);
            if (!NEED_L) {      // the alarm of this form looks at the scores
                float sv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) sv[r] = Sp[r];
                lt0 = max16v(sv);
                lt1 = 0.0f;
            }
            finish_prev(Sp, labp, lt0, lt1, pk0, pk1);
        };
        if (p < n_steps) {
            if (grp_b) step(GrpB(), S0, S1);
            else step(GrpA(), S0, S1);
            if (MAT != 1) drain(S0);
        } else {
            if (MAT != 1) drain(S1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the look-ahead pieces before the ring is re-staged
        __syncthreads();

        if (MAT == 1) continue;
        if (TK == 2) {      // how many groups this lane dumped in this share (every lane writes: no memset between steps)
            A.tk_cnt[((size_t)t * 2 + h) * A.tk_chunks + part_slot] = (unsigned)tk_cur;
            continue;
        }
        if (TK == 1) {      // this segment's lists: 256 columns x (half, rank)
            // lists of one target pixel are CONTIGUOUS: [target tile][column][slot rank][half][KS] (the select kernel streams them)
            const int s_first = A.tk_off[tt], s_count = A.tk_off[tt + 1] - s_first;
            float* pl = A.part + (((size_t)s_first * kBT + (size_t)(wave * kColsPerWave + j) * s_count + (part_slot - s_first)) * 2 + h) * KS;
#pragma unroll
            for (int i = 0; i < KS; i += 4) *(f32x4*)(pl + i) = f32x4{tkv[i], tkv[i + 1], tkv[i + 2], tkv[i + 3]};
            continue;
        }
        // ---- this segment's partial: rows (m, l, numerators[d]) x 256 columns ----
        int tid_e = tid;
        asm volatile("" : "+v"(tid_e));
        float* part = A.part + ((size_t)part_slot * A.part_rows) * kBT + wave * kColsPerWave + (tid_e & 31);
        int hh = (tid_e >> 5) & 1;
        asm volatile("" : "+v"(part), "+v"(hh));   // row addresses are computed HERE: hoisted out of the segment loop as loop
                                                    // invariants they were 22 registers spilled to scratch at kernel entry
        const float lsum = half_sum(st.l);
        if (hh == 0) {
            part[0] = st.m;
            part[kBT] = lsum;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cls = acc_row(r, hh);
            if (cls < A.d) part[(size_t)(2 + cls) * kBT] = st.Y[r];
        }
    }
}

}  // namespace vosprop
