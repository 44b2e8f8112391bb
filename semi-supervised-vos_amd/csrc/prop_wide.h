// The dense propagation kernel in its WIDE shape: four waves per workgroup - ONE per SIMD - each owning 64 target columns (two
// 32-column MFMA blocks), for the form the per-frame step runs (label mode, mask only: prop_dense.h NEED_L = false, FUSED).
//
// Same arithmetic, LDS ring, staging pieces, work map and partial format as prop_dense_kernel<false, false, 0, false> - a workgroup
// still covers one 256-column target tile and walks the same segments - another distribution of the tile step over waves.  Why
// (tools/ubench_wide.hip, profiles/r03_ubench_step_skeleton.txt): the chain itself runs at the matrix pipe's rate in both shapes,
// but everything a step does BESIDE its chain - the cursor's scalar arithmetic, the LDS-DMA pieces, the label fragments, the alarm
// branch, the barrier - is paid per WAVE and step and hides under nothing; here a wave's step carries 32 score MFMAs instead of 16,
// so that overhead is paid half as often per MFMA, every A-operand fragment read from LDS feeds two MFMAs (half the LDS bytes),
// the two chains of a wave (independent accumulators) never wait on each other, and no two waves compete for one matrix pipe.
// One wave per SIMD owns the SIMD's 512 registers: the 128 registers of the two target-fragment sets go where the matrix pipe
// can read them (AGPRs), the softmax state stays in the 256 architectural ones.
#pragma once
#include "common.h"
#include "prop_bf16.h"
#include "prop_dense.h"

namespace vosprop {

#ifndef VOSPROP_WABLATE
#define VOSPROP_WABLATE 0   // timing experiments only (results wrong): 1 no alarm / label MFMAs, 2 no rare block, 4 no staging, 8 no barrier, 16 no softmax rows, 32 no fragment refills, 64 no score MFMAs
#endif
#ifndef VOSPROP_WIDE_ASM
#define VOSPROP_WIDE_ASM 1     // the chain's gaps as hand-ordered asm statements (0: the form hipcc schedules)
#endif
#ifndef VOSPROP_WIDE_SGB
#define VOSPROP_WIDE_SGB 1
#endif
constexpr int kWavesW = 4;        // waves per workgroup (one per SIMD)
constexpr int kBlocksW = 2;       // 32-column MFMA blocks per wave

__global__ __launch_bounds__(kWavesW * 64, 1) void prop_wide_kernel(const PropArgs A) {
    static_assert(kWavesW * kBlocksW * kColsPerWave == kBT, "a workgroup covers one target tile");
    __shared__ __attribute__((aligned(16))) unsigned char smem[kRing5 * kLdsBuf];
    __shared__ __attribute__((aligned(16))) bf16x8 s_bx[2][kBlocksW][kWavesW * 64];   // per-lane prior constants: [sigma][block][thread]
    __shared__ float s_kq[2][kBlocksW][kWavesW * 64];

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

    // ---- staging: every wave issues five 1-KiB LDS-DMA pieces per tile - feature pieces w, w + 4, w + 8, w + 12 and a fifth:
    // wave 0 feature piece 16, wave 1 the coordinates, waves 2-3 the two halves of the label tile (prop_dense.h, kStageA scheme)
    auto feat_src_off = [&](int piece) -> unsigned {
        int qq = 64 * piece + lane;
        if (qq >= kTileR * 33) qq = 0;
        int row = qq / 33, ch = qq - row * 33;
        if (ch == 32) ch = 31;
        return (unsigned)(row * 512 + ch * 16);
    };
    const unsigned src_a = feat_src_off(wave), src_b = feat_src_off(wave + 4);
    const unsigned src_c = feat_src_off(wave + 8), src_d = feat_src_off(wave + 12);
    const size_t feat_slot_stride = (size_t)A.HWp * (kC * 2);
    const unsigned char* third_base = (const unsigned char*)A.feat_ring;
    size_t third_slot_stride = feat_slot_stride;
    unsigned third_tile_stride = kGlbFeat, third_lane = feat_src_off(16), third_lds = 16 * 1024;
    if (wave == 1) {
        third_base = (const unsigned char*)A.coord_tab;
        third_slot_stride = 0;
        third_tile_stride = kLdsCoord;
        third_lane = lane * 16;
        third_lds = kOffCoord;
    } else if (wave >= 2) {
        third_base = (const unsigned char*)A.lab_hi + (wave - 2) * 1024;
        third_slot_stride = (size_t)TPF * kLdsLab;
        third_tile_stride = kLdsLab;
        third_lane = lane * 16;
        third_lds = kOffLabHi + (wave - 2) * 1024;
    }
    third_tile_stride = __builtin_amdgcn_readfirstlane(third_tile_stride);
    third_lds = __builtin_amdgcn_readfirstlane(third_lds);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const unsigned smem_base = (unsigned)(size_t)(lds_ptr)smem;
    const unsigned my_slot = (unsigned)A.slot[lane];
    const unsigned fo = my_slot * (unsigned)feat_slot_stride;      // lane n: byte offset of sampled frame n in the feature ring
    const unsigned to = my_slot * (unsigned)third_slot_stride;     // ... and in this wave's fifth-piece array
    const unsigned char* const feat_base = (const unsigned char*)A.feat_ring;

    const int seg0 = A.seg_off[blockIdx.x], seg1 = A.seg_off[blockIdx.x + 1];
    for (int si = seg0; si < seg1; ++si) {
        const Segment sg = A.segs[si];
        const int tt = __builtin_amdgcn_readfirstlane(sg.tt);
        const int r_lo = __builtin_amdgcn_readfirstlane(sg.r_lo);
        const int n_steps = __builtin_amdgcn_readfirstlane(sg.n_steps);
        const int part_slot = __builtin_amdgcn_readfirstlane(sg.slot);
        STAMP_DECL;
#ifdef VOSPROP_STAMP
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
#endif

        // per-segment values from an opaque copy of the thread id (kept out of the hoisted loop invariants, see prop_dense.h)
        int tid_l = tid, wd_l = A.Wd;
        asm volatile("" : "+v"(tid_l), "+s"(wd_l));
        const int j_l = tid_l & 31, h_l = (tid_l >> 5) & 1;
        // target (B operand) fragments: 2 x 32 columns x 256 channels per wave
        bf16x8 Bt[kBlocksW][16];
#pragma unroll
        for (int b = 0; b < kBlocksW; ++b) {
            const int t = tt * kBT + (wave * kBlocksW + b) * kColsPerWave + j_l;
            const int t_ld = t < A.target_rows ? t : A.target_rows - 1;
            const bf16_t* trow = A.target_feat + (size_t)t_ld * kC + h_l * 8;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) Bt[b][ks] = *(const bf16x8*)(trow + ks * 16);
            // target-side spatial channels for both sigmas and the per-column constants g Q_t c -> LDS
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
                bf16x8 B;   // pairs with the reference-side table of engine.hip build_coord_table (prop_bf16.h has the derivation)
                B[0] = (bf16_t)(h_l ? kl : ah);
                B[1] = (bf16_t)(h_l ? kh : am);
                B[2] = (bf16_t)(h_l ? km : al);
                B[3] = (bf16_t)(h_l ? kh : bh);
                B[4] = (bf16_t)(h_l ? 0.0f : bm);
                B[5] = (bf16_t)(h_l ? 0.0f : bl);
                B[6] = (bf16_t)(h_l ? 0.0f : kh);
                B[7] = (bf16_t)(h_l ? 0.0f : km);
                s_bx[sgm][b][tid_l] = B;
                s_kq[sgm][b][tid_l] = (float)(g * qt * (double)c);
            }
        }

        float m[kBlocksW];          // running max of the raw scores of this lane's column (shared by the two half-waves)
        f32x16 Y[kBlocksW];         // numerators: rows = classes
        float Wt[kBlocksW][16];     // LM = log2 w - m c of the tile being finished
#pragma unroll
        for (int b = 0; b < kBlocksW; ++b) {
            m[b] = kNegBig;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                Y[b][r] = 0.0f;
                Wt[b][r] = 0.0f;
            }
        }
        bool w_sparse = false;

        // ---- staging cursor (frame inner) ----
        int sn = 0, stile = 0;
        unsigned so_feat = 0, so_third = 0;
        auto stage_bases = [&]() __attribute__((always_inline)) {
            so_feat = (unsigned)__builtin_amdgcn_readlane((int)fo, sn) + (unsigned)stile * (unsigned)kGlbFeat;
            so_third = (unsigned)__builtin_amdgcn_readlane((int)to, sn) + (unsigned)stile * third_tile_stride;
        };
        auto stage_piece = [&](unsigned lds, int i) __attribute__((always_inline)) {
            if (i == 0) glds16s2(src_a, so_feat, feat_base, lds, (unsigned)wave * 1024);
            else if (i == 1) glds16s2(src_b, so_feat, feat_base, lds, ((unsigned)wave + 4) * 1024);
            else if (i == 2) glds16s2(src_c, so_feat, feat_base, lds, ((unsigned)wave + 8) * 1024);
            else if (i == 3) glds16s2(src_d, so_feat, feat_base, lds, ((unsigned)wave + 12) * 1024);
            else glds16s2(third_lane, so_third, third_base, lds, third_lds);
        };
        auto stage_advance = [&]() __attribute__((always_inline)) {   // next tile of the stream; stays on the last one at its end
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
        // "tile -1" has probabilities 0 and takes its labels from the last slot: zero them (stale bits could spell a NaN)
        float zf = 0.0f;
        asm volatile("" : "+v"(zf));
        if (tid_l < 2 * kLdsLab / 16) *(f32x4*)(smem + kRingLast + kOffLabHi + tid_l * 16) = f32x4{zf, zf, zf, zf};
        stile = r_lo / N;
        sn = r_lo - stile * N;
        for (int q = 0; q < 3; ++q) {   // prologue: tiles 0, 1, 2
            stage_bases();
#pragma unroll
            for (int i = 0; i < 5; ++i) stage_piece(smem_base + q * kLdsBuf, i);
            stage_advance();
        }
#pragma unroll
        for (int b = 0; b < kBlocksW; ++b)
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) asm volatile("" : "+v"(Bt[b][ks]));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (A.target_f16) {      // an f16 encoder's features, read where it left them (prop_dense.h)
            typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
            for (int b = 0; b < kBlocksW; ++b)
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    const f16x8 hv = __builtin_bit_cast(f16x8, Bt[b][ks]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) Bt[b][ks][e] = (bf16_t)(float)hv[e];
                }
        }
        // the target fragments live in ACCUMULATION registers from here on (the matrix pipe reads its B operand from either file;
        // nothing else ever touches them): the 256 architectural registers stay free for the softmax state
#pragma unroll
        for (int b = 0; b < kBlocksW; ++b)
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) asm volatile("" : "+a"(Bt[b][ks]));
        __syncthreads();
#ifdef VOSPROP_STAMP
        STAMP_AT(7);   // 7: segment prologue
        tsum[10] += 1;
#endif

        // ---- the stream as a per-step CONTROL TABLE: lane i of three registers holds what step base + i needs - the byte offsets
        // of the tile it stages (stream position + 3, held at the stream's end) in the feature ring and in this wave's fifth-piece
        // array, and a flag word (1: tile is a frame's ragged last tile, 2: it needs a prior tile, 4: its frame's sigma class,
        // 8: last entry, refill).  A step fetches its three entries with v_readlane and does ONE rare branch; the ~30 dependent
        // scalar instructions of two walking cursors (pixel tile / frame / sigma bookkeeping, ~10 cycles each with nothing to hide
        // under on a SIMD with one wave) are gone from the loop; the table is rebuilt by vector arithmetic every 64 steps.
        unsigned t_feat = 0, t_third = 0, t_flags = 0;
        auto ctl_refill = [&](int base) __attribute__((always_inline)) {
            const int q = base + lane;
            int ps = r_lo + q + 3;
            const int last = TPF * N - 1;
            ps = ps < last ? ps : last;
            const int st_tile = ps / N, st_n = ps - st_tile * N;
            t_feat = (unsigned)__shfl((int)fo, st_n) + (unsigned)st_tile * (unsigned)kGlbFeat;
            t_third = (unsigned)__shfl((int)to, st_n) + (unsigned)st_tile * third_tile_stride;
            const int pc = r_lo + q;
            const int c_tile = pc / N, c_n = pc - c_tile * N;
            const unsigned sp = (unsigned)((A.sparse_mask >> c_n) & 1ull);
            const unsigned sp_prev = c_n > 0 ? (unsigned)((A.sparse_mask >> (c_n - 1)) & 1ull) : sp;
            const bool nw = q == 0 || c_n == 0 || sp != sp_prev;
            t_flags = ((ragged && c_tile == TPF - 1) ? 1u : 0u) | (nw ? 2u : 0u) | (sp << 2) | (lane == 63 ? 8u : 0u);
        };
        ctl_refill(0);
        int idx = 0, ctl_base = 0;
        // entries of the COMING step, fetched in gap 13 of the step before (the three lane reads and what hangs on them are
        // dependent instructions: between two chains every one of them costs ~10 cycles with nothing to fill the slots)
        unsigned flags = (unsigned)__builtin_amdgcn_readlane((int)t_flags, 0);
        so_feat = (unsigned)__builtin_amdgcn_readlane((int)t_feat, 0);
        so_third = (unsigned)__builtin_amdgcn_readlane((int)t_third, 0);

        AFrag<false> fr;
        fr.prefetch(smem, j, h);
#if VOSPROP_WIDE_ASM
        // the loop refills these registers from asm statements hipcc's wait-count pass cannot see: make the values asm-defined here
        // too, or the pass keeps the prologue's eight reads "pending" around the back edge and drains the queue (lgkmcnt(0)) in
        // gap 7 of every step
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(fr.a[k]));
#endif
        f32x16 S0[kBlocksW], S1[kBlocksW];
#pragma unroll
        for (int b = 0; b < kBlocksW; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) S1[b][r] = -__builtin_inff();   // "tile -1": every probability 0
        int s_cur = 0, s_nxt = kLdsBuf, s_prv = kRingLast, s_stg = 3 * kLdsBuf;

        // the (rare) overflow alarm of tile p-1: the running maxima of BOTH blocks are raised, what was accumulated is rescaled once
        // and the tile's weights are redone against the new maxima
        auto alarm_fix = [&](const f32x16 (&Sp)[kBlocksW], const float (&lt0)[kBlocksW], bf16x8 (&pk0)[kBlocksW],
                             bf16x8 (&pk1)[kBlocksW]) __attribute__((always_inline)) {
            // (one compare for both blocks: `||` would make hipcc evaluate the second block's maximum lazily, under an exec mask)
            if (VOSPROP_UNLIKELY(__any(__builtin_fmaxf(lt0[0] - m[0], lt0[1] - m[1]) > kAlarmExp / c))) {
                asm volatile("; rescale" ::: "memory");
#pragma unroll
                for (int b = 0; b < kBlocksW; ++b) {
                    float sv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) sv[r] = Sp[b][r];
                    const float mn = fmaxf(m[b], half_max(max16v(sv)));
                    const float sc = __builtin_amdgcn_exp2f((m[b] - mn) * c);
#pragma unroll
                    for (int r = 0; r < 16; ++r) Y[b][r] *= sc;
                    m[b] = mn;
                    prior_tile<true>(smem + s_prv, j, h, s_bx[w_sparse ? 1 : 0][b][tid], c, s_kq[w_sparse ? 1 : 0][b][tid] + mn * c, Wt[b]);
                    float d0, d1;
                    softmax_rows<false, true>(Sp[b], Wt[b], c, mn * c, d0, d1, pk0[b], pk1[b]);
                }
            }
        };
        // The label product of a tile is PENDING for one step: its weights qk and labels labq wait here and its four MFMAs ride in
        // gaps 0-3 of the next chain (one wave per SIMD hides nothing that stands between two chains; ablation: 19 of 212 us).
        bf16x8 qk0[kBlocksW], qk1[kBlocksW];
        LabFrag<false> labq;
        {
            bf16x8 z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = (bf16_t)0.0f;
            labq.h0 = z;
            labq.h1 = z;
#pragma unroll
            for (int b = 0; b < kBlocksW; ++b) {
                qk0[b] = z;
                qk1[b] = z;
            }
        }
        auto pending_label_mfma = [&](int i) __attribute__((always_inline)) {
#if !(VOSPROP_WABLATE & 1)
            if (i == 0) Y[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(labq.h0, qk0[0], Y[0], 0, 0, 0);
            if (i == 1) Y[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(labq.h0, qk0[1], Y[1], 0, 0, 0);
            if (i == 2) Y[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(labq.h1, qk1[0], Y[0], 0, 0, 0);
            if (i == 3) Y[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(labq.h1, qk1[1], Y[1], 0, 0, 0);
#endif
        };

        // one step: scores of tile p into S (both blocks), softmax rows of tile p-1 (scores Sp) in the gaps of the chain
        auto step = [&](f32x16 (&S)[kBlocksW], const f32x16 (&Sp)[kBlocksW]) __attribute__((always_inline)) {
            const unsigned char* lb = smem + s_cur;
            const unsigned char* lbn = smem + s_nxt;
            const unsigned b_st = smem_base + (unsigned)s_stg;
            unsigned so_feat_n = 0, so_third_n = 0, flags_n = 0;
            int s_nxt_n = 0, s_stg_n = 0;
            const unsigned flags_now = flags;
            LabFrag<false> labp;
            float lt0[kBlocksW] = {kNegBig, kNegBig};
            float qprev[kBlocksW] = {0.0f, 0.0f};
            bf16x8 pk0[kBlocksW], pk1[kBlocksW];
            const unsigned char* arow = lb + j * kRowB + h * 16;
            const unsigned char* nrow = lbn + j * kRowB + h * 16;
#if VOSPROP_WIDE_ASM
            const unsigned a_addr = smem_base + (unsigned)s_cur + (unsigned)(j * kRowB + h * 16);      // LDS byte addresses
            const unsigned n_addr = smem_base + (unsigned)s_nxt + (unsigned)(j * kRowB + h * 16);
            const unsigned lab_addr = smem_base + (unsigned)s_prv + (unsigned)(kOffLabHi + lane * 16);
            unsigned pkw[kBlocksW][8];
#else
#pragma unroll
            for (int b = 0; b < kBlocksW; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) S[b][r] = 0.0f;
#endif
#ifdef VOSPROP_STAMP
            STAMP_AT(0);   // 0: step head
#endif
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
#ifdef VOSPROP_STAMP
                if (ks == 8) STAMP_AT(1);   // 1: gaps 0-7
#endif
#if VOSPROP_WIDE_ASM
                {
                    // ---- one gap, hand-ordered: MFMA (block 0) ; the two rows' fma ; first exponential ; MFMA (block 1) ; the
                    // fragment refill ; second exponential ; (odd gaps) the alarm's running maxima and the two packs.  Every
                    // instruction of the chain sits in a volatile asm statement: the ORDER is the one written here.  The LDS reads of
                    // these statements are invisible to hipcc's wait-count pass and are waited for HERE: a fragment was read eight
                    // gaps ago, seven refills (and at most the two label reads) are younger -> lgkmcnt(7) (LDS returns in order).
                    bf16x8& fa = fr.a[ks & 7];
                    const unsigned rd_addr = ks < 8 ? a_addr : n_addr;
                    float q0, q1;
                    if (ks == 0) {      // (the accumulators start from the inline constant 0: no zeroing)
                        asm volatile("s_waitcnt lgkmcnt(7)\n\t"
                                     "v_mfma_f32_32x32x16_bf16 %[s0], %[fa], %[b0], 0\n\t"
                                     "v_fma_f32 %[q0], %[sp0], %[c], %[w0]\n\t"
                                     "v_fma_f32 %[q1], %[sp1], %[c], %[w1]\n\t"
                                     "v_exp_f32 %[q0], %[q0]\n\t"
                                     "v_mfma_f32_32x32x16_bf16 %[s1], %[fa], %[b1], 0\n\t"
                                     "ds_read_b128 %[fa], %[ad] offset:%[off]\n\t"
                                     "v_exp_f32 %[q1], %[q1]"
                                     : [s0] "=&v"(S[0]), [s1] "=&v"(S[1]), [q0] "=&v"(q0), [q1] "=&v"(q1), [fa] "+v"(fa)
                                     : [b0] "a"(Bt[0][ks]), [b1] "a"(Bt[1][ks]), [sp0] "v"(Sp[0][ks]), [sp1] "v"(Sp[1][ks]),
                                   [w0] "v"(Wt[0][ks]), [w1] "v"(Wt[1][ks]), [c] "s"(c), [ad] "v"(rd_addr), [off] "n"(((ks & 7) + (ks < 8 ? 8 : 0)) * 32)
                                     : "memory");
                        qprev[0] = q0;
                        qprev[1] = q1;
                    } else if (!(ks & 1)) {
                        asm volatile("s_waitcnt lgkmcnt(7)\n\t"
                                     "v_mfma_f32_32x32x16_bf16 %[s0], %[fa], %[b0], %[s0]\n\t"
                                     "v_fma_f32 %[q0], %[sp0], %[c], %[w0]\n\t"
                                     "v_fma_f32 %[q1], %[sp1], %[c], %[w1]\n\t"
                                     "v_exp_f32 %[q0], %[q0]\n\t"
                                     "v_mfma_f32_32x32x16_bf16 %[s1], %[fa], %[b1], %[s1]\n\t"
                                     "ds_read_b128 %[fa], %[ad] offset:%[off]\n\t"
                                     "v_exp_f32 %[q1], %[q1]"
                                     : [s0] "+v"(S[0]), [s1] "+v"(S[1]), [q0] "=&v"(q0), [q1] "=&v"(q1), [fa] "+v"(fa)
                                     : [b0] "a"(Bt[0][ks]), [b1] "a"(Bt[1][ks]), [sp0] "v"(Sp[0][ks]), [sp1] "v"(Sp[1][ks]),
                                   [w0] "v"(Wt[0][ks]), [w1] "v"(Wt[1][ks]), [c] "s"(c), [ad] "v"(rd_addr), [off] "n"(((ks & 7) + (ks < 8 ? 8 : 0)) * 32)
                                     : "memory");
                        qprev[0] = q0;
                        qprev[1] = q1;
                    } else {
                        unsigned w0, w1;
                        asm volatile("s_waitcnt lgkmcnt(7)\n\t"
                                     "v_mfma_f32_32x32x16_bf16 %[s0], %[fa], %[b0], %[s0]\n\t"
                                     "v_fma_f32 %[q0], %[sp0], %[c], %[w0]\n\t"
                                     "v_fma_f32 %[q1], %[sp1], %[c], %[w1]\n\t"
                                     "v_exp_f32 %[q0], %[q0]\n\t"
                                     "v_max3_f32 %[l0], %[l0], %[pp0], %[sp0]\n\t"
                                     "v_mfma_f32_32x32x16_bf16 %[s1], %[fa], %[b1], %[s1]\n\t"
                                     "ds_read_b128 %[fa], %[ad] offset:%[off]\n\t"
                                     "v_exp_f32 %[q1], %[q1]\n\t"
                                     "v_max3_f32 %[l1], %[l1], %[pp1], %[sp1]\n\t"
                                     "v_cvt_pk_bf16_f32 %[k0], %[qp0], %[q0]\n\t"
                                     "v_cvt_pk_bf16_f32 %[k1], %[qp1], %[q1]"
                                     : [s0] "+v"(S[0]), [s1] "+v"(S[1]), [q0] "=&v"(q0), [q1] "=&v"(q1), [fa] "+v"(fa),
                                       [l0] "+v"(lt0[0]), [l1] "+v"(lt0[1]), [k0] "=&v"(w0), [k1] "=&v"(w1)
                                     : [b0] "a"(Bt[0][ks]), [b1] "a"(Bt[1][ks]), [sp0] "v"(Sp[0][ks]), [sp1] "v"(Sp[1][ks]),
                                   [w0] "v"(Wt[0][ks]), [w1] "v"(Wt[1][ks]), [c] "s"(c), [ad] "v"(rd_addr), [off] "n"(((ks & 7) + (ks < 8 ? 8 : 0)) * 32),
                                       [pp0] "v"(Sp[0][ks - 1]), [pp1] "v"(Sp[1][ks - 1]), [qp0] "v"(qprev[0]), [qp1] "v"(qprev[1])
                                     : "memory");
                        pkw[0][ks >> 1] = w0;
                        pkw[1][ks >> 1] = w1;
                    }
                    if (ks < 4) {      // the label product of tile p-2 (same statement kind: its place in the chain is fixed too)
                        f32x16& Yb = Y[ks & 1];
                        asm volatile("v_mfma_f32_32x32x16_bf16 %[y], %[l], %[k], %[y]"
                                     : [y] "+v"(Yb)
                                     : [l] "v"(ks < 2 ? labq.h0 : labq.h1), [k] "v"(ks < 2 ? qk0[ks & 1] : qk1[ks & 1]));
                    }
                    if (ks % 3 == 1) stage_piece(b_st, ks / 3);      // gaps 1, 4, 7, 10, 13: pieces 0..4 of tile p+3
                    if (ks == 10)      // labels of tile p-1 (waited for by the lgkmcnt(7) of the next chain's first gaps)
                        asm volatile("ds_read_b128 %[h0], %[ad]\n\tds_read_b128 %[h1], %[ad] offset:1024"
                                     : [h0] "=&v"(labp.h0), [h1] "=&v"(labp.h1)
                                     : [ad] "v"(lab_addr)
                                     : "memory");
                    if (ks == 5) {
                        s_nxt_n = s_nxt == kRingLast ? 0 : s_nxt + kLdsBuf;
                        s_stg_n = s_stg == kRingLast ? 0 : s_stg + kLdsBuf;
                    }
                    if (ks == 13) {
                        const int ix = (idx + 1) & 63;
                        flags_n = (unsigned)__builtin_amdgcn_readlane((int)t_flags, ix);
                        so_feat_n = (unsigned)__builtin_amdgcn_readlane((int)t_feat, ix);
                        so_third_n = (unsigned)__builtin_amdgcn_readlane((int)t_third, ix);
                    }
                    continue;
                }
#endif
#if VOSPROP_WABLATE & 64
                asm volatile("" : "+v"(fr.a[ks & 7]));
#else
                S[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.a[ks & 7], Bt[0][ks], S[0], 0, 0, 0);
                S[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.a[ks & 7], Bt[1][ks], S[1], 0, 0, 0);
#endif
                if (ks < 4) pending_label_mfma(ks);      // the label product of tile p-2
                if (ks == 5) {      // the ring one slot on (scalar; used after the barrier)
                    s_nxt_n = s_nxt == kRingLast ? 0 : s_nxt + kLdsBuf;
                    s_stg_n = s_stg == kRingLast ? 0 : s_stg + kLdsBuf;
                }
                if (ks == 13) {     // the coming step's table entries (lane idx + 1; a refill at the end of this step re-reads them)
                    const int ix = (idx + 1) & 63;
                    flags_n = (unsigned)__builtin_amdgcn_readlane((int)t_flags, ix);
                    so_feat_n = (unsigned)__builtin_amdgcn_readlane((int)t_feat, ix);
                    so_third_n = (unsigned)__builtin_amdgcn_readlane((int)t_third, ix);
                }
                // refill the fragment slot just consumed: second half of this tile, then the first half of the next one
#if !(VOSPROP_WABLATE & 32)
                if (ks < 8) fr.a[ks] = *(const bf16x8*)(arow + (ks + 8) * 32);
                else fr.a[ks - 8] = *(const bf16x8*)(nrow + (ks - 8) * 32);
#endif
#if !(VOSPROP_WABLATE & 4)
                if (ks % 3 == 1) stage_piece(b_st, ks / 3);      // gaps 1, 4, 7, 10, 13: pieces 0..4 of tile p+3
#endif
                if (ks == 10) labp.load(smem + s_prv, lane);     // labels of tile p-1
                // row ks of tile p-1, both blocks: a = 2^(S c + LM), packed in pairs; the alarm is a running max of the raw scores
#pragma unroll
                for (int b = 0; b < kBlocksW; ++b) {
#if VOSPROP_WABLATE & 16
                    const float q = Sp[b][ks];
                    if (ks == 15) { pk0[b] = fr.a[0]; pk1[b] = fr.a[1]; }
                    if (false)
#else
                    const float q = __builtin_amdgcn_exp2f(__builtin_fmaf(Sp[b][ks], c, Wt[b][ks]));
#endif
                    if (ks & 1) {
                        lt0[b] = __builtin_fmaxf(__builtin_fmaxf(lt0[b], Sp[b][ks - 1]), Sp[b][ks]);
                        if (ks < 8) { pk0[b][ks - 1] = (bf16_t)qprev[b]; pk0[b][ks] = (bf16_t)q; }
                        else { pk1[b][ks - 9] = (bf16_t)qprev[b]; pk1[b][ks - 8] = (bf16_t)q; }
                    } else {
                        qprev[b] = q;
                    }
                }
#ifndef VOSPROP_WIDE_NO_SGB
#if VOSPROP_WIDE_SGB == 1
                // MFMA ; one block's row ; MFMA ; the fragment read ; the other block's row (even gaps: fma + exp, odd: + max3 + pack)
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (ks & 1) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                else __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                if (ks < 4) __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                else __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (ks & 1) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                else __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                if (ks == 5 || ks == 13) __builtin_amdgcn_sched_group_barrier(0x006, 6, 0);      // the scalar / lane-read work placed there
#elif VOSPROP_WIDE_SGB == 2
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // 2 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
                if (ks & 1) __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                else __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
#else
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // 2 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);   // the two rows' vector instructions
#endif
#endif
            }
#if VOSPROP_WIDE_ASM
#pragma unroll
            for (int b = 0; b < kBlocksW; ++b) {
                pk0[b] = __builtin_bit_cast(bf16x8, u32x4{pkw[b][0], pkw[b][1], pkw[b][2], pkw[b][3]});
                pk1[b] = __builtin_bit_cast(bf16x8, u32x4{pkw[b][4], pkw[b][5], pkw[b][6], pkw[b][7]});
            }
#else
#pragma unroll
            for (int b = 0; b < kBlocksW; ++b) asm volatile("" : "+v"(pk0[b]), "+v"(pk1[b]));
#endif
#ifdef VOSPROP_STAMP
            STAMP_AT(2);   // 2: gaps 8-15
#endif
#if !(VOSPROP_WABLATE & 1)
            alarm_fix(Sp, lt0, pk0, pk1);
#endif
#pragma unroll
            for (int b = 0; b < kBlocksW; ++b) {
                qk0[b] = pk0[b];
                qk1[b] = pk1[b];
            }
            labq = labp;
#ifdef VOSPROP_STAMP
            STAMP_AT(3);   // 3: alarm check + label MFMAs
#endif
            ++idx;
            if (VOSPROP_UNLIKELY(flags_now & ((VOSPROP_WABLATE & 2) ? 8u : 11u))) {
                if (flags_now & 1u) {      // tile p: padded rows of a frame's last tile never enter the softmax
                    asm volatile("; tail tile" ::: "memory");
#pragma unroll
                    for (int b = 0; b < kBlocksW; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (acc_row(r, h) >= rows_last) S[b][r] = kNegBig;
                }
                if (flags_now & 2u) {      // the prior tile of tile p (used from the next step on; tile p-1 is finished)
                    asm volatile("; prior tile" ::: "memory");
                    const int sg_i = (flags_now >> 2) & 1u;
#pragma unroll
                    for (int b = 0; b < kBlocksW; ++b)
                        prior_tile<true>(lb, j, h, s_bx[sg_i][b][tid], c, s_kq[sg_i][b][tid] + m[b] * c, Wt[b]);
                    w_sparse = sg_i != 0;
                }
                if (flags_now & 8u) {      // the table's last entry: the next 64 steps
                    ctl_base += 64;
                    ctl_refill(ctl_base);
                    idx = 0;
                    flags_n = (unsigned)__builtin_amdgcn_readlane((int)t_flags, 0);
                    so_feat_n = (unsigned)__builtin_amdgcn_readlane((int)t_feat, 0);
                    so_third_n = (unsigned)__builtin_amdgcn_readlane((int)t_third, 0);
                }
            }
#ifdef VOSPROP_STAMP
            STAMP_AT(4);   // 4: the rare block (tail mask, prior tile, table refill)
#endif
            // this wave's pieces of tile p+2 have landed (the five of tile p+3 may stay in flight); the barrier makes every wave's
            // pieces of p+2 visible and retires the slot of tile p-2 for the DMA of step p+1
#if !(VOSPROP_WABLATE & 8)
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
#endif
#ifdef VOSPROP_STAMP
            STAMP_AT(5);   // 5: wait for the own pieces of tile p+2
#endif
#if !(VOSPROP_WABLATE & 8)
#if VOSPROP_WIDE_ASM
            asm volatile("s_barrier" ::: "memory");      // (the fragment reads of tile p+1 may stay in flight across it: nobody writes that slot)
#else
            __syncthreads();
#endif
#endif
#ifdef VOSPROP_STAMP
            STAMP_AT(6);   // 6: barrier
#endif
            s_prv = s_cur;
            s_cur = s_nxt;
            s_nxt = s_nxt_n;
            s_stg = s_stg_n;
            flags = flags_n;
            so_feat = so_feat_n;
            so_third = so_third_n;
        };

        int p = 0;
        for (; p + 1 < n_steps; p += 2) {
            step(S0, S1);
            step(S1, S0);
        }
        // the segment's last tile has no chain to hide under (its labels sit in slot s_prv after the last ring advance)
        auto drain = [&](const f32x16 (&Sp)[kBlocksW]) __attribute__((always_inline)) {
            bf16x8 pk0[kBlocksW], pk1[kBlocksW];
            float lt0[kBlocksW];
            LabFrag<false> labp;
            labp.load(smem + s_prv, lane);
#pragma unroll
            for (int b = 0; b < kBlocksW; ++b) {
                float d0, d1;
                softmax_rows<false, true>(Sp[b], Wt[b], c, m[b] * c, d0, d1, pk0[b], pk1[b]);
                float sv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) sv[r] = Sp[b][r];
                lt0[b] = max16v(sv);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) pending_label_mfma(i);      // tile n-2
            alarm_fix(Sp, lt0, pk0, pk1);
#pragma unroll
            for (int b = 0; b < kBlocksW; ++b) {
                qk0[b] = pk0[b];
                qk1[b] = pk1[b];
            }
            labq = labp;
#pragma unroll
            for (int i = 0; i < 4; ++i) pending_label_mfma(i);      // tile n-1
        };
        if (p < n_steps) {
            step(S0, S1);
            drain(S0);
        } else {
            drain(S1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the look-ahead pieces before the ring is re-staged
        __syncthreads();
#ifdef VOSPROP_STAMP
        STAMP_AT(8);
        if (A.dbg && lane == 0)
            for (int k = 0; k < VOSPROP_NSTAMP; ++k)
                atomicAdd(&A.dbg[((size_t)blockIdx.x * kWaves + wave) * VOSPROP_NSTAMP + k], tsum[k]);
#endif

        // ---- this segment's partial: rows (m, l = 0, numerators[d]) x 256 columns ----
        int tid_e = tid;
        asm volatile("" : "+v"(tid_e));
        const int hh = (tid_e >> 5) & 1;
#pragma unroll
        for (int b = 0; b < kBlocksW; ++b) {
            float* part = A.part + ((size_t)part_slot * A.part_rows) * kBT + (wave * kBlocksW + b) * kColsPerWave + (tid_e & 31);
            if (hh == 0) {
                part[0] = m[b];
                part[kBT] = 0.0f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cls = acc_row(r, hh);
                if (cls < A.d) part[(size_t)(2 + cls) * kBT] = Y[b][r];
            }
        }
    }
}

}  // namespace vosprop
