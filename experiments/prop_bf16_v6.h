// prop_bf16_v6_kernel - dense one-hot label propagation (PROB = false, no top-k), third structure.  EXPERIMENTAL: selected
// with VOSPROP_V6=1, bit-identical results to prop_bf16_kernel, same speed (240 vs 237 us on the bench clip, 270 vs 273 us on
// random features) - kept because it isolates, one at a time, what bounds this problem on a CDNA4 SIMD.
//
// What the round-1 ablations of prop_bf16_kernel (8 waves x 32 columns) showed on MI355X: MFMA ~91 us, vector ALU ~93 us and
// LDS fragment reads ~82 us add up to the measured ~270 us - nothing overlaps - and the LDS pipe alone (8 waves x 17 KB per
// tile at 128 B/clk) is at 90 % of the MFMA time.  v6 therefore
//   * gives every wave 64 target pixels (4 waves, one per SIMD, all 512 registers) and feeds BOTH 32-pixel column blocks from
//     each A fragment it reads (all 16 fragments of a tile stay resident in AGPRs): LDS fragment traffic is halved;
//   * runs block 0's MFMA chain with the DENOMINATOR work of the previous tile interleaved (one row pair per MFMA), then - the
//     rescale decision is known by then - block 1's chain with the NUMERATOR work and the label MFMAs interleaved;
//   * writes the softmax as 2-wide f32 vectors (v_pk_fma_f32 / v_pk_add_f32) and drops the per-tile max tree: the running max
//     is only revisited when a tile's partial denominator exceeds 2^8 (or overflowed);
//   * keeps target fragments, spatial B operands and the A fragments in AGPRs (LDS loads target the accumulation file
//     directly), everything the VALU touches in arch VGPRs;
//   * spreads the five LDS-DMA pieces a wave issues per tile over the k-steps, stages THREE tiles ahead in a 4-slot ring and
//     only waits for the pieces issued a whole step earlier (s_waitcnt vmcnt(5)), one fence-less barrier per tile;
//   * has every MFMA, LDS read and wait as inline asm (exact order and counts), two tiles per loop iteration.
// Measured per tile-step and wave (s_memtime stamps, 1.9-2.1 GHz under load): chains 2250 cycles for 32 + 4 MFMAs = 1150 cycles
// of matrix pipe - an MFMA and the transcendental half of a slot do not overlap inside one wave (tools/ubench_issue.hip:
// {MFMA, 3 VALU, 2 EXP} = 41 cycles per 32-cycle MFMA; here 2 exp + 2 packed ops + LDS/DMA issue = ~60) - plus ~800 cycles of
// step overhead (S_w MFMAs, LDS latency, barrier, bookkeeping).
// Pitfalls found on the way (all fixed below, each with a note where it bites):
//   - a parity branch inside the loop makes hipcc reconcile the two bodies' register assignments with ~150 copies per step;
//   - hipcc's waitcnt insertion falls back to lgkmcnt(0) for loads consumed one loop iteration later -> LDS reads as asm;
//   - asm loads are asynchronous but hipcc does not know: a destination that is dead in its eyes is reused before the data
//     lands -> every waiting statement takes the loaded registers as operands;
//   - pure VALU code moves freely across asm volatile and sched_barrier (IR level) -> empty asm statements pin values;
//   - two independent MFMAs back to back block the wave's issue for 32 cycles; s_nop N costs 4*(N+1) cycles, not N+1.
// Maths, staging image, work map and partial layout are those of prop_bf16_kernel.
#pragma once
#include "prop_bf16_v5.h"

namespace vosprop {

#ifndef V6_ABLATE
#define V6_ABLATE 0   // timing experiments: 1 no DMA, 2 no barrier, 4 no softmax, 8 no LDS fragment refills, 16 no chain MFMAs
#endif

constexpr int kW6 = 4;
constexpr int kRing6 = 4;
constexpr float kSumThr = 256.0f;   // 2^kRescaleThr

// Two floats handled as TWO SCALARS on purpose.  Packed f32 instructions (v_pk_fma_f32, v_pk_add_f32, v_pk_mul_f32) do not
// overlap with MFMAs on gfx950 - they behave as if they ran on the matrix pipe: {MFMA ; 4 packed ops} costs 66 cycles per slot,
// {MFMA ; add, add, exp, exp, fma, fma} costs 45 (tools/ubench_slot.hip).  A real 2-vector type would be selected as packed
// instructions, so this is a struct, and the library is built with -fno-slp-vectorize so that hipcc does not pack scalars again.
struct f32x2 {
    float v[2];
    __device__ __forceinline__ float& operator[](int i) { return v[i]; }
    __device__ __forceinline__ const float& operator[](int i) const { return v[i]; }
    __device__ __forceinline__ f32x2& operator+=(const f32x2& o) {
        v[0] += o.v[0];
        v[1] += o.v[1];
        return *this;
    }
};
__device__ __forceinline__ f32x2 v6_fma2(const f32x2& a, const f32x2& b, const f32x2& c) {
    f32x2 r;
    r.v[0] = __builtin_fmaf(a.v[0], b.v[0], c.v[0]);
    r.v[1] = __builtin_fmaf(a.v[1], b.v[1], c.v[1]);
    return r;
}
#define V6_PIN2(x) asm volatile("" : "+v"((x).v[0]), "+v"((x).v[1]))

// Every group of MFMAs is ONE asm statement: hipcc cannot see that the statements are matrix instructions, so any register
// copy it drops between two of them would sit inside their hazard windows.  Each statement starts with the wait states a
// VALU-written operand needs (copies hipcc places in front of it).
// label product (4 MFMAs): numerators in arch VGPRs, B operands were just written by v_cvt_pk
#define V6_MFMA_Y4(y0, y1, h0, h1, p00, p10, p01, p11)                                                     \
    asm volatile("s_nop 4\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %4, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %5, %1\n\t" \
                 "v_mfma_f32_32x32x16_bf16 %0, %3, %6, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %3, %7, %1"          \
                 : "+v"(y0), "+v"(y1) : "v"(h0), "v"(h1), "v"(p00), "v"(p10), "v"(p01), "v"(p11))
#define V6_MFMA_Y1(y, h, p) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(y) : "v"(h), "v"(p))
// score chains, one k-step of both column blocks: A and B operands all come from AGPRs - the A fragments are ds_read
// straight into the accumulation file (gfx90a+ LDS loads can target it), so the fragment ring costs no arch VGPRs
#define V6_MFMA1_FIRST(acc, a, b) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "a"(a), "a"(b))
#define V6_MFMA1(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "a"(b))
// S_w = S + spatial term for both column blocks, C read from another tuple: the caller guarantees >= 4 independent MFMAs
// (or 20 wait states) between the last MFMA of the chains and this statement
#define V6_MFMA2_SW(d0, d1, a, b0, b1, c0, c1)                                                                     \
    asm volatile("s_nop 4\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %3, %5\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %4, %6" \
                 : "=&v"(d0), "=&v"(d1) : "a"(a), "a"(b0), "a"(b1), "v"(c0), "v"(c1))
// LDS reads are inline asm too: hipcc's waitcnt insertion loses track of loads that are consumed one loop iteration later and
// falls back to s_waitcnt lgkmcnt(0) in front of every k-step, which serialises the 8-deep fragment ring with the LDS latency.
// With asm loads the counts are explicit: every step ends with lgkmcnt(0) (before the barrier), and everything loaded in a
// step (the 16 fragments of the next tile, its coordinate fragment, this tile's label fragments) is consumed in the next one.
#define V6_LDS_A(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(dst) : "v"(addr), "n"(off))
#define V6_LDS_V(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
// The destination registers of the asm LDS loads must stay allocated until the s_waitcnt that covers them: hipcc does not
// know the loads are asynchronous, and where a destination is dead in its eyes (the refills of the LAST step of a segment) it
// reuses the register at once - the data then lands on top of the new value.  So the waiting statements take every
// asynchronously loaded register as an in/out operand.
#define V6_LDS_TIES(fa, fx, lab)                                                                                          \
    "+a"(fa[0]), "+a"(fa[1]), "+a"(fa[2]), "+a"(fa[3]), "+a"(fa[4]), "+a"(fa[5]), "+a"(fa[6]), "+a"(fa[7]), "+a"(fa[8]),   \
        "+a"(fa[9]), "+a"(fa[10]), "+a"(fa[11]), "+a"(fa[12]), "+a"(fa[13]), "+a"(fa[14]), "+a"(fa[15]), "+a"(fx),        \
        "+v"(lab.h0), "+v"(lab.h1)
#define V6_WAIT64 asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory")
// the same, tied to the registers whose MFMA results are awaited: asm volatile does not order register-only VALU code, so
// without the operands hipcc may hoist the VALU accesses above the wait
#define V6_WAIT64_2(a, b) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(a), "+v"(b)::"memory")

// Raw scores of a tile (two column blocks).  S is double-buffered by tile parity (the chain of tile p+1 accumulates while the
// rows of tile p are read); S_w exists once: it is written after the previous tile's softmax has consumed it.
struct V6Tile {
    f32x16 S[2];
};
struct V6Sw {
    f32x16 w[2];
};

struct V6Soft {
    f32x2 l[2];
    bf16x8 pk[2][2];
};

// rows (2i', 2i'+1) of column block cb = i >> 3 of a finished tile, in two halves so that each half can sit behind its own
// MFMA (two MFMAs back to back block the wave's issue for a whole MFMA, 32 cycles, during which no VALU work proceeds):
//   den: pk_fma + 2 exp + pk_add      (denominator terms)
//   num: pk_fma + 2 exp + cvt_pk      (weighted numerator terms, packed to bf16 for the label MFMA)
__device__ __forceinline__ void v6_pair_den(const V6Tile& u, V6Soft& s, int i, f32x2 c2, const f32x2 (&nmc)[2]) {
    const int cb = i >> 3, r = 2 * (i & 7);
    f32x2 sv;
    sv[0] = u.S[cb][r];
    sv[1] = u.S[cb][r + 1];
    const f32x2 x = v6_fma2(sv, c2, nmc[cb]);
    f32x2 e;
    e[0] = __builtin_amdgcn_exp2f(x[0]);
    e[1] = __builtin_amdgcn_exp2f(x[1]);
    s.l[cb] += e;
}
__device__ __forceinline__ void v6_pair_num(const V6Sw& sw, V6Soft& s, int i, f32x2 c2, const f32x2 (&nmq)[2]) {
    const int cb = i >> 3, r = 2 * (i & 7);
    f32x2 wv;
    wv[0] = sw.w[cb][r];
    wv[1] = sw.w[cb][r + 1];
    const f32x2 y = v6_fma2(wv, c2, nmq[cb]);
    const float p0 = __builtin_amdgcn_exp2f(y[0]);
    const float p1 = __builtin_amdgcn_exp2f(y[1]);
    s.pk[cb][r >> 3][r & 7] = (bf16_t)p0;
    s.pk[cb][r >> 3][(r & 7) + 1] = (bf16_t)p1;
}
// The same work as three independent stages, so that a k-step can run stage A of pair k+1, stage B of pair k and stage C of
// pair k-1: no instruction of a slot depends on another one of the same slot (a transcendental's result may not be used by
// the next VALU instruction - hipcc pads that with s_nop, 4 cycles each, twice per slot in the unskewed form).
__device__ __forceinline__ f32x2 v6_den_a(const V6Tile& u, int i, f32x2 c2, const f32x2 (&nmc)[2]) {
    const int cb = i >> 3, r = 2 * (i & 7);
    f32x2 sv;
    sv[0] = u.S[cb][r];
    sv[1] = u.S[cb][r + 1];
    return v6_fma2(sv, c2, nmc[cb]);
}
__device__ __forceinline__ f32x2 v6_num_a(const V6Sw& sw, int i, f32x2 c2, const f32x2 (&nmq)[2]) {
    const int cb = i >> 3, r = 2 * (i & 7);
    f32x2 wv;
    wv[0] = sw.w[cb][r];
    wv[1] = sw.w[cb][r + 1];
    return v6_fma2(wv, c2, nmq[cb]);
}
__device__ __forceinline__ f32x2 v6_exp2(f32x2 x) {
    f32x2 e;
    e[0] = __builtin_amdgcn_exp2f(x[0]);
    e[1] = __builtin_amdgcn_exp2f(x[1]);
    return e;
}
__device__ __forceinline__ void v6_num_c(V6Soft& s, int i, f32x2 p) {
    const int cb = i >> 3, r = 2 * (i & 7);
    s.pk[cb][r >> 3][r & 7] = (bf16_t)p[0];
    s.pk[cb][r >> 3][(r & 7) + 1] = (bf16_t)p[1];
}
__device__ __forceinline__ void v6_pair(const V6Tile& u, const V6Sw& sw, V6Soft& s, int i, f32x2 c2, const f32x2 (&nmc)[2],
                                        const f32x2 (&nmq)[2]) {
    v6_pair_den(u, s, i, c2, nmc);
    v6_pair_num(sw, s, i, c2, nmq);
}

__global__ __launch_bounds__(kW6 * 64, 1) void prop_bf16_v6_kernel(const PropArgs A) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[kRing6 * kLdsBuf];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;
    const int h = lane >> 5;

    const int TPF = A.tiles_per_frame;
    const float c = A.c;
    f32x2 c2;
    c2[0] = c;
    c2[1] = c;
    const bool ragged = A.HW != A.HWp;
    const int rows_last = A.HW - (TPF - 1) * kTileR;

    // LDS-DMA roles: wave w issues feature pieces w, w+4, w+8, w+12; as its fifth piece wave 0 issues feature piece 16,
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

        float st_m[2] = {kNegBig, kNegBig};
        float st_l[2] = {0.0f, 0.0f};
        f32x16 Y0, Y1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            Y0[r] = 0.0f;
            Y1[r] = 0.0f;
        }

        // ---- staging: tile p lives in ring slot p & 3.  Branch-free inside the k-loop: every wave issues exactly five
        // pieces per step, always (past the end of the segment it stages tiles nobody reads, clamped to the reference stream),
        // so the chain stays one basic block and the deferred s_waitcnt count is a constant ----
        int sn = r_lo / TPF, stile = r_lo - sn * TPF;
        const unsigned char *f_base = nullptr, *base4 = nullptr;
        // fifth piece: wave 0 feature piece 16, wave 1 the coordinate piece, waves 2 / 3 the two label pieces
        const unsigned stride4 = wave == 0 ? (unsigned)kGlbFeat : wave == 1 ? (unsigned)kLdsCoord : (unsigned)kLdsLab;
        const unsigned off4 = wave == 0 ? src_off[4] : (unsigned)(lane * 16 + (wave == 3 ? 1024 : 0));
        const unsigned lds4 = wave == 0 ? 16u * 1024u : wave == 1 ? (unsigned)kOffCoord
                                                                  : (unsigned)(kOffLabHi + (wave == 3 ? 1024 : 0));
        auto stage_frame = [&]() {
            const int slot = A.slot[sn];
            f_base = (const unsigned char*)A.feat_ring + (size_t)slot * A.HWp * kC * 2;
            const unsigned char* lh_base = (const unsigned char*)A.lab_hi + (size_t)slot * TPF * kLdsLab;
            base4 = wave == 0 ? f_base : wave == 1 ? (const unsigned char*)A.coord_tab : lh_base;
        };
        typedef __attribute__((address_space(3))) void* lds_ptr;
        typedef const __attribute__((address_space(1))) void* glb_ptr;
        auto stage_piece = [&](int buf, int i) {   // i = 0..4, this wave's i-th piece of the tile being staged
            unsigned char* lds = smem + buf * kLdsBuf;
            if (i < 4) {
                const unsigned char* f = f_base + (size_t)stile * kGlbFeat;
                __builtin_amdgcn_global_load_lds((glb_ptr)(f + src_off[i]), (lds_ptr)(lds + (wave + 4 * i) * 1024), 16, 0, 0);
            } else {
                __builtin_amdgcn_global_load_lds((glb_ptr)(base4 + (size_t)stile * stride4 + off4), (lds_ptr)(lds + lds4), 16, 0, 0);
            }
        };
        auto stage_advance = [&]() {
            int ns = stile + 1, nn = sn;
            if (ns == TPF) {
                ns = 0;
                nn = sn + 1;
            }
            if (nn < A.n_ref) {
                stile = ns;
                if (nn != sn) {
                    asm volatile("; next staged frame" ::: "memory");
                    sn = nn;
                    stage_frame();
                }
            }
        };
        stage_frame();
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int i = 0; i < 5; ++i) stage_piece(p, i);
            stage_advance();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        int cn = r_lo / TPF, ctile = r_lo - cn * TPF;
        bool sparse = (A.sparse_mask >> cn) & 1ull;
        bool sparse_prev = sparse;

        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
        const unsigned frag_off = (unsigned)(j * kRowB + h * 16);          // A fragment of reference row j, k-half h
        const unsigned coord_off = (unsigned)(kOffCoord + h * 512 + j * 16);
        const unsigned lab_off = (unsigned)(kOffLabHi + lane * 16);
        // all 16 A fragments of a tile stay resident (64 AGPRs): block 0's chain runs through them, then block 1's chain, and
        // each fragment is re-filled from the NEXT tile right after block 1 has consumed it - a whole chain (>= 500 cycles)
        // before block 0 needs it again, and every step ends with lgkmcnt(0), so the chains never wait on LDS.
        // Two chains interleaved MFMA by MFMA measured 1.8x slower than the MFMA time: alternating accumulators cannot use the
        // matrix pipe's accumulator forwarding and the C/D traffic competes with the VALU for register-file ports.
        bf16x8 fa[16], faxA, faxB;   // coordinate fragments of even / odd tiles
        {
            const unsigned a0 = lds0 + frag_off;
            V6_LDS_A(fa[0], a0, 0);
            V6_LDS_A(fa[1], a0, 32);
            V6_LDS_A(fa[2], a0, 64);
            V6_LDS_A(fa[3], a0, 96);
            V6_LDS_A(fa[4], a0, 128);
            V6_LDS_A(fa[5], a0, 160);
            V6_LDS_A(fa[6], a0, 192);
            V6_LDS_A(fa[7], a0, 224);
            V6_LDS_A(fa[8], a0, 256);
            V6_LDS_A(fa[9], a0, 288);
            V6_LDS_A(fa[10], a0, 320);
            V6_LDS_A(fa[11], a0, 352);
            V6_LDS_A(fa[12], a0, 384);
            V6_LDS_A(fa[13], a0, 416);
            V6_LDS_A(fa[14], a0, 448);
            V6_LDS_A(fa[15], a0, 480);
            const unsigned c0 = lds0 + coord_off;
            V6_LDS_A(faxA, c0, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+a"(fa[0]), "+a"(fa[1]), "+a"(fa[2]), "+a"(fa[3]), "+a"(fa[4]), "+a"(fa[5]),
                         "+a"(fa[6]), "+a"(fa[7]), "+a"(fa[8]), "+a"(fa[9]), "+a"(fa[10]), "+a"(fa[11]), "+a"(fa[12]),
                         "+a"(fa[13]), "+a"(fa[14]), "+a"(fa[15]), "+a"(faxA)::"memory");
        }
        V6Tile T0, T1;
        V6Sw Sw;
        LabFrag<false> lab_prev;   // labels of the tile whose softmax runs under the current chain

        // Denominators first, numerators second: all 16 denominator row pairs of the previous tile run under block 0's chain,
        // so whether the running max must be raised is known BEFORE any numerator is formed.  The rare slow path then sits
        // between the two chains: it finds the true max of the bad block from the raw scores, rescales (l, Y) once and redoes
        // that block's denominators; the numerators (under block 1's chain) are simply computed against the new max.
        auto rescale = [&](const V6Tile& u, V6Soft& sf, f32x2 (&nmc)[2], f32x2 (&nmq)[2], int cb, float pkq) {
            float sr[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) sr[r] = u.S[cb][r];
            float mn = fmaxf(st_m[cb], half_max(max16v(sr)));
            asm volatile("" : "+v"(mn));
            const float scl = __builtin_amdgcn_exp2f((st_m[cb] - mn) * c);
            st_l[cb] *= scl;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (cb == 0) Y0[r] *= scl;
                else Y1[r] *= scl;
            }
            st_m[cb] = mn;
            nmc[cb][0] = nmc[cb][1] = -(mn * c);
            nmq[cb][0] = nmq[cb][1] = -(mn * c + pkq);
            sf.l[cb][0] = 0.0f;
            sf.l[cb][1] = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) v6_pair_den(u, sf, cb * 8 + i, c2, nmc);
        };
        auto den_check = [&](const V6Tile& u, V6Soft& sf, f32x2 (&nmc)[2], f32x2 (&nmq)[2], bool u_sparse) {
            const bool bad0 = sf.l[0][0] + sf.l[0][1] > kSumThr, bad1 = sf.l[1][0] + sf.l[1][1] > kSumThr;
            // NaN-safe: inf - inf never arises here (sums of non-negative terms), inf compares greater
            if (__any(bad0 || bad1)) {
                asm volatile("; rescale" ::: "memory");
                V6_WAIT64_2(Y0, Y1);   // label MFMAs of the previous step may still be writing Y
                if (__any(bad0)) rescale(u, sf, nmc, nmq, 0, u_sparse ? kq2[0] : kq1[0]);
                if (__any(bad1)) rescale(u, sf, nmc, nmq, 1, u_sparse ? kq2[1] : kq1[1]);
                asm volatile("s_nop 4" : "+v"(Y0), "+v"(Y1)::"memory");   // VALU-written Y -> MFMA SrcC
            }
            st_l[0] += sf.l[0][0] + sf.l[0][1];
            st_l[1] += sf.l[1][0] + sf.l[1][1];
        };

        STAMP_DECL;
#ifdef VOSPROP_STAMP
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
#endif
        auto step = [&](int p, V6Tile& cur, const V6Tile& prev, const bf16x8& fax, bf16x8& fax_n, auto sm_tag) {
            constexpr bool SM = decltype(sm_tag)::value;
            const unsigned lb = lds0 + (unsigned)((p & 3) * kLdsBuf);
            const unsigned lbn = lds0 + (unsigned)(((p + 1) & 3) * kLdsBuf);
            const unsigned nrow = lbn + frag_off;
            const int sbuf = (p + 3) & 3;
            f32x2 nmc[2], nmq[2];
            V6Soft sf;
            if (SM) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const float mc = st_m[cb] * c;
                    nmc[cb][0] = nmc[cb][1] = -mc;
                    nmq[cb][0] = nmq[cb][1] = -(mc + (sparse_prev ? kq2[cb] : kq1[cb]));
                    sf.l[cb][0] = 0.0f;
                    sf.l[cb][1] = 0.0f;
                }
            }
            STAMP_AT(0);   // 0: step prologue
            // block 0: {MFMA ; DMA piece ; exps of pair ks ; fma of pair ks+1 ; add of pair ks-1}
            f32x2 xn, ep;
            if (SM) xn = v6_den_a(prev, 0, c2, nmc);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                if (ks == 0) V6_MFMA1_FIRST(cur.S[0], fa[0], Bt[0][0]);
                else if (!(V6_ABLATE & 16)) V6_MFMA1(cur.S[0], fa[ks], Bt[0][ks]);
                if (ks % 3 == 1 && !(V6_ABLATE & 1)) stage_piece(sbuf, ks / 3);
                if (SM && !(V6_ABLATE & 4)) {
                    // the empty asm statements pin each stage inside its slot: volatile asm keeps its order relative to the
                    // MFMA statements, and the value it "modifies" has to be computed before it (pure code is otherwise moved
                    // across slots freely - sched_barrier only binds the machine scheduler)
                    // order inside the slot: add (pair ks-1), exps (pair ks), fma (pair ks+1).  Each empty asm "redefines" the
                    // input of the next stage, so the stages cannot be hoisted over one another
                    if (ks > 0) sf.l[(ks - 1) >> 3] += ep;
                    V6_PIN2(sf.l[0]);
                    V6_PIN2(sf.l[1]);
                    V6_PIN2(xn);
                    f32x2 e = v6_exp2(xn);
                    V6_PIN2(e);
                    V6_PIN2(nmc[(ks < 15 ? ks + 1 : ks) >> 3]);
                    if (ks < 15) {
                        xn = v6_den_a(prev, ks + 1, c2, nmc);
                        V6_PIN2(xn);
                    }
                    ep = e;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (SM && !(V6_ABLATE & 4)) sf.l[1] += ep;
            STAMP_AT(1);   // 1: chain 0 + denominators + DMA issue
            if (SM) den_check(prev, sf, nmc, nmq, sparse_prev);
            // block 1: {MFMA ; refill ; exps of pair ks ; fma of pair ks+1 ; pack of pair ks-1 ; label MFMA when a B operand is full}
            f32x2 yn, pp;
            if (SM) yn = v6_num_a(Sw, 0, c2, nmq);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                if (ks == 0) V6_MFMA1_FIRST(cur.S[1], fa[0], Bt[1][0]);
                else if (!(V6_ABLATE & 16)) V6_MFMA1(cur.S[1], fa[ks], Bt[1][ks]);
                if (SM) {   // rows 0-7 of block 0 are packed in slot 4, rows 8-15 in slot 8, block 1 in slots 12 / after
                    if (ks == 5) V6_MFMA_Y1(Y0, lab_prev.h0, sf.pk[0][0]);
                    if (ks == 9) V6_MFMA_Y1(Y0, lab_prev.h1, sf.pk[0][1]);
                    if (ks == 13) V6_MFMA_Y1(Y1, lab_prev.h0, sf.pk[1][0]);
                }
                if (!(V6_ABLATE & 8)) V6_LDS_A(fa[ks], nrow, ks * 32);
                if (SM && !(V6_ABLATE & 4)) {
                    if (ks > 0) {
                        v6_num_c(sf, ks - 1, pp);
                        asm volatile("" : "+v"(sf.pk[(ks - 1) >> 3][((ks - 1) >> 2) & 1]));
                        V6_PIN2(yn);
                    }
                    f32x2 pe = v6_exp2(yn);
                    V6_PIN2(pe);
                    V6_PIN2(nmq[(ks < 15 ? ks + 1 : ks) >> 3]);
                    if (ks < 15) {
                        yn = v6_num_a(Sw, ks + 1, c2, nmq);
                        V6_PIN2(yn);
                    }
                    pp = pe;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (SM && !(V6_ABLATE & 4)) v6_num_c(sf, 15, pp);
            if (SM) V6_MFMA_Y1(Y1, lab_prev.h1, sf.pk[1][1]);
            else asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // no MFMA in between: 20+ wait states before S_w
            {   // next step's label fragments and coordinate fragment: issued now, they land while the S_w MFMAs run
                const unsigned ca = lbn + coord_off, la = lb + lab_off;
                V6_LDS_V(lab_prev.h0, la, 0);   // slot p is re-staged (tile p+4) only after the barrier below
                V6_LDS_V(lab_prev.h1, la, 1024);
                V6_LDS_A(fax_n, ca, 0);
            }
            STAMP_AT(2);   // 2: chain 1 + numerators + label MFMAs
            if (sparse) V6_MFMA2_SW(Sw.w[0], Sw.w[1], fax, Bx2[0], Bx2[1], cur.S[0], cur.S[1]);
            else V6_MFMA2_SW(Sw.w[0], Sw.w[1], fax, Bx1[0], Bx1[1], cur.S[0], cur.S[1]);
            __builtin_amdgcn_sched_barrier(0);
            if (ragged && ctile == TPF - 1) {
                asm volatile("; tail tile" ::: "memory");
                V6_WAIT64_2(cur.S[0], cur.S[1]);   // MFMA results must have landed before the VALU overwrites them
                V6_WAIT64_2(Sw.w[0], Sw.w[1]);
                mask_tail_rows(cur.S[0], Sw.w[0], h, rows_last);
                mask_tail_rows(cur.S[1], Sw.w[1], h, rows_last);
            }
            sparse_prev = sparse;
            if (++ctile == TPF) {
                asm volatile("; next scored frame" ::: "memory");
                ctile = 0;
                ++cn;
                sparse = (A.sparse_mask >> cn) & 1ull;
            }
            // tile p+2 (issued during step p-1, or in the prologue) must have landed before the barrier; the pieces of
            // tile p+3 issued in this step may stay in flight (loads return in order)
            stage_advance();
            STAMP_AT(3);   // 3: S_w MFMAs, LDS loads, tail, bookkeeping
#ifdef VOSPROP_STAMP
            asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
            STAMP_AT(4);   // 4: DMA wait
#endif
            // no __syncthreads(): its fence would drain the look-ahead pieces (vmcnt(0)) every step
            if (!(V6_ABLATE & 2))
                asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" : V6_LDS_TIES(fa, fax_n, lab_prev)::"memory");
            else asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" : V6_LDS_TIES(fa, fax_n, lab_prev)::"memory");
            STAMP_AT(5);   // 5: barrier
        };

        // two tiles per loop iteration: with a parity branch inside the loop the two bodies end with different register
        // assignments and hipcc reconciles them with ~150 copies per step at the join (800 cycles, measured)
        step(0, T0, T1, faxA, faxB, TagF());
        int p = 1;
        for (; p + 1 < n_steps; p += 2) {
            step(p, T1, T0, faxB, faxA, TagT());
            step(p + 1, T0, T1, faxA, faxB, TagT());
        }
        if (p < n_steps) step(p, T1, T0, faxB, faxA, TagT());
        // softmax of the last tile: there are no scores left to hide it under
        auto finish = [&](V6Tile& u) {
            V6_WAIT64_2(u.S[0], u.S[1]);
            V6_WAIT64_2(Sw.w[0], Sw.w[1]);
            f32x2 nmc[2], nmq[2];
            V6Soft sf;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const float mc = st_m[cb] * c;
                nmc[cb][0] = nmc[cb][1] = -mc;
                nmq[cb][0] = nmq[cb][1] = -(mc + (sparse_prev ? kq2[cb] : kq1[cb]));
                sf.l[cb][0] = 0.0f;
                sf.l[cb][1] = 0.0f;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) v6_pair_den(u, sf, i, c2, nmc);
            den_check(u, sf, nmc, nmq, sparse_prev);
#pragma unroll
            for (int i = 0; i < 16; ++i) v6_pair_num(Sw, sf, i, c2, nmq);
            V6_MFMA_Y4(Y0, Y1, lab_prev.h0, lab_prev.h1, sf.pk[0][0], sf.pk[1][0], sf.pk[0][1], sf.pk[1][1]);
            V6_WAIT64_2(Y0, Y1);
            V6_WAIT64_2(Y0, Y1);
        };
        if ((n_steps - 1) & 1) finish(T1);
        else finish(T0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the look-ahead pieces nobody reads
        __syncthreads();   // the ring is re-staged by the next segment

#ifdef VOSPROP_STAMP
        if (A.dbg && lane == 0)
            for (int k = 0; k < VOSPROP_NSTAMP; ++k)
                atomicAdd(&A.dbg[((size_t)blockIdx.x * kWaves + wave) * VOSPROP_NSTAMP + k], tsum[k]);
#endif
        // ---- this segment's partial ----
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const float lsum = half_sum(st_l[cb]);
            float* part = A.part + ((size_t)part_slot * A.part_rows) * kBT
                          + wave * 64 + cb * 32 + j;
            if (h == 0) {
                part[0] = st_m[cb];
                part[kBT] = lsum;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cls = acc_row(r, h);
                if (cls < A.d) part[(size_t)(2 + cls) * kBT] = cb ? Y1[r] : Y0[r];
            }
        }
    }
}

}  // namespace vosprop
