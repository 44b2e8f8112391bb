// The parity kernel (VOSPROP_PREC_F32): the same fused step as prop_bf16.h - affinity -> online column softmax -> spatial prior ->
// label product, reference src/model/predict.py:46-70 - with the arithmetic of the reference's CPU path, which is fp32 end to end:
//   * features stay f32 in the ring; S^T tile = 128 x v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: an exact fmaf chain, at the
//     f32 vector rate = 1/16 of the bf16 MFMA - this kernel is bounded by the f32 matrix peak, 157 TFLOP/s, not by 2.5 PFLOP/s);
//   * the spatial prior is evaluated in f32 from the same f32 coordinates the reference builds (predict.py:167-173: row = p / W as a
//     TRUE division, col = p % W; d2 = drow^2 + dcol^2; w = exp(-d2 / sigma^2)) and multiplies the un-normalised probability, as the
//     reference multiplies the softmax (predict.py:59-66) - no bilinear trick, no bf16 splits;
//   * exact online softmax (the running max is raised at every tile that needs it; no optimistic pass);
//   * the label product stays on the bf16 matrix core but loses nothing that matters: the weighted probability a is split into
//     bf16 hi + lo parts (a - hi is exact in f32; the two parts carry 16+ significant bits), one-hot labels are exact in bf16 and
//     probability labels are stored as hi + lo already: Y += L_hi a_hi + L_hi a_lo (+ L_lo a_hi + L_lo a_lo), error ~2^-17.
// Same work decomposition, segment table, partial format and combine_kernel as the bf16 path (common.h), so everything around the
// kernel - ring, plan, label packing, mask - is shared.  Tolerance met (tests/test_gpu_precision.py): <= 1e-4 relative on
// well-conditioned columns against the reference's un-rounded goldens, masks identical.
#pragma once
#include "common.h"
#include "prop_bf16.h"

namespace vosprop {

constexpr int kF32RowB = kC * 4 + 16;               // 1040-B padded LDS row: conflict-free ds_read_b128 down a column of rows
constexpr int kF32Pieces = 33;                      // 32 rows x 65 sixteen-byte units = 2 080 units = 32.5 one-KiB LDS-DMA pieces
constexpr int kF32Feat = kF32Pieces * 1024;         // 33 792
constexpr int kF32OffLabHi = kF32Feat;
constexpr int kF32OffLabLo = kF32Feat + kLdsLab;
constexpr int kF32OffCoord = kF32Feat + 2 * kLdsLab;   // 32 x (row, col) f32 = 256 B
constexpr int kF32Buf = kF32OffCoord + 256;         // 38 144; two buffers = 76 288 B of LDS

template <bool PROB, bool LAB_LO>
__global__ __launch_bounds__(kWaves * 64, 2) void prop_f32_kernel(const PropArgs A) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * kF32Buf];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;       // MFMA row (A operand) / column (B operand, C/D)
    const int h = lane >> 5;       // k-half of the operands / row-half of the accumulator
    const int TPF = A.tiles_per_frame;
    const int N = A.n_ref;
    const float c = A.c;
    const bool ragged = A.HW != A.HWp;
    const int rows_last = A.HW - (TPF - 1) * kTileR;

    const int seg0 = A.seg_off[blockIdx.x], seg1 = A.seg_off[blockIdx.x + 1];
    for (int si = seg0; si < seg1; ++si) {
        const Segment sg = A.segs[si];
        const int tt = __builtin_amdgcn_readfirstlane(sg.tt);
        const int r_lo = __builtin_amdgcn_readfirstlane(sg.r_lo);
        const int n_steps = __builtin_amdgcn_readfirstlane(sg.n_steps);
        const int part_slot = __builtin_amdgcn_readfirstlane(sg.slot);

        // target (B operand): MFMA step ks of this kernel contracts channels {ks, 128 + ks} (any pairing is a valid order of the
        // dot product as long as A and B agree): the k-half h of a lane owns channels [128 h, 128 h + 128) of its column
        const int t = tt * kBT + wave * kColsPerWave + j;
        const int t_ld = t < A.HWp ? t : A.HWp - 1;
        const float* trow = A.feat_f32 + ((size_t)A.target_slot * A.HWp + t_ld) * kC + h * 128;
        float Bt[128];
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const f32x4 v = *(const f32x4*)(trow + q * 4);
            Bt[4 * q] = v[0]; Bt[4 * q + 1] = v[1]; Bt[4 * q + 2] = v[2]; Bt[4 * q + 3] = v[3];
        }
        const int tq = t < A.HW ? t : A.HW - 1;
        const float2 tc = A.coord_f32[tq];          // (row, col) of the target pixel, the reference's f32 values

        ColState st;
        st.m = kNegBig;
        st.l = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) st.Y[r] = 0.0f;

        // ---- staging = LDS-DMA (global_load_lds: no staging registers), one tile ahead, one barrier per tile.  The padded row image is
        // produced by the per-lane SOURCE offsets (LDS unit q of 16 B holds chunk q % 65 of row q / 65, chunk 64 = padding).  Wave w
        // issues feature pieces w, w + 8, w + 16, w + 24; piece 32 wave 0; label hi waves 1-2, label lo waves 3-4, coordinates wave 5.
        typedef __attribute__((address_space(3))) void* lds_ptr;
        typedef const __attribute__((address_space(1))) void* glb_ptr;
        auto feat_src_off = [&](int piece) -> unsigned {
            int qq = 64 * piece + lane;
            if (qq >= kTileR * 65) qq = 0;      // lanes past the image (piece 32, lanes 32-63): any valid source, lands in slack
            int row = qq / 65, ch = qq - row * 65;
            if (ch == 64) ch = 63;
            return (unsigned)(row * 1024 + ch * 16);
        };
        const unsigned so0 = feat_src_off(wave), so1 = feat_src_off(wave + 8), so2 = feat_src_off(wave + 16),
                       so3 = feat_src_off(wave + 24), so4 = feat_src_off(32);
        auto stage = [&](int step, int buf) {
            const int rt = r_lo + step;               // pixel tile by pixel tile, the N frames inner (as prop_bf16.h)
            const int tile = rt / N, n = rt - tile * N;
            const int slot = A.slot[n];
            unsigned char* lds = smem + buf * kF32Buf;
            const unsigned char* f = (const unsigned char*)A.feat_f32 + ((size_t)slot * A.HWp + (size_t)tile * kTileR) * kC * 4;
            __builtin_amdgcn_global_load_lds((glb_ptr)(f + so0), (lds_ptr)(lds + wave * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr)(f + so1), (lds_ptr)(lds + (wave + 8) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr)(f + so2), (lds_ptr)(lds + (wave + 16) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr)(f + so3), (lds_ptr)(lds + (wave + 24) * 1024), 16, 0, 0);
            const size_t lab_off = ((size_t)slot * TPF + tile) * kLdsLab;
            if (wave == 0) {
                __builtin_amdgcn_global_load_lds((glb_ptr)(f + so4), (lds_ptr)(lds + 32 * 1024), 16, 0, 0);
            } else if (wave == 1 || wave == 2) {
                __builtin_amdgcn_global_load_lds((glb_ptr)((const unsigned char*)A.lab_hi + lab_off + (wave - 1) * 1024 + lane * 16),
                                                 (lds_ptr)(lds + kF32OffLabHi + (wave - 1) * 1024), 16, 0, 0);
            } else if ((wave == 3 || wave == 4) && LAB_LO) {
                __builtin_amdgcn_global_load_lds((glb_ptr)((const unsigned char*)A.lab_lo + lab_off + (wave - 3) * 1024 + lane * 16),
                                                 (lds_ptr)(lds + kF32OffLabLo + (wave - 3) * 1024), 16, 0, 0);
            } else if (wave == 5 && !PROB) {
                __builtin_amdgcn_global_load_lds((glb_ptr)((const unsigned char*)(A.coord_f32 + (size_t)tile * kTileR) + lane * 4),
                                                 (lds_ptr)(lds + kF32OffCoord), 4, 0, 0);
            }
        };
        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        int ctile = r_lo / N, cn = r_lo - ctile * N;
        bool sparse = (A.sparse_mask >> cn) & 1ull;
        for (int p = 0; p < n_steps; ++p) {
            const unsigned char* lb = smem + (p & 1) * kF32Buf;
            if (p + 1 < n_steps) stage(p + 1, (p + 1) & 1);
            // ---- S^T tile: 128 f32 MFMAs, A fragments straight from LDS ----
            f32x16 S;
#pragma unroll
            for (int r = 0; r < 16; ++r) S[r] = 0.0f;
            const unsigned char* arow = lb + j * kF32RowB + h * 512;
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                const f32x4 a4 = *(const f32x4*)(arow + q * 16);
                S = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0], Bt[4 * q], S, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[1], Bt[4 * q + 1], S, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[2], Bt[4 * q + 2], S, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[3], Bt[4 * q + 3], S, 0, 0, 0);
            }
            const bf16x8 lh0 = *(const bf16x8*)(lb + kF32OffLabHi + lane * 16);
            const bf16x8 lh1 = *(const bf16x8*)(lb + kF32OffLabHi + 1024 + lane * 16);
            bf16x8 ll0, ll1;
            if (LAB_LO) {
                ll0 = *(const bf16x8*)(lb + kF32OffLabLo + lane * 16);
                ll1 = *(const bf16x8*)(lb + kF32OffLabLo + 1024 + lane * 16);
            }
            if (ragged && ctile == TPF - 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (acc_row(r, h) >= rows_last) S[r] = kNegBig;
            }
            // ---- exact online softmax ----
            float sv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) sv[r] = S[r];
            const float tmax = half_max(max16v(sv));
            const float mn = fmaxf(st.m, tmax);
            const float sc = __builtin_amdgcn_exp2f((st.m - mn) * c);      // 1 when the max did not move, 0 for the first tile
            st.m = mn;
            st.l *= sc;
#pragma unroll
            for (int r = 0; r < 16; ++r) st.Y[r] *= sc;
            const float mc = mn * c;
            // -1 / sigma^2, one correctly rounded division per tile; the 16 entries then multiply by it (the reference divides each
            // entry: <= 1 ulp apart in the exponent, |rel| <= 1.2e-7 |x| in w)
            const float ninv_s2 = -1.0f / (sparse ? A.sig2_sq : A.sig1_sq);
            float a[16];
            float l0 = 0.0f, l1 = 0.0f;
            const float2* ct = (const float2*)(lb + kF32OffCoord);
#pragma unroll
            for (int half = 0; half < 2; ++half) {      // two halves: keeps the coordinate / exponent temporaries short-lived
#pragma unroll
                for (int r = half * 8; r < half * 8 + 8; r += 2) {
                    const float pa = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r], c, -mc));
                    const float pb = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r + 1], c, -mc));
                    l0 += pa;
                    l1 += pb;
                    if (PROB) {
                        a[r] = pa;
                        a[r + 1] = pb;
                    } else {
                        // reference predict.py:167-173, f32 op for op: diff = coords[t] - coords[r]; d2 = diff0^2 + diff1^2;
                        // w = exp(-d2 / sigma^2) (exp through v_exp_f32: |rel err| <= 6e-8 |x|, far inside the 1e-4 budget)
                        const float2 ra = ct[acc_row(r, h)], rb = ct[acc_row(r + 1, h)];
                        const float da = tc.x - ra.x, db = tc.y - ra.y;
                        const float ea = tc.x - rb.x, eb = tc.y - rb.y;
                        const float d2a = da * da + db * db, d2b = ea * ea + eb * eb;
                        a[r] = pa * __expf(d2a * ninv_s2);
                        a[r + 1] = pb * __expf(d2b * ninv_s2);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            st.l += l0 + l1;
            // ---- label product: a = hi + lo in bf16 (a - hi is exact), all on the bf16 matrix core ----
            bf16x8 h0, h1, o0, o1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                h0[e] = (bf16_t)a[e];
                h1[e] = (bf16_t)a[8 + e];
                o0[e] = (bf16_t)(a[e] - (float)h0[e]);
                o1[e] = (bf16_t)(a[8 + e] - (float)h1[e]);
            }
            st.Y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lh0, h0, st.Y, 0, 0, 0);
            st.Y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lh1, h1, st.Y, 0, 0, 0);
            st.Y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lh0, o0, st.Y, 0, 0, 0);
            st.Y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lh1, o1, st.Y, 0, 0, 0);
            if (LAB_LO) {
                st.Y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ll0, h0, st.Y, 0, 0, 0);
                st.Y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ll1, h1, st.Y, 0, 0, 0);
                st.Y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ll0, o0, st.Y, 0, 0, 0);
                st.Y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ll1, o1, st.Y, 0, 0, 0);
            }
            if (++cn == N) {
                cn = 0;
                ++ctile;
            }
            sparse = (A.sparse_mask >> cn) & 1ull;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile p + 1 have landed
            __syncthreads();
        }

        // ---- this segment's partial: rows (m, l, numerators[d]) x 256 columns, the format of the bf16 kernel ----
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
