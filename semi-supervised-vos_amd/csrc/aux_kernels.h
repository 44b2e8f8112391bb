// Small HBM-bound kernels either side of the propagation kernel: feature push (NCHW -> pixel-major bf16),
// label packing into MFMA A-operand order, partial combine + argmax, nearest up-sampling of the mask.
#pragma once
#include "common.h"
#include <hip/hip_fp16.h>

namespace vosprop {

template <typename T> __device__ inline float to_f32(T v);
template <> __device__ inline float to_f32<float>(float v) { return v; }
template <> __device__ inline float to_f32<__half>(__half v) { return __half2float(v); }
template <> __device__ inline float to_f32<bf16_t>(bf16_t v) { return (float)v; }

// Replaces reference src/model/predict.py:47 (permute(0,2,3,1).reshape) + the history append
// (src/utils/inference_utils.py:72): (C, HW) channel-major -> ring slot [HWp][C] pixel-major bf16.
// grid = (ceil(HW/64), C/64), block = 256: one 64-pixel x 64-channel tile per block, transposed through LDS so that both
// the reads (256 B runs along pixels) and the writes (16 B per lane along channels) are coalesced.
// Rows >= HW of the slot stay zero (set once at allocation).
template <typename Out> struct Out8;
template <> struct Out8<bf16_t> {
    __device__ static inline void store(bf16_t* dst, const float (&v)[8]) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
        *(bf16x8*)dst = o;
    }
};
template <> struct Out8<float> {      // the f32 ring of the parity path (VOSPROP_PREC_F32): values pass through unrounded
    __device__ static inline void store(float* dst, const float (&v)[8]) {
        f32x4 a, b;
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = v[e]; b[e] = v[4 + e]; }
        *(f32x4*)dst = a;
        *(f32x4*)(dst + 4) = b;
    }
};

template <typename T, typename Out>
__global__ __launch_bounds__(256) void push_kernel(const T* __restrict__ src, Out* __restrict__ dst, int HW) {
    __shared__ float tile[64][65];
    const int p0 = blockIdx.x * 64, cc = blockIdx.y * 64;
    const int tid = threadIdx.x;
    const int p = tid & 63;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int ch = i * 4 + (tid >> 6);
        float v = 0.0f;
        if (p0 + p < HW) v = to_f32<T>(src[(size_t)(cc + ch) * HW + p0 + p]);
        tile[ch][p] = v;
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int px = pass * 32 + (tid >> 3);
        const int c8 = (tid & 7) * 8;
        if (p0 + px < HW) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = tile[c8 + e][px];
            Out8<Out>::store(dst + (size_t)(p0 + px) * kC + cc + c8, o);
        }
    }
}

// Pixel-major (channels-last) source: (HW, C) rows are already in ring order, only the element type changes.
template <typename T, typename Out>
__global__ __launch_bounds__(256) void push_hwc_kernel(const T* __restrict__ src, Out* __restrict__ dst, int n8) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = to_f32<T>(src[(size_t)i * 8 + e]);
    Out8<Out>::store(dst + (size_t)i * 8, o);
}

// Element (s, lane, e) of a label tile is L[class = lane & 31][row = 16 s + 8 (e >> 2) + 4 (lane >> 5) + (e & 3)]:
// the A-operand order that matches an accumulator tile reused as the B operand (see prop_bf16.h).
__device__ inline int lab_row(int s, int lane, int e) { return 16 * s + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3); }

// One-hot labels from a class-index map (reference index_to_onehot, src/utils/utils.py:59-68).
// One thread per 16-byte chunk: grid*block >= tiles*128.
__global__ void pack_cls_kernel(const uint8_t* __restrict__ cls, bf16_t* __restrict__ lab_hi, int HW, int tiles) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= tiles * 128) return;
    const int tile = gid >> 7, s = (gid >> 6) & 1, lane = gid & 63;
    const int k = lane & 31;
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int p = tile * kTileR + lab_row(s, lane, e);
        o[e] = (bf16_t)((p < HW && cls[p] == k) ? 1.0f : 0.0f);
    }
    *(bf16x8*)(lab_hi + (size_t)gid * 8) = o;
}

// General f32 labels L[k][p] (row stride ld floats between classes) -> hi (+ lo) bf16 parts.
__global__ void pack_f32_kernel(const float* __restrict__ L, size_t ld, int d, bf16_t* __restrict__ lab_hi,
                                bf16_t* __restrict__ lab_lo, uint8_t* __restrict__ cls, int HW, int tiles) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= tiles * 128) return;
    if (cls && gid < HW) {   // class index of pixel gid = arg-max of its label column (exact for one-hot labels)
        int best = 0;
        float bv = L[gid];
        for (int k = 1; k < d; ++k) {
            const float v = L[(size_t)k * ld + gid];
            if (v > bv) { bv = v; best = k; }
        }
        cls[gid] = (uint8_t)best;
    }
    const int tile = gid >> 7, s = (gid >> 6) & 1, lane = gid & 63;
    const int k = lane & 31;
    bf16x8 oh, ol;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int p = tile * kTileR + lab_row(s, lane, e);
        float v = 0.0f;
        if (p < HW && k < d) v = L[(size_t)k * ld + p];
        const float hi = bf16_round(v);
        oh[e] = (bf16_t)hi;
        ol[e] = (bf16_t)(v - hi);
    }
    *(bf16x8*)(lab_hi + (size_t)gid * 8) = oh;
    if (lab_lo) *(bf16x8*)(lab_lo + (size_t)gid * 8) = ol;
}

__device__ inline void pack_block_labels(const float (*outv)[64], const uint8_t* clsv, int d, int HW, int prob,
                                         bf16_t* __restrict__ lab_hi, bf16_t* __restrict__ lab_lo);

// Merge the partials of target pixels, normalise, arg-max, and write the new frame's labels in MFMA operand order.
//   out[k,t] = sum_u Y_u[k] 2^((m_u - M) c) / sum_u l_u 2^((m_u - M) c)      (reference predict.py:55-70)
//   cls[t]   = argmax_k out[k,t], first maximum wins (reference inference_utils.py:70, torch.argmax on CPU)
//   new label of the frame = one-hot(cls) or out itself in probability mode   (inference_utils.py:67-71)
// The partial slots that hold target tile tt are slots plist_off[tt] .. plist_off[tt+1] - 1: the host numbers the slots of a
// tile consecutively (engine.hip get_plan), so the slot list itself (plist) is the identity and is not read.
// grid = ceil(HW/64), block = 256 = 64 target pixels x 4 partial lanes (each lane folds every 4th partial with its own
// running max; the 4 lanes are merged through LDS), then the block's two 32-pixel label tiles are packed (256 chunks).
// Optional tail of combine_kernel: the nearest up-sampling of the block's 64 class indices into the full-size mask (reference
// inference_utils.py:74-75; argmax and nearest interpolation commute) - saves the separate up-sampling launch and its dispatch gap.
// y0 / x0: first output row / column whose ATen nearest source index (min(floor(dst * scale), in - 1), scale = (float)in / out
// computed on the host as ATen does) is >= i, for i = 0..Hd / 0..Wd; built on the host with the same float arithmetic.
struct UpArgs {
    uint8_t* mask;     // (H, W) or nullptr
    const int* y0;     // Hd + 1
    const int* x0;     // Wd + 1
    int H, W, Hd, Wd;
    float sx;
};

__global__ __launch_bounds__(256) void combine_kernel(const float* __restrict__ part, const int* __restrict__ plist_off,
                                                      const int* __restrict__ plist, int d, int HW, float c,
                                                      float* __restrict__ pred, uint8_t* __restrict__ cls,
                                                      bf16_t* __restrict__ lab_hi, bf16_t* __restrict__ lab_lo, int prob,
                                                      const UpArgs up, const uint4* __restrict__ cp_src,
                                                      uint4* __restrict__ cp_dst, int cp_n, int no_l) {
    __shared__ float red[4][kMaxClasses + 2][64];
    __shared__ float outv[kMaxClasses][64];
    __shared__ uint8_t clsv[64];
    const int tid = threadIdx.x, col = tid & 63, g = tid >> 6;
    // Optional rider: the ring copy of the frame just propagated (cp_n 16-byte units of channels-last bf16 features, caller's buffer
    // -> ring slot; vosprop_step).  The propagation kernel read the target from the caller's buffer; the slot is first needed as a
    // REFERENCE by the next step.  This kernel is latency-bound (three dependent round trips), eight independent 16-byte copies per
    // thread in front of them cost nothing and save the separate copy launch and its dispatch gap.
    for (int i = blockIdx.x * 256 + tid; i < cp_n; i += gridDim.x * 256) cp_dst[i] = cp_src[i];
    const int t = blockIdx.x * 64 + col;
    const int tt = (blockIdx.x * 64) / kBT, tcol = (blockIdx.x * 64) % kBT + col;
    const size_t ustride = (size_t)(2 + d) * kBT;
    const int u0 = plist_off[tt], u1 = plist_off[tt + 1];
    float M = -3.0e38f, Lsum = 0.0f;
    float acc[kMaxClasses];
#pragma unroll
    for (int k = 0; k < kMaxClasses; ++k) acc[k] = 0.0f;
    // The kernel is latency-bound (101 workgroups, a few KB each): the first kPre partials of a lane are fetched with all their
    // loads in flight at once - slot ids, then every value - instead of one dependent round trip per partial (a tile has ~10
    // partials with the lockstep map, so the loop below normally does not run).
    constexpr int kPre = 3;
    int sl[kPre];
#pragma unroll
    for (int q = 0; q < kPre; ++q) {
        const int u = u0 + g + 4 * q;
        sl[q] = u < u1 ? u : -1;          // slots of a target tile are consecutive (engine.hip get_plan): plist is the identity
    }
    float pm[kPre], pl[kPre], pa[kPre][kMaxClasses];
#pragma unroll
    for (int q = 0; q < kPre; ++q) {
        const float* pu = part + (size_t)(sl[q] < 0 ? 0 : sl[q]) * ustride + tcol;
        const bool on = sl[q] >= 0;
        pm[q] = on ? pu[0] : -3.0e38f;
        pl[q] = on ? pu[kBT] : 0.0f;
#pragma unroll
        for (int k = 0; k < kMaxClasses; ++k) pa[q][k] = (on && k < d) ? pu[(size_t)(2 + k) * kBT] : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < kPre; ++q) {
        if (sl[q] < 0) continue;
        const float Mn = fmaxf(M, pm[q]);
        const float so = __builtin_amdgcn_exp2f((M - Mn) * c), sn = __builtin_amdgcn_exp2f((pm[q] - Mn) * c);
        Lsum = Lsum * so + pl[q] * sn;
#pragma unroll
        for (int k = 0; k < kMaxClasses; ++k)
            if (k < d) acc[k] = acc[k] * so + pa[q][k] * sn;
        M = Mn;
    }
    for (int u = u0 + g + 4 * kPre; u < u1; u += 4) {
        const float* pu = part + (size_t)u * ustride + tcol;
        const float m = pu[0];
        const float Mn = fmaxf(M, m);
        const float so = __builtin_amdgcn_exp2f((M - Mn) * c), sn = __builtin_amdgcn_exp2f((m - Mn) * c);
        Lsum = Lsum * so + pu[kBT] * sn;
#pragma unroll
        for (int k = 0; k < kMaxClasses; ++k)
            if (k < d) acc[k] = acc[k] * so + pu[(size_t)(2 + k) * kBT] * sn;
        M = Mn;
    }
    red[g][0][col] = M;
    red[g][1][col] = Lsum;
#pragma unroll
    for (int k = 0; k < kMaxClasses; ++k)
        if (k < d) red[g][2 + k][col] = acc[k];
    __syncthreads();
    if (g == 0) {
        float Mt = fmaxf(fmaxf(red[0][0][col], red[1][0][col]), fmaxf(red[2][0][col], red[3][0][col]));
        float sc[4], Lt = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sc[q] = __builtin_amdgcn_exp2f((red[q][0][col] - Mt) * c);
            Lt += red[q][1][col] * sc[q];
        }
        // no_l: the propagation kernel ran without denominators (label mode, nobody asked for the prediction): `pred` then holds
        // the un-normalised numerators, whose arg-max is the same
        const float inv = no_l ? 1.0f : 1.0f / Lt;
        int best = 0;
        float bv = -1.0f;
        for (int k = 0; k < d; ++k) {
            const float v = (red[0][2 + k][col] * sc[0] + red[1][2 + k][col] * sc[1] + red[2][2 + k][col] * sc[2] +
                             red[3][2 + k][col] * sc[3]) * inv;
            outv[k][col] = v;
            if (t < HW) pred[(size_t)k * HW + t] = v;
            if (v > bv) { bv = v; best = k; }
        }
        clsv[col] = (uint8_t)best;
        if (t < HW) cls[t] = (uint8_t)best;
    }
    if (!lab_hi && !up.mask) return;
    __syncthreads();
    if (up.mask) {
        // the block's low-res pixels [t0, t1) lie in at most two rows of the map; each row segment owns a rectangle of the mask
        const int t0 = blockIdx.x * 64, t1 = t0 + 64 < HW ? t0 + 64 : HW;
        for (int ry = t0 / up.Wd; ry * up.Wd < t1; ++ry) {
            const int ra = t0 - ry * up.Wd > 0 ? t0 - ry * up.Wd : 0, rb = t1 - ry * up.Wd < up.Wd ? t1 - ry * up.Wd : up.Wd;
            const int xa = up.x0[ra], nx = up.x0[rb] - xa, ya = up.y0[ry], ny = up.y0[ry + 1] - ya;
            for (int i = tid; i < nx * ny; i += 256) {
                const int yy = i / nx, x = xa + i - yy * nx;
                int ix = (int)floorf((float)x * up.sx);
                ix = ix < up.Wd - 1 ? ix : up.Wd - 1;
                up.mask[(size_t)(ya + yy) * up.W + x] = clsv[ry * up.Wd + ix - t0];
            }
        }
    }
    if (!lab_hi) return;
    pack_block_labels(outv, clsv, d, HW, prob, lab_hi, lab_lo);
}

// Label tiles of one 64-pixel block in MFMA A-operand order, from LDS copies of the block's results (256 threads).
__device__ inline void pack_block_labels(const float (*outv)[64], const uint8_t* clsv, int d, int HW, int prob,
                                         bf16_t* __restrict__ lab_hi, bf16_t* __restrict__ lab_lo) {
    const int tid = threadIdx.x;
    const int tl = tid >> 7, s = (tid >> 6) & 1, lane = tid & 63, k = lane & 31;
    const int tile = blockIdx.x * 2 + tl;
    bf16x8 oh, ol;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int pc = tl * 32 + lab_row(s, lane, e);           // column within the block
        const bool ok = blockIdx.x * 64 + pc < HW && k < d;
        float v = 0.0f;
        if (ok) v = prob ? outv[k][pc] : (clsv[pc] == k ? 1.0f : 0.0f);
        const float hi = bf16_round(v);
        oh[e] = (bf16_t)hi;
        ol[e] = (bf16_t)(v - hi);
    }
    const size_t off = ((size_t)tile * 128 + (size_t)s * 64 + lane) * 8;
    *(bf16x8*)(lab_hi + off) = oh;
    if (lab_lo && prob) *(bf16x8*)(lab_lo + off) = ol;
}

// ---------------------------------------------------------------------------------------------------------------
// Top-k variant (NOT in the reference): per target pixel keep the k largest entries of the weighted
// affinity A[.,t] = P[.,t] w[.,t], zero the rest, no renormalisation (k >= N*HW reproduces the dense result).  Ranking by A is
// ranking by the exponent E = S c + log2 w (the softmax max and denominator are column constants).  [r3] Two passes on the dense
// kernel's pipeline (prop_dense.h TK = 1 / 2) with these two kernels between and after them:
//   pass 1   per lane the KS = ceil(k/8)*8 largest GROUP maxima (group = the 16 rows of a reference tile a lane owns), each packed
//            with its (stream index r, half h) in the low `bits` mantissa bits;
//   select2  merges a column's lists: v_k = its k-th largest packed maximum.  With D = |v_k| 2^(bits-22) >= the packing error:
//            the true k-th largest group maximum G_k >= v_k - D, every element of the true top-k is >= G_k (k groups have a
//            maximum >= G_k, each holds an element >= G_k) and lives in a group whose packed maximum is >= v_k - 2D.  So
//            thr_elem = v_k - D bounds the elements, thr_grp = v_k - 2D the groups, and the tiles of the groups that reach
//            thr_grp - at most ~k per column - are marked in the target tile's bitmap;
//   pass 2   re-scores the marked tiles only and dumps, per lane, the 16 exponents of every group that reaches thr_grp;
//   combine2 takes the k largest dumped exponents of a column exactly, sums 2^E per class, arg-maxes, packs the new labels.
// No atomics on values, no counters shared between lanes, no second full scoring pass.
template <int KS>
__device__ __forceinline__ void topk_list_insert(float (&lst)[KS], float x) {
#pragma unroll
    for (int q = KS - 1; q >= 1; --q) lst[q] = __builtin_amdgcn_fmed3f(lst[q - 1], lst[q], x);
    lst[0] = fmaxf(lst[0], x);
}

struct TopkSelectArgs {
    const float* part;        // pass-1 lists: [slot][2 * KS][kBT]
    const int* plist_off;     // slots of target tile tt: plist_off[tt] .. plist_off[tt + 1] - 1 (consecutive, engine.hip get_plan)
    int k, HW, bits, words;   // bits: index bits of a packed maximum; words: bitmap words per target tile
    float c;
    float* thr_grp;           // [TT*256]
    float* thr_elem;          // [TT*256]
    unsigned* bitmap;         // [TT][words], cleared by pass 1
};

// grid = TT * 4, block = 256 = 64 target pixels x 4 lanes (a block's pixels belong to ONE target tile: 256 = 4 x 64)
template <int KS>
__global__ __launch_bounds__(256) void topk_select2_kernel(const TopkSelectArgs a) {
    __shared__ float lsts[3][KS][64];
    __shared__ unsigned bm[2048];      // this block's marks (words <= 2048: NT <= 65 536, checked on the host)
    const int tid = threadIdx.x, col = tid & 63, g = tid >> 6;
    const int t = blockIdx.x * 64 + col;
    const int tt = (blockIdx.x * 64) / kBT, tcol = (blockIdx.x * 64) % kBT + col;
    for (int i = tid; i < a.words; i += 256) bm[i] = 0u;
    const bool live = t < a.HW;
    float lst[KS];
#pragma unroll
    for (int i = 0; i < KS; ++i) lst[i] = kTkDummy;
    if (live) {
        const int u0 = a.plist_off[tt], u1 = a.plist_off[tt + 1];
        const size_t ustride = (size_t)(2 * KS) * kBT;
        for (int u = u0 + g; u < u1; u += 4) {      // this lane's quarter of the slots; a slot's two lists fetched whole
            const float* pu = a.part + (size_t)u * ustride + tcol;
            float v[2][KS];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int i = 0; i < KS; ++i) v[hh][i] = pu[(size_t)(hh * KS + i) * kBT];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                for (int i = 0; i < KS; ++i) {
                    if (v[hh][i] <= lst[KS - 1]) break;      // both lists are descending
                    topk_list_insert<KS>(lst, v[hh][i]);
                }
            }
        }
    }
    if (g > 0) {
#pragma unroll
        for (int i = 0; i < KS; ++i) lsts[g - 1][i][col] = lst[i];
    }
    __syncthreads();
    if (g == 0) {
        float tg = 3.0e38f, te = 3.0e38f;      // dead columns: nothing reaches the threshold
        if (live) {
            for (int gg = 0; gg < 3; ++gg)
                for (int i = 0; i < KS; ++i) {
                    const float x = lsts[gg][i][col];
                    if (x <= lst[KS - 1]) break;
                    topk_list_insert<KS>(lst, x);
                }
            float vk = lst[0];
#pragma unroll
            for (int q = 1; q < KS; ++q)
                if (q < a.k) vk = lst[q];
            const float D = fmaxf(fabsf(vk), 1.0e-30f) * __builtin_amdgcn_exp2f((float)(a.bits - 22));
            te = vk - D;
            tg = vk - 2.0f * D;
            const unsigned imask = (1u << a.bits) - 1u;
#pragma unroll
            for (int q = 0; q < KS; ++q) {
                if (lst[q] >= tg && lst[q] > -1.0e37f) {      // (not a "no group" filler)
                    const unsigned r = (__float_as_uint(lst[q]) & imask) >> 1;
                    atomicOr(&bm[r >> 5], 1u << (r & 31));
                }
            }
        }
        a.thr_grp[t] = tg;
        a.thr_elem[t] = te;
    }
    __syncthreads();
    unsigned* gb = a.bitmap + (size_t)tt * a.words;
    for (int i = tid; i < a.words; i += 256)
        if (bm[i]) atomicOr(&gb[i], bm[i]);
}

struct TopkCombineArgs {
    const float* thr_elem;    // [TT*256]
    const float* dump;        // [TT*256][2][chunks][cap][16]
    const unsigned* dump_r;   // [TT*256][2][chunks][cap]
    const unsigned* cnt;      // [TT*256][2][chunks]
    const uint8_t* cls_ring;  // [ring slots][HWp] class index of every reference pixel
    const float* norm_part;   // dense partials (m, l, ...) of the same step when the PREDICTION is wanted, else nullptr: the result
    const int* plist_off;     //   is then the un-normalised sum (its arg-max is the same)
    int norm_rows;            // rows of one dense partial slot (2 + d)
    int slot[kMaxRef];
    int k, d, HW, HWp, n_ref, chunks, cap;
    float c;
};

// grid = ceil(HW/64), block = 256 = 64 target pixels x 4 lanes; all 256 then pack the block's label tiles.
__global__ __launch_bounds__(256) void topk_combine2_kernel(const TopkCombineArgs a, float* __restrict__ pred,
                                                            uint8_t* __restrict__ cls, bf16_t* __restrict__ lab_hi,
                                                            bf16_t* __restrict__ lab_lo) {
    __shared__ float outv[kMaxClasses][64];
    __shared__ float outg[3][kMaxClasses][64];      // per-class sums of lanes 1..3 of a pixel
    __shared__ float lsts[3][kTopkMax][64];         // sorted candidate lists of lanes 1..3
    __shared__ float red[2][64];                    // [0] = tau, [1] = reference exponent
    __shared__ uint8_t clsv[64];
    const int tid = threadIdx.x, col = tid & 63, g = tid >> 6;
    const int t = blockIdx.x * 64 + col;
    const bool live = t < a.HW;
    if (g == 0) {
        for (int k = 0; k < a.d; ++k) outv[k][col] = 0.0f;
        clsv[col] = 0;
    } else {
        for (int k = 0; k < a.d; ++k) outg[g - 1][k][col] = 0.0f;
    }
    float lst[kTopkMax];
#pragma unroll
    for (int i = 0; i < kTopkMax; ++i) lst[i] = kTkDummy;
    const float floor_e = -1.0e29f * a.c;      // masked rows (S = -1e30) sit below this
    const float te = live ? fmaxf(a.thr_elem[t], floor_e) : 3.0e38f;
    const int n_units = 2 * a.chunks;          // (half, share) pairs of this pixel
    // phase 1: the k largest exponents among the dumped groups (this lane: every 4th group)
    if (live) {
        int gi = 0;
        for (int u = 0; u < n_units; ++u) {
            const size_t ub = (size_t)t * n_units + u;
            unsigned n = a.cnt[ub];
            if (n > (unsigned)a.cap) n = a.cap;
            for (unsigned q = 0; q < n; ++q, ++gi) {
                if ((gi & 3) != g) continue;
                const f32x4* e4 = (const f32x4*)(a.dump + (ub * a.cap + q) * 16);
                const f32x4 v0 = e4[0], v1 = e4[1], v2 = e4[2], v3 = e4[3];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float x = e < 4 ? v0[e & 3] : e < 8 ? v1[e & 3] : e < 12 ? v2[e & 3] : v3[e & 3];
                    if (x >= te && x > lst[kTopkMax - 1]) topk_list_insert<kTopkMax>(lst, x);
                }
            }
        }
    }
    if (g > 0) {
#pragma unroll
        for (int i = 0; i < kTopkMax; ++i) lsts[g - 1][i][col] = lst[i];
    }
    __syncthreads();
    if (g == 0 && live) {
        for (int gg = 0; gg < 3; ++gg)
            for (int i = 0; i < kTopkMax; ++i) {
                const float x = lsts[gg][i][col];
                if (x <= lst[kTopkMax - 1]) break;      // descending
                topk_list_insert<kTopkMax>(lst, x);
            }
        float tau = lst[0];
#pragma unroll
        for (int q = 1; q < kTopkMax; ++q)
            if (q < a.k) tau = lst[q];
        red[0][col] = fmaxf(tau, te);      // fewer than k real elements: everything real is kept
        red[1][col] = lst[0];              // reference exponent of the un-normalised sum: the largest one
    }
    __syncthreads();
    // phase 2: the kept elements, summed per class
    if (live) {
        const float tau = red[0][col];
        float eref = red[1][col], inv = 1.0f;
        if (a.norm_part) {      // the prediction is wanted: the softmax max and denominator of the column, from the dense partials
            const int tt = t / kBT, tcol = t % kBT;
            const int u0 = a.plist_off[tt], u1 = a.plist_off[tt + 1];
            float M = -3.0e38f, L = 0.0f;
            for (int u = u0; u < u1; ++u) {
                const float* pu = a.norm_part + (size_t)u * a.norm_rows * kBT + tcol;
                const float m = pu[0], Mn = fmaxf(M, m);
                L = L * __builtin_amdgcn_exp2f((M - Mn) * a.c) + pu[kBT] * __builtin_amdgcn_exp2f((m - Mn) * a.c);
                M = Mn;
            }
            eref = M * a.c;
            inv = 1.0f / L;
        }
        float (*acc)[64] = g == 0 ? outv : outg[g - 1];
        int gi = 0;
        for (int u = 0; u < n_units; ++u) {
            const size_t ub = (size_t)t * n_units + u;
            const int hh = u / a.chunks;
            unsigned n = a.cnt[ub];
            if (n > (unsigned)a.cap) n = a.cap;
            for (unsigned q = 0; q < n; ++q, ++gi) {
                if ((gi & 3) != g) continue;
                const f32x4* e4 = (const f32x4*)(a.dump + (ub * a.cap + q) * 16);
                const f32x4 v0 = e4[0], v1 = e4[1], v2 = e4[2], v3 = e4[3];
                const unsigned r = a.dump_r[ub * a.cap + q];
                const unsigned pt = r / (unsigned)a.n_ref, fn = r - pt * (unsigned)a.n_ref;
                if (pt >= (unsigned)(a.HWp / kTileR)) continue;      // (the "no group" filler of a segment's first step)
                const uint8_t* cr = a.cls_ring + (size_t)a.slot[fn] * a.HWp + pt * kTileR;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float x = e < 4 ? v0[e & 3] : e < 8 ? v1[e & 3] : e < 12 ? v2[e & 3] : v3[e & 3];
                    if (x >= tau && x > floor_e) {
                        const int kcls = cr[acc_row(e, hh)];
                        if (kcls < a.d) acc[kcls][col] += __builtin_amdgcn_exp2f(x - eref) * inv;
                    }
                }
            }
        }
    }
    __syncthreads();
    if (g == 0 && live) {
        int best = 0;
        float bv = -1.0f;
        for (int k = 0; k < a.d; ++k) {
            const float v = ((outv[k][col] + outg[0][k][col]) + outg[1][k][col]) + outg[2][k][col];
            outv[k][col] = v;
            pred[(size_t)k * a.HW + t] = v;
            if (v > bv) { bv = v; best = k; }
        }
        clsv[col] = (uint8_t)best;
        cls[t] = (uint8_t)best;
    }
    if (!lab_hi) return;
    __syncthreads();
    pack_block_labels(outv, clsv, a.d, a.HW, 0, lab_hi, lab_lo);
}

// Nearest up-sampling of the class map (reference inference_utils.py:74-75; argmax and nearest
// interpolation commute, so the index map is up-sampled instead of the d-channel prediction).
// ATen's nearest source index: min(floor(dst * (float)in/out), in-1).
// sy, sx = (float)in / (float)out computed on the host, as ATen does (device f32 division is not correctly rounded).
__global__ void upsample_kernel(const uint8_t* __restrict__ cls, int Hd, int Wd, uint8_t* __restrict__ mask, int H, int W,
                                float sy, float sx) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W) return;
    int iy = (int)floorf((float)y * sy), ix = (int)floorf((float)x * sx);
    iy = iy < Hd - 1 ? iy : Hd - 1;
    ix = ix < Wd - 1 ? ix : Wd - 1;
    mask[(size_t)y * W + x] = cls[iy * Wd + ix];
}

}  // namespace vosprop
