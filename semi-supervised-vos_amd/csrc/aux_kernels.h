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

// Top-k, between the passes: per target pixel merge the sorted group-maximum lists of all partial slots (2 half-wave
// lists per slot) and take the k-th largest value = lower bound of the k-th largest weighted exponent; also the exact
// column max of the raw scores; zero the candidate counter.  grid = ceil(HWp/64), block = 256 (64 pixels x 4 lanes).
// [r2] The round-1 form walked every list with one DEPENDENT global load per entry (load, compare, break or insert: ~100 round trips
// per pixel, 60 us for 26 blocks).  Now a partial slot's two lists are fetched whole (2 x KS independent loads in flight, the next
// slot id with them), the merged list has KS = ceil(k/8)*8 slots like pass 1's, an insertion is one v_med3 per slot, and four lanes
// share a pixel's slots (merged through LDS).
template <int KS>
__device__ __forceinline__ void topk_select_body(const float* __restrict__ part, const int* __restrict__ plist, int u0, int u1,
                                                 int tcol, int g, float (&lst)[KS], float& M) {
    const size_t ustride = (size_t)(1 + 2 * kTopkMax) * kBT;
#pragma unroll
    for (int i = 0; i < KS; ++i) lst[i] = -3.0e38f;
    M = -3.0e38f;
    for (int u = u0 + g; u < u1; u += 4) {       // this lane's quarter of the partial slots (slots of a target tile are consecutive,
                                                 // engine.hip get_plan: the slot list is the identity and is not read)
        const float* pu = part + (size_t)u * ustride + tcol;
        float v[2][KS];
        const float m = pu[0];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int i = 0; i < KS; ++i) v[hh][i] = pu[(size_t)(1 + hh * kTopkMax + i) * kBT];
        M = fmaxf(M, m);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
            for (int i = 0; i < KS; ++i) {
                const float x = v[hh][i];
                if (x <= lst[KS - 1]) break;   // both lists are descending
#pragma unroll
                for (int q = KS - 1; q >= 1; --q) lst[q] = __builtin_amdgcn_fmed3f(lst[q - 1], lst[q], x);
                lst[0] = fmaxf(lst[0], x);
            }
        }
    }
}

// block = 256 = 64 target pixels x 4 lanes: every lane merges a quarter of the pixel's partial slots, lane 0 merges the four lists
template <int KS>
__device__ __forceinline__ void topk_select_pixel(const float* __restrict__ part, const int* __restrict__ plist_off,
                                                  const int* __restrict__ plist, int k, int t, bool live, int col, int g,
                                                  float (*lsts)[kTopkMax][64], float (*mred)[64], float* __restrict__ thr,
                                                  float* __restrict__ mfin) {
    float lst[KS], M = -3.0e38f;
#pragma unroll
    for (int i = 0; i < KS; ++i) lst[i] = -3.0e38f;
    if (live) {
        const int tt = t / kBT, tcol = t % kBT;
        topk_select_body<KS>(part, plist, plist_off[tt], plist_off[tt + 1], tcol, g, lst, M);
    }
    mred[g][col] = M;
    if (g > 0) {
#pragma unroll
        for (int i = 0; i < KS; ++i) lsts[g - 1][i][col] = lst[i];
    }
    __syncthreads();
    if (g == 0 && live) {
        for (int gg = 0; gg < 3; ++gg)
            for (int i = 0; i < KS; ++i) {
                const float x = lsts[gg][i][col];
                if (x <= lst[KS - 1]) break;
#pragma unroll
                for (int q = KS - 1; q >= 1; --q) lst[q] = __builtin_amdgcn_fmed3f(lst[q - 1], lst[q], x);
                lst[0] = fmaxf(lst[0], x);
            }
        float vk = lst[0];
#pragma unroll
        for (int q = 1; q < KS; ++q)
            if (q < k) vk = lst[q];
        thr[t] = vk;
        mfin[t] = fmaxf(fmaxf(mred[0][col], mred[1][col]), fmaxf(mred[2][col], mred[3][col]));
    }
}

__global__ __launch_bounds__(256) void topk_select_kernel(const float* __restrict__ part, const int* __restrict__ plist_off,
                                                          const int* __restrict__ plist, int k, int HW, int HWp,
                                                          float* __restrict__ thr, float* __restrict__ mfin,
                                                          unsigned* __restrict__ cnt) {
    __shared__ float lsts[3][kTopkMax][64];
    __shared__ float mred[4][64];
    const int col = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + col;
    if (g == 0 && t < HWp) {
        cnt[t] = 0;
        if (t >= HW) { thr[t] = 3.0e38f; mfin[t] = 0.0f; }
    }
    const bool live = t < HW;
    if (k <= 8) topk_select_pixel<8>(part, plist_off, plist, k, t, live, col, g, lsts, mred, thr, mfin);
    else if (k <= 16) topk_select_pixel<16>(part, plist_off, plist, k, t, live, col, g, lsts, mred, thr, mfin);
    else if (k <= 24) topk_select_pixel<24>(part, plist_off, plist, k, t, live, col, g, lsts, mred, thr, mfin);
    else topk_select_pixel<kTopkMax>(part, plist_off, plist, k, t, live, col, g, lsts, mred, thr, mfin);
}

struct TopkCombineArgs {
    const float* part;       // pass-2 partials: rows (m, l)
    const int* plist_off;
    const int* plist;
    const float* thr;
    const float* mfin;
    const unsigned* cnt;
    const uint2* cand;
    const uint8_t* cls_ring; // [cap][HWp] class index of every reference pixel
    int slot[kMaxRef];
    int k, d, HW, HWp;
    float c;
};

// Top-k, after pass 2: denominators from the partials, the k largest candidates of each target pixel summed per class.
// grid = ceil(HW/64), block = 256 = 64 target pixels x 4 lanes; all 256 then pack the block's label tiles.
// [r2] Round 1 gave a pixel ONE thread: ~10 dependent partial loads, then a load per candidate, then a gather per kept candidate
// (36 us).  Now the four lanes of a pixel take every fourth partial / candidate with their loads in flight four at a time, keep
// private sorted lists (one v_med3 per slot), and lane 0 merges the four lists and the four per-class sums through LDS in a fixed
// order.
template <int KS>
__device__ __forceinline__ void topk_list_insert(float (&lst)[KS], float x) {
#pragma unroll
    for (int q = KS - 1; q >= 1; --q) lst[q] = __builtin_amdgcn_fmed3f(lst[q - 1], lst[q], x);
    lst[0] = fmaxf(lst[0], x);
}

__global__ __launch_bounds__(256) void topk_combine_kernel(const TopkCombineArgs a, float* __restrict__ pred,
                                                           uint8_t* __restrict__ cls, bf16_t* __restrict__ lab_hi,
                                                           bf16_t* __restrict__ lab_lo) {
    __shared__ float outv[kMaxClasses][64];
    __shared__ float outg[3][kMaxClasses][64];      // per-class sums of lanes 1..3 of a pixel
    __shared__ float lsts[3][kTopkMax][64];         // sorted candidate lists of lanes 1..3
    __shared__ float lred[4][64];                   // denominator parts, then [0] = tau
    __shared__ uint8_t clsv[64];
    const int tid = threadIdx.x, col = tid & 63, g = tid >> 6;
    const int t = blockIdx.x * 64 + col;
    const bool live = t < a.HW;
    if (g == 0) {
        for (int k = 0; k < a.d; ++k) outv[k][col] = 0.0f;
        clsv[col] = 0;
    }
    unsigned n = 0;
    const uint2* cd = a.cand + (size_t)(live ? t : 0) * kTopkCap;
    float lst[kTopkMax];
#pragma unroll
    for (int i = 0; i < kTopkMax; ++i) lst[i] = -3.0e38f;
    float L = 0.0f;
    if (live) {
        const int tt = t / kBT, tcol = t % kBT;
        const int u0 = a.plist_off[tt], u1 = a.plist_off[tt + 1];
        n = a.cnt[t];
        if (n > (unsigned)kTopkCap) n = kTopkCap;
        // this lane's quarter of the denominators: slot ids first, then the values, all in flight together
        int sl[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) sl[q] = u0 + g + 4 * q < u1 ? u0 + g + 4 * q : -1;     // (identity slot list)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (sl[q] >= 0) L += a.part[((size_t)sl[q] * 2 + 1) * kBT + tcol];
        for (int u = u0 + g + 16; u < u1; u += 4) L += a.part[((size_t)u * 2 + 1) * kBT + tcol];
        // this lane's quarter of the candidates, four loads in flight
        for (unsigned i = g; i < n; i += 16) {
            float x[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) x[q] = i + 4 * q < n ? __uint_as_float(cd[i + 4 * q].x) : -3.0e38f;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (x[q] > lst[kTopkMax - 1]) topk_list_insert<kTopkMax>(lst, x[q]);
        }
    }
    lred[g][col] = L;
    if (g > 0) {
#pragma unroll
        for (int i = 0; i < kTopkMax; ++i) lsts[g - 1][i][col] = lst[i];
        for (int k = 0; k < a.d; ++k) outg[g - 1][k][col] = 0.0f;
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
        const float Lt = ((lred[0][col] + lred[1][col]) + lred[2][col]) + lred[3][col];
        lred[0][col] = tau;
        lred[1][col] = 1.0f / Lt;
    }
    __syncthreads();
    if (live) {
        const float tau = lred[0][col], inv = lred[1][col];
        const float mc = a.mfin[t] * a.c;
        float (*acc)[64] = g == 0 ? outv : outg[g - 1];
        for (unsigned i = g; i < n; i += 4) {
            const uint2 e = cd[i];
            const float E = __uint_as_float(e.x);
            if (E < tau) continue;
            const unsigned fn = e.y / (unsigned)a.HWp, px = e.y - fn * (unsigned)a.HWp;
            const int kcls = a.cls_ring[(size_t)a.slot[fn] * a.HWp + px];
            if (kcls < a.d) acc[kcls][col] += __builtin_amdgcn_exp2f(E - mc) * inv;
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
