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

// The same labels as ONE v_mfma_f32_16x16x32_bf16 A fragment per tile (prop_mask.h; one-hot labels of <= 16 classes): element j of
// lane l is L[class = l & 15][row], row = 4 (l >> 4) + j for j < 4 and 16 + 4 (l >> 4) + (j - 4) above - the rows whose weights the
// lane with k block l >> 4 holds in its two row-block accumulators.  [tile][64 lanes][8] bf16 = 1 KiB per tile.
__device__ inline int lab16_row(int lane, int j) { return (j < 4 ? 0 : 12) + 4 * (lane >> 4) + j; }

// One-hot labels from a class-index map (reference index_to_onehot, src/utils/utils.py:59-68).
// One thread per 16-byte chunk: grid*block >= tiles*128.
__global__ void pack_cls_kernel(const uint8_t* __restrict__ cls, bf16_t* __restrict__ lab_hi, bf16_t* __restrict__ lab16, int HW,
                                int tiles) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= tiles * 128) return;
    if (lab16 && gid < tiles * 64) {
        const int tile16 = gid >> 6, lane16 = gid & 63;
        bf16x8 o16;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int p = tile16 * kTileR + lab16_row(lane16, e);
            o16[e] = (bf16_t)((p < HW && cls[p] == (lane16 & 15)) ? 1.0f : 0.0f);
        }
        *(bf16x8*)(lab16 + (size_t)gid * 8) = o16;
    }
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
                                         bf16_t* __restrict__ lab_hi, bf16_t* __restrict__ lab_lo, bf16_t* __restrict__ lab16 = nullptr);

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
                                                      uint4* __restrict__ cp_dst, int cp_n, int no_l, int cp_f16,
                                                      bf16_t* __restrict__ lab16) {
    __shared__ float red[4][kMaxClasses + 2][64];
    __shared__ float outv[kMaxClasses][64];
    __shared__ uint8_t clsv[64];
    const int tid = threadIdx.x, col = tid & 63, g = tid >> 6;
    // Optional rider: the ring copy of the frame just propagated (cp_n 16-byte units of channels-last bf16 features, caller's buffer
    // -> ring slot; vosprop_step).  The propagation kernel read the target from the caller's buffer; the slot is first needed as a
    // REFERENCE by the next step.  This kernel is latency-bound (three dependent round trips), eight independent 16-byte copies per
    // thread in front of them cost nothing and save the separate copy launch and its dispatch gap.
    // [r3] cp_f16: the caller's features are f16 - converted on the way (what push_hwc_kernel would have done up front)
    for (int i = blockIdx.x * 256 + tid; i < cp_n; i += gridDim.x * 256) {
        uint4 v = cp_src[i];
        if (cp_f16) {
            typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
            const f16x8 hv = __builtin_bit_cast(f16x8, v);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(float)hv[e];
            v = __builtin_bit_cast(uint4, o);
        }
        cp_dst[i] = v;
    }
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
    pack_block_labels(outv, clsv, d, HW, prob, lab_hi, lab_lo, lab16);
}

// Label tiles of one 64-pixel block in MFMA A-operand order, from LDS copies of the block's results (256 threads).
__device__ inline void pack_block_labels(const float (*outv)[64], const uint8_t* clsv, int d, int HW, int prob,
                                         bf16_t* __restrict__ lab_hi, bf16_t* __restrict__ lab_lo, bf16_t* __restrict__ lab16) {
    const int tid = threadIdx.x;
    if (lab16 && !prob && tid < 128) {      // the two tiles of this 64-pixel block in the 16x16x32 fragment order (lab16_row)
        const int tl16 = tid >> 6, lane16 = tid & 63;
        bf16x8 o16;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int pc = tl16 * 32 + lab16_row(lane16, e);
            o16[e] = (bf16_t)((blockIdx.x * 64 + pc < HW && clsv[pc] == (lane16 & 15)) ? 1.0f : 0.0f);
        }
        *(bf16x8*)(lab16 + ((size_t)(blockIdx.x * 2 + tl16) * 64 + lane16) * 8) = o16;
    }
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
// ---- k-th largest of up to 64 * NV values spread over a wave, by radix selection on sortable keys ----
// key(x) orders like x (unsigned compare); 0 = "no value".  Bit by bit from the top: keep the bit if at least k keys reach the
// candidate; a count is NV ballots + scalar popcounts (no LDS, no shuffles).  Stops as soon as EXACTLY k keys reach the candidate
// (then those k are the k largest, whatever the lower bits).  Returns T with: {key >= T, key != 0} = the k largest, ties of the k-th
// included (what the oracle's `S >= kth` keeps); fewer than k values: T = 0 (all of them).
__device__ __forceinline__ unsigned sortable_key(float x) {
    const unsigned u = __float_as_uint(x);
    return u ^ ((unsigned)((int)u >> 31) | 0x80000000u);
}
__device__ __forceinline__ float key_value(unsigned k) {
    return __uint_as_float(k ^ ((k & 0x80000000u) ? 0x80000000u : 0xFFFFFFFFu));
}
template <int NV>
__device__ __forceinline__ unsigned wave_kth_largest(const unsigned (&key)[NV], int nv, int k) {
    unsigned T = 0u;
    for (int b = 31; b >= 0; --b) {
        const unsigned cand = T | (1u << b);
        int c = 0;
#pragma unroll
        for (int j = 0; j < NV; ++j)
            if (j < nv) c += __builtin_popcountll(__ballot(key[j] >= cand));
        if (c >= k) {
            T = cand;
            if (c == k) break;
        }
    }
    return T;
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned w = (unsigned)__shfl_xor((int)v, o);
        v = w < v ? w : v;
    }
    return v;
}
__device__ __forceinline__ float wave_sum_f32(float v) {      // fixed order: the same bits on every run
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

struct TopkSelectArgs {
    const float* part;        // pass-1 lists: [target tile][kBT columns][slot rank][2 * KS]: a target pixel's entries are contiguous
    const int* plist_off;     // slots of target tile tt: plist_off[tt] .. plist_off[tt + 1] - 1 (consecutive, engine.hip get_plan)
    int k, ks, HW, bits, words;   // ks: list slots per lane; bits: index bits of a packed maximum; words: bitmap words per target tile
    float* thr_grp;           // [TT*256]
    float* thr_elem;          // [TT*256]
    unsigned* bitmap;         // [TT][words], cleared by pass 1
    unsigned* over;           // [3] clamp counters (common.h PropArgs.tk_over)
};

constexpr int kTkSelCols = 8;      // columns (= waves) per block of the select kernel
constexpr int kTkSelCap = 512;     // candidate keys a column can carry through LDS (8 per lane)
constexpr int kTkSelRegs = 24;     // sweep steps of a column kept in registers (24 x 64 = 1 536 list entries)

// k-th largest of ONE key per lane (the common case after compaction): 2-3 instructions per bit
__device__ __forceinline__ unsigned wave_kth_largest_1(unsigned key, int k) {
    unsigned T = 0u;
    for (int b = 31; b >= 0; --b) {
        const unsigned cand = T | (1u << b);
        const int c = __builtin_popcountll(__ballot(key >= cand));
        if (c >= k) {
            T = cand;
            if (c == k) break;
        }
    }
    return T;
}

// One WAVE per target pixel.  The column's (slots x 2 x KS) packed group maxima are STREAMED, 64 at a time (consecutive lanes read
// consecutive floats); nothing about their number is assumed:
//   sweep 1  every lane keeps the maximum of the values it sees; the k-th largest of the 64 lane maxima, L0, is a lower bound of
//            v_k (k lanes hold a value >= L0);
//   sweep 2  the values >= L0 - 2 D(L0) (a superset of everything that can reach thr_grp = v_k - 2 D(v_k), see above) are compacted
//            into LDS with ballot prefixes - normally a few dozen;
//   then     v_k = their k-th largest by radix selection (8 keys per lane at most), and the lanes mark the tiles of the candidates
//            that reach thr_grp.
// Should more than kTkSelCap values pass sweep 2 (lists that leave most lanes without a real value), L0 is replaced by the exact
// v_k from a streamed radix selection (32 sweeps; correct, slow, not seen on the bench shapes).
// grid = TT * 256 / 8, block = 8 waves (the block's pixels belong to ONE target tile).
__global__ __launch_bounds__(kTkSelCols * 64) void topk_select2_kernel(const TopkSelectArgs a) {
    __shared__ unsigned bm[2048];      // this block's marks (words <= 2048: NT <= 65 536, checked on the host)
    __shared__ unsigned carry[kTkSelCols][kTkSelCap];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int t = blockIdx.x * kTkSelCols + wv;
    const int tt = (blockIdx.x * kTkSelCols) / kBT, tcol = (blockIdx.x * kTkSelCols) % kBT + wv;
    for (int i = tid; i < a.words; i += kTkSelCols * 64) bm[i] = 0u;
    __syncthreads();
    float tg = 3.0e38f, te = 3.0e38f;      // dead columns: nothing reaches the threshold
    if (t < a.HW) {                        // (wave-uniform)
        const int u0 = a.plist_off[tt], u1 = a.plist_off[tt + 1];
        const int two_ks = 2 * a.ks;
        const int nvals = (u1 - u0) * two_ks;
        const unsigned kfloor = sortable_key(-1.0e37f);      // "no group" fillers are not values
        const float dscale = __builtin_amdgcn_exp2f((float)(a.bits - 22));
        // the column's nvals list entries are contiguous: sweep step i reads entries 64 i .. 64 i + 63, one per lane
        const float* colp = a.part + ((size_t)u0 * kBT + (size_t)tcol * (u1 - u0)) * two_ks;
        int c_v = lane;
        auto rewind = [&]() { c_v = lane; };
        auto next_key = [&]() -> unsigned {      // the cursor's value (0 past the end or for a filler), then advance.  The load
            const bool in = c_v < nvals;         // itself is unconditional (clamped address): a run of them goes out back to back
            unsigned kk = sortable_key(colp[in ? c_v : 0]);
            kk = in && kk > kfloor ? kk : 0u;
            c_v += 64;
            return kk;
        };
        // the first kTkSelRegs sweep steps (64 values each) stay in registers - all of a column at the bench shapes - so sweep 2
        // re-reads nothing; longer columns stream the rest
        const int rounds = (nvals + 63) >> 6;
        unsigned kreg[kTkSelRegs];
#pragma unroll
        for (int i = 0; i < kTkSelRegs; ++i) kreg[i] = next_key();      // (past the end: 0)
        unsigned mx = 0u;
#pragma unroll
        for (int i = 0; i < kTkSelRegs; ++i) mx = kreg[i] > mx ? kreg[i] : mx;
        for (int i = kTkSelRegs; i < rounds; ++i) {
            const unsigned kk = next_key();
            mx = kk > mx ? kk : mx;
        }
        unsigned L0 = wave_kth_largest_1(mx, a.k);
        if (L0) L0 = wave_min_u32(mx >= L0 && mx ? mx : 0xFFFFFFFFu);      // the k-th largest lane maximum itself
        int n_c = 0;
        for (int attempt = 0; attempt < 2; ++attempt) {
            unsigned Lc = 0u;
            if (L0) {
                const float l0 = key_value(L0);
                Lc = sortable_key(l0 - 2.0f * fmaxf(fabsf(l0), 1.0e-30f) * dscale);
            }
            int base = 0;
            auto compact = [&](unsigned kk) {
                const bool keep = kk != 0u && kk >= Lc;
                const unsigned long long m = __ballot(keep);
                if (m) {      // (wave-uniform: most sweep steps carry nothing over)
                    const int pos = base + __builtin_popcountll(m & ((1ull << lane) - 1ull));
                    if (keep && pos < kTkSelCap) carry[wv][pos] = kk;
                    base += __builtin_popcountll(m);
                }
            };
#pragma unroll
            for (int i = 0; i < kTkSelRegs; ++i) compact(kreg[i]);
            if (rounds > kTkSelRegs) {
                rewind();
                for (int i = 0; i < rounds; ++i) {
                    const unsigned kk = next_key();
                    if (i >= kTkSelRegs) compact(kk);
                }
            }
            n_c = base;
            if (n_c <= kTkSelCap || attempt == 1) break;
            // too many: the exact k-th largest by a streamed radix selection becomes the bound
            unsigned T = 0u;
            for (int b = 31; b >= 0; --b) {
                const unsigned cand = T | (1u << b);
                int c = 0;
                rewind();
                for (int v0 = 0; v0 < nvals; v0 += 64) c += __builtin_popcountll(__ballot(next_key() >= cand));
                if (c >= a.k) T = cand;
            }
            L0 = T;
        }
        if (n_c > kTkSelCap && lane == 0) atomicAdd(a.over + 2, 1u);      // candidates beyond the LDS carry are dropped: reported
        n_c = n_c < kTkSelCap ? n_c : kTkSelCap;
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the wave's own LDS writes have landed
        __builtin_amdgcn_wave_barrier();
        constexpr int NVc = kTkSelCap / 64;
        unsigned key[NVc];
#pragma unroll
        for (int j = 0; j < NVc; ++j) key[j] = 64 * j + lane < n_c ? carry[wv][64 * j + lane] : 0u;
        const int nvc = (n_c + 63) >> 6;
        const unsigned T = nvc <= 1 ? wave_kth_largest_1(key[0], a.k) : wave_kth_largest<NVc>(key, nvc, a.k);
        unsigned mine = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < NVc; ++j)
            if (key[j] >= T && key[j] != 0u) mine = key[j] < mine ? key[j] : mine;
        mine = wave_min_u32(mine);
        // v_k: the k-th largest packed maximum (the smallest one kept); fewer than k real groups: everything real is a candidate
        const float vk = (mine == 0xFFFFFFFFu || n_c < a.k) ? kTkDummy : key_value(mine);
        const float D = fmaxf(fabsf(vk), 1.0e-30f) * dscale;
        te = vk - D;
        tg = vk - 2.0f * D;
        const unsigned kg = sortable_key(tg), imask = (1u << a.bits) - 1u;
#pragma unroll
        for (int j = 0; j < NVc; ++j) {
            if (key[j] != 0u && key[j] >= kg) {
                const unsigned r = (__float_as_uint(key_value(key[j])) & imask) >> 1;
                atomicOr(&bm[r >> 5], 1u << (r & 31));
            }
        }
    }
    if (lane == 0) {
        a.thr_grp[t] = tg;
        a.thr_elem[t] = te;
    }
    __syncthreads();
    unsigned* gb = a.bitmap + (size_t)tt * a.words;
    for (int i = tid; i < a.words; i += kTkSelCols * 64)
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
    unsigned* over;           // [3] clamp counters (common.h PropArgs.tk_over)
    int force_radix;          // test switch (VOSPROP_TK_FORCE_RADIX): 1 = every column takes the overflow fallback (radix selection on all keys)
};

constexpr int kTkComNV = 10;      // groups per quarter-wave the combine kernel holds: 40 dumped groups (640 exponents) per pixel
constexpr int kTkComWaves = 16;   // waves = pixels per block: half a 32-pixel label tile
constexpr int kTkComCap = 128;    // candidate keys a pixel carries through LDS (2 per lane)

// One WAVE per target pixel: lane = (group g % 4, element e) holds the exponents of up to 40 dumped groups (normally ~20).  The k
// largest: lane maxima give a lower bound L0 of the k-th largest (k lanes hold a value >= L0), the few values >= L0 are compacted
// through LDS and radix-selected (2 keys per lane); the classes of the kept elements come from the class-index ring, the per-class
// sums from fixed-order wave reductions (reproducible).  grid = ceil(HW / 16), block = 16 waves; 64 threads then pack the block's
// half of its label tile.  [r3: the first form - four lanes per pixel walking the groups with sorted insertion - took 57 us on the
// bench clip and 159 us on flat logits, more than pass 2 itself]
__global__ __launch_bounds__(kTkComWaves * 64) void topk_combine2_kernel(const TopkCombineArgs a, float* __restrict__ pred,
                                                                          uint8_t* __restrict__ cls, bf16_t* __restrict__ lab_hi,
                                                                          bf16_t* __restrict__ lab_lo) {
    __shared__ unsigned tab[kTkComWaves][4 * kTkComNV];      // group g of the wave's pixel -> (unit << 8 | index in the unit)
    __shared__ unsigned carry[kTkComWaves][kTkComCap];
    __shared__ uint8_t clsv[kTkComWaves];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int e = lane & 15, gq = lane >> 4;
    const float floor_e = -1.0e29f * a.c;      // masked rows (S = -1e30) sit below this
    const int n_units = 2 * a.chunks;          // (half, share) pairs of a pixel: <= 64 (host)
    const int t = blockIdx.x * kTkComWaves + wv;
    if (lane == 0) clsv[wv] = 0;
    if (t < a.HW) {      // (wave-uniform)
        // ---- the pixel's dumped groups: counts per unit, exclusive prefix, table group -> (unit, index) ----
        int cu = 0;
        if (lane < n_units) {
            cu = (int)a.cnt[(size_t)t * n_units + lane];
            cu = cu < a.cap ? cu : a.cap;
        }
        int incl = cu;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o);
            if (lane >= o) incl += u;
        }
        const int before = incl - cu;
        int total = __shfl(incl, 63);
        if (total > 4 * kTkComNV && lane == 0) atomicAdd(a.over + 1, 1u);      // groups beyond 40 are dropped: reported
        total = total < 4 * kTkComNV ? total : 4 * kTkComNV;      // (more than 40 groups reach the threshold only under mass ties)
        for (int q = 0; q < cu; ++q)
            if (before + q < 4 * kTkComNV) tab[wv][before + q] = ((unsigned)lane << 8) | (unsigned)q;
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): this wave's own LDS writes have landed (one wave: no barrier needed)
        __builtin_amdgcn_wave_barrier();
        const float te = fmaxf(a.thr_elem[t], floor_e);
        unsigned key[kTkComNV], meta[kTkComNV];
        float x[kTkComNV];
        const int nvg = (total + 3) >> 2;
        unsigned mx = 0u;
#pragma unroll
        for (int j = 0; j < kTkComNV; ++j) {
            key[j] = 0u;
            meta[j] = 0u;
            x[j] = 0.0f;
            const int g = 4 * j + gq;
            if (g < total) {
                const unsigned m = tab[wv][g];
                const size_t gi = ((size_t)t * n_units + (m >> 8)) * a.cap + (m & 0xFFu);
                x[j] = a.dump[gi * 16 + e];
                meta[j] = (a.dump_r[gi] << 1) | ((m >> 8) >= (unsigned)a.chunks ? 1u : 0u);      // (frame << 16 | pixel tile, half)
                if (x[j] >= te) key[j] = sortable_key(x[j]);
            }
            mx = key[j] > mx ? key[j] : mx;
        }
        // ---- the k-th largest exponent ----
        unsigned L0 = wave_kth_largest_1(mx, a.k);
        if (L0) L0 = wave_min_u32(mx >= L0 && mx ? mx : 0xFFFFFFFFu);      // the k-th largest lane maximum: a lower bound
        int base = 0;
#pragma unroll
        for (int j = 0; j < kTkComNV; ++j) {
            const bool keep = key[j] != 0u && key[j] >= L0;
            const unsigned long long m = __ballot(keep);
            const int pos = base + __builtin_popcountll(m & ((1ull << lane) - 1ull));
            if (keep && pos < kTkComCap) carry[wv][pos] = key[j];
            base += __builtin_popcountll(m);
        }
        unsigned T;
        if (base <= kTkComCap && !(a.force_radix & 1)) {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            unsigned k2[2];
            k2[0] = lane < base ? carry[wv][lane] : 0u;
            k2[1] = 64 + lane < base ? carry[wv][64 + lane] : 0u;
            T = base <= 64 ? wave_kth_largest_1(k2[0], a.k) : wave_kth_largest<2>(k2, 2, a.k);
            T = T > L0 ? T : L0;      // (an early exit leaves T a prefix that separates the k largest CANDIDATES only: below L0 it says
                                      //  nothing about the keys that were not carried over)
        } else {
            T = wave_kth_largest<kTkComNV>(key, nvg, a.k);
        }
        // ---- reference exponent and denominator ----
        float eref, inv = 1.0f;
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
        } else {                // un-normalised: against the largest exponent
            const unsigned top = ~wave_min_u32(~mx);
            eref = top ? key_value(top) : 0.0f;
        }
        // ---- the kept elements: class from the class-index ring, weight 2^(E - eref) ----
        float w[kTkComNV];
        int kc[kTkComNV];
        const unsigned tiles = (unsigned)(a.HWp / kTileR);
#pragma unroll
        for (int j = 0; j < kTkComNV; ++j) {
            w[j] = 0.0f;
            kc[j] = -1;
            if (key[j] != 0u && key[j] >= T) {
                const unsigned r = meta[j] >> 1, hh = meta[j] & 1u;
                const unsigned pt = r & 0xFFFFu, fn = (r >> 16) & 0x3Fu;
                if (pt < tiles && fn < (unsigned)a.n_ref) {
                    kc[j] = a.cls_ring[(size_t)a.slot[fn] * a.HWp + pt * kTileR + acc_row(e, (int)hh)];
                    w[j] = __builtin_amdgcn_exp2f(x[j] - eref) * inv;
                }
            }
        }
        int best = 0;
        float bv = -1.0f;
        for (int k = 0; k < a.d; ++k) {
            float sk = 0.0f;
#pragma unroll
            for (int j = 0; j < kTkComNV; ++j) sk += kc[j] == k ? w[j] : 0.0f;
            sk = wave_sum_f32(sk);
            if (lane == 0) pred[(size_t)k * a.HW + t] = sk;
            if (sk > bv) { bv = sk; best = k; }
        }
        if (lane == 0) {
            clsv[wv] = (uint8_t)best;
            cls[t] = (uint8_t)best;
        }
    }
    if (!lab_hi) return;
    __syncthreads();
    if (tid < 64) {      // the block's half (rows 16 s .. 16 s + 15) of its label tile, MFMA A-operand order (one-hot: label mode only)
        const int s2 = blockIdx.x & 1, k = tid & 31;
        bf16x8 oh;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int pc = lab_row(s2, tid, q) - 16 * s2;      // pixel within the block
            const bool on = blockIdx.x * kTkComWaves + pc < a.HW && k < a.d && clsv[pc] == k;
            oh[q] = (bf16_t)(on ? 1.0f : 0.0f);
        }
        *(bf16x8*)(lab_hi + ((size_t)(blockIdx.x >> 1) * 128 + (size_t)s2 * 64 + tid) * 8) = oh;
    }
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
