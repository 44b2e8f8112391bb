/*
 * CPU ORACLE (plain C restatement) - TEST INFRASTRUCTURE ONLY.
 *
 * A from-scratch C restatement of the reference's label-propagation hot path, independent of
 * torch/BLAS, used by tests/ (as the checker), by __graft_entry__.smoke() and by bench.py's
 * cpu_baseline leg.  The product path (semi-supervised-vos_amd/) never links or calls this.
 *
 * Parity status: PINNED - tests/test_oracle_golden.py checks every entry point against
 * tests/golden/reference_goldens.npz (outputs of the reference's own Python run in the build
 * container) to <= 2e-5 relative (f32 summation order differs from BLAS; integers are exact).
 *
 * file:line citations are into the reference tree (hynekdav/semi-supervised-VOS).
 *
 * Build: see oracle/Makefile (gcc -O3 -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define VOS_CONTINUOUS_FRAME 4 /* src/config.py:13 */

/* src/model/predict.py:74-89.  out must hold num_refs ints; returns the number written.
 * np.linspace(a, b, n) is  a + k*step, step=(b-a)/(n-1) in float64 with the last element forced to b;
 * .astype(int) truncates toward zero. */
int vos_oracle_sample_frames(int frame_idx, int take_range, int num_refs, int* out) {
    int n = 0;
    if (frame_idx <= num_refs) {
        for (int i = 0; i < frame_idx; ++i) out[n++] = i;
        return n;
    }
    const int dense_num = VOS_CONTINUOUS_FRAME - 1;
    const int sparse_num = num_refs - dense_num;
    const int ref_end = frame_idx - dense_num - 1;
    int ref_start = ref_end - take_range;
    if (ref_start < 0) ref_start = 0;
    if (sparse_num == 1) {
        out[n++] = ref_start;
    } else if (sparse_num > 1) {
        const double step = ((double)ref_end - (double)ref_start) / (double)(sparse_num - 1);
        for (int k = 0; k < sparse_num; ++k) {
            double v = (k == sparse_num - 1) ? (double)ref_end : (double)ref_start + (double)k * step;
            out[n++] = (int)v;
        }
    }
    for (int j = 0; j < dense_num; ++j) out[n++] = frame_idx - dense_num + j;
    return n;
}

/* One entry of get_spatial_weight, src/model/predict.py:158-175, in the reference's own f32 steps:
 * coords = (idx / float(W), idx % W) as f32; diff; pow(2); sum; -d / sigma**2; exp. */
static inline float spatial_weight_entry(int i, int j, int W, float sigma_sq) {
    const float ui = (float)i / (float)W, uj = (float)j / (float)W;
    const float vi = (float)(i % W), vj = (float)(j % W);
    const float du = uj - ui, dv = vj - vi;
    const float d2 = du * du + dv * dv;
    return expf(-d2 / sigma_sq);
}

void vos_oracle_spatial_weight(int H, int W, float sigma, float* out) {
    const int HW = H * W;
    const float s2 = (float)((double)sigma * (double)sigma);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < HW; ++i)
        for (int j = 0; j < HW; ++j) out[(size_t)i * HW + j] = spatial_weight_entry(i, j, W, s2);
}

/* F.interpolate(mode='nearest') source index: floor(dst * (in/out)) with an f32 scale
 * (predict.py:94, inference_utils.py:74). */
static inline int nearest_src(int dst, int in_size, int out_size) {
    const float scale = (float)in_size / (float)out_size;
    int s = (int)floorf((float)dst * scale);
    return s < in_size - 1 ? s : in_size - 1;
}

/* get_labels, src/model/predict.py:92-96 (+ index_to_onehot, src/utils/utils.py:59-68):
 * class-index image (H,W) -> one-hot (d, Hd*Wd) f32, nearest down-sampled. */
void vos_oracle_get_labels(const uint8_t* label, int H, int W, int Hd, int Wd, int d, float* onehot) {
    memset(onehot, 0, sizeof(float) * (size_t)d * Hd * Wd);
    for (int y = 0; y < Hd; ++y) {
        const int sy = nearest_src(y, H, Hd);
        for (int x = 0; x < Wd; ++x) {
            const int sx = nearest_src(x, W, Wd);
            const int k = label[(size_t)sy * W + sx];
            if (k < d) onehot[(size_t)k * Hd * Wd + (size_t)y * Wd + x] = 1.0f;
        }
    }
}

/* predict, src/model/predict.py:19-71.
 *   ref        (T, C, HW)  f32, NCHW history (frame-major, channel-major inside a frame)
 *   target     (C, HW)     f32
 *   ref_label  (d, T, HW)  f32 (one-hot or probabilities)
 *   out        (d, HW)     f32
 * Steps: sample frames (:41-43); S = R.T (:46-49); S *= temperature (:52); column softmax over ALL
 * N*HW rows (:55); post-softmax spatial prior unless probability mode (:58-66: sigma2 for the
 * first N-4 sampled frames when frame_idx > 15, sigma1 otherwise); out = L . A (:70), not renormalised.
 * Returns 0, or -1 on allocation failure / bad arguments. */
int vos_oracle_predict(const float* ref, const float* target, const float* ref_label, int T, int C, int Hd,
                       int Wd, int d, int frame_idx, int take_range, int ref_num, float temperature,
                       float sigma1, float sigma2, int probability_propagation, float* out) {
    const int HW = Hd * Wd;
    if (frame_idx < 1 || frame_idx > T || ref_num < 1) return -1;
    int* sidx = (int*)malloc(sizeof(int) * (size_t)(ref_num > frame_idx ? ref_num : frame_idx));
    if (!sidx) return -1;
    const int N = vos_oracle_sample_frames(frame_idx, take_range, ref_num, sidx);
    const size_t rows = (size_t)N * HW;
    float* S = (float*)malloc(sizeof(float) * rows * HW);
    float* colmax = (float*)malloc(sizeof(float) * HW);
    float* colsum = (float*)malloc(sizeof(float) * HW);
    if (!S || !colmax || !colsum) { free(sidx); free(S); free(colmax); free(colsum); return -1; }

    /* S[r,t] = sum_c R[r,c] * T[c,t], accumulated in f32 in channel order */
#pragma omp parallel for schedule(static)
    for (size_t r = 0; r < rows; ++r) {
        const int n = (int)(r / HW), p = (int)(r % HW);
        const float* f = ref + (size_t)sidx[n] * C * HW + p;
        float* s = S + r * HW;
        for (int t = 0; t < HW; ++t) s[t] = 0.0f;
        for (int c = 0; c < C; ++c) {
            const float a = f[(size_t)c * HW];
            const float* tr = target + (size_t)c * HW;
            for (int t = 0; t < HW; ++t) s[t] += a * tr[t];
        }
        for (int t = 0; t < HW; ++t) s[t] *= temperature;
    }
    /* column softmax (dim=0) */
    for (int t = 0; t < HW; ++t) { colmax[t] = -INFINITY; colsum[t] = 0.0f; }
    for (size_t r = 0; r < rows; ++r) {
        const float* s = S + r * HW;
        for (int t = 0; t < HW; ++t) if (s[t] > colmax[t]) colmax[t] = s[t];
    }
#pragma omp parallel for schedule(static)
    for (size_t r = 0; r < rows; ++r) {
        float* s = S + r * HW;
        for (int t = 0; t < HW; ++t) s[t] = expf(s[t] - colmax[t]);
    }
    for (size_t r = 0; r < rows; ++r) {
        const float* s = S + r * HW;
        for (int t = 0; t < HW; ++t) colsum[t] += s[t];
    }
    const float s1 = (float)((double)sigma1 * (double)sigma1), s2 = (float)((double)sigma2 * (double)sigma2);
#pragma omp parallel for schedule(static)
    for (size_t r = 0; r < rows; ++r) {
        const int n = (int)(r / HW), p = (int)(r % HW);
        float* s = S + r * HW;
        if (probability_propagation) {
            for (int t = 0; t < HW; ++t) s[t] = s[t] / colsum[t];
        } else {
            const int sparse = (frame_idx > 15) && (n < N - VOS_CONTINUOUS_FRAME);
            const float sg = sparse ? s2 : s1;
            for (int t = 0; t < HW; ++t) s[t] = (s[t] / colsum[t]) * spatial_weight_entry(p, t, Wd, sg);
        }
    }
    /* out[k,t] = sum_r L[k,r] * A[r,t] */
#pragma omp parallel for schedule(static)
    for (int k = 0; k < d; ++k) {
        float* o = out + (size_t)k * HW;
        for (int t = 0; t < HW; ++t) o[t] = 0.0f;
        for (size_t r = 0; r < rows; ++r) {
            const int n = (int)(r / HW), p = (int)(r % HW);
            const float l = ref_label[((size_t)k * T + sidx[n]) * HW + p];
            if (l == 0.0f) continue;
            const float* s = S + r * HW;
            for (int t = 0; t < HW; ++t) o[t] += l * s[t];
        }
    }
    free(sidx); free(S); free(colmax); free(colsum);
    return 0;
}

/* Per-frame glue, src/utils/inference_utils.py:70,74-75: argmax over classes (first maximum wins, as
 * torch.argmax on CPU), low-res class map (HW) and nearest-upsampled mask (H,W).  argmax and nearest
 * up-sampling commute exactly, so the mask is the up-sampled class map. */
void vos_oracle_argmax_upsample(const float* pred, int d, int Hd, int Wd, int H, int W, uint8_t* cls_lowres,
                                uint8_t* mask) {
    const int HW = Hd * Wd;
    for (int t = 0; t < HW; ++t) {
        int best = 0;
        float bv = pred[t];
        for (int k = 1; k < d; ++k) {
            const float v = pred[(size_t)k * HW + t];
            if (v > bv) { bv = v; best = k; }
        }
        cls_lowres[t] = (uint8_t)best;
    }
    if (!mask) return;
    for (int y = 0; y < H; ++y) {
        const int sy = nearest_src(y, Hd, H);
        for (int x = 0; x < W; ++x) mask[(size_t)y * W + x] = cls_lowres[(size_t)sy * Wd + nearest_src(x, Wd, W)];
    }
}
