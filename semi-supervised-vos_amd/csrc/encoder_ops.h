// Encoder epilogue: y = act(y + bias[c] (+ residual)) in ONE pass over an NHWC tensor, in place.
// The encoder's convolutions stay MIOpen's (implicit-GEMM kernels); what PyTorch adds around every one of them - a broadcast
// bias add, a ReLU, and for the last convolution of a residual unit an element-wise add - are separate launches that together
// cost as much as the convolutions themselves at batch 16 (rocprofv3, profiles/r01_cli_kernel_time.txt: 315 ms of element-wise
// kernels against 330 ms of igemm per 800 frames).  HBM-bound, 16 B per lane, f32 arithmetic, one rounding at the end.
#pragma once
#include <hip/hip_fp16.h>

#include "common.h"

namespace vosprop {

template <typename T>
struct Vec8;
template <>
struct Vec8<bf16_t> {
    typedef bf16x8 type;
};
template <>
struct Vec8<_Float16> {
    typedef _Float16 type __attribute__((ext_vector_type(8)));
};

// n8 = number of 8-element groups (pixels * C / 8); C % 8 == 0 so a group never straddles two pixels
template <typename T, bool RELU, bool RES>
__global__ __launch_bounds__(256) void bias_act_kernel(T* __restrict__ y, const T* __restrict__ bias, const T* __restrict__ res,
                                                       long long n8, int C) {
    typedef typename Vec8<T>::type V;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const int c = (int)((i * 8) % C);
        V v = *(const V*)(y + i * 8);
        const V b = *(const V*)(bias + c);
        V r;
        if (RES) r = *(const V*)(res + i * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float f = (float)v[k] + (float)b[k];
            if (RES) f += (float)r[k];
            if (RELU) f = f > 0.0f ? f : 0.0f;
            v[k] = (T)f;
        }
        *(V*)(y + i * 8) = v;
    }
}

template <bool RELU, bool RES>
__global__ __launch_bounds__(256) void bias_act_f32_kernel(float* __restrict__ y, const float* __restrict__ bias,
                                                           const float* __restrict__ res, long long n4, int C) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const int c = (int)((i * 4) % C);
        float4 v = *(const float4*)(y + i * 4);
        const float4 b = *(const float4*)(bias + c);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        if (RES) {
            const float4 r = *(const float4*)(res + i * 4);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (RELU) {
            v.x = v.x > 0.0f ? v.x : 0.0f; v.y = v.y > 0.0f ? v.y : 0.0f;
            v.z = v.z > 0.0f ? v.z : 0.0f; v.w = v.w > 0.0f ? v.w : 0.0f;
        }
        *(float4*)(y + i * 4) = v;
    }
}

// Stem epilogue: y[n, oh, ow, c] = relu(max over the 3x3 / stride 2 / pad 1 window of x[n, ., ., c] + bias[c]) - the bias (folded
// BatchNorm), ReLU and max-pool that follow the 7x7 stem convolution (reference src/model/backbone/resnet.py:104-107, children
// 1-3 of the truncated ResNet) in one pass: the largest activation of the network (64 x H/2 x W/2) is read once and a quarter of
// it written, instead of read + written by the bias / ReLU pass and read again by the pooling kernel.  Bias add, rounding and
// ReLU are monotone, so taking the max first gives the same bits as pooling the rounded activations.  G elements (16 B) per
// lane, C % G == 0; HBM-bound.
template <typename T, typename V, int G>
__global__ __launch_bounds__(256) void bias_relu_maxpool_kernel(const T* __restrict__ x, const T* __restrict__ bias,
                                                                T* __restrict__ y, int N, int H, int W, int C, int OH, int OW) {
    const int cg = C / G;
    const long long total = (long long)N * OH * OW * cg;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % cg) * G;
        long long p = i / cg;
        const int ow = (int)(p % OW);
        p /= OW;
        const int oh = (int)(p % OH), n = (int)(p / OH);
        float m[G];
#pragma unroll
        for (int k = 0; k < G; ++k) m[k] = -3.0e38f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int ih = oh * 2 + dy;
            if (ih < 0 || ih >= H) continue;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int iw = ow * 2 + dx;
                if (iw < 0 || iw >= W) continue;
                const V v = *(const V*)(x + (((long long)n * H + ih) * W + iw) * C + c);
#pragma unroll
                for (int k = 0; k < G; ++k) m[k] = fmaxf(m[k], (float)v[k]);
            }
        }
        const V b = *(const V*)(bias + c);
        V o;
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const T r = (T)(m[k] + (float)b[k]);      // round as the separate bias pass does, then ReLU
            o[k] = (float)r > 0.0f ? r : (T)0.0f;
        }
        *(V*)(y + i * G) = o;
    }
}

}  // namespace vosprop
