// Shared device/host definitions of the vosprop engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vosprop {

constexpr int kC = 256;            // embedding width (reference src/model/vos_net.py:22-23)
constexpr int kTileR = 32;         // reference rows (pixels) per MFMA tile
constexpr int kWaves = 8;          // waves per workgroup of the propagation kernel
constexpr int kColsPerWave = 32;   // target pixels per wave (one 32x32 MFMA column block)
constexpr int kBT = kWaves * kColsPerWave;   // 256 target pixels per workgroup
constexpr int kMaxClasses = 32;
constexpr int kMaxRef = 64;
constexpr int kCoordCh = 16;       // extra K channels that carry the spatial prior
constexpr int kContinuousFrame = 4;   // reference src/config.py:13

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// Row of a 32x32 MFMA accumulator held in register `reg` by a lane of half `h` (= lane >> 5):
// v_mfma_f32_32x32x16_bf16 C/D layout, col = lane & 31.
__host__ __device__ inline int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// Round-to-nearest-even f32 -> bf16 kept as f32 (finite inputs only).
__host__ __device__ inline float bf16_round(float x) {
    union { float f; uint32_t u; } v;
    v.f = x;
    v.u = (v.u + 0x7FFFu + ((v.u >> 16) & 1u)) & 0xFFFF0000u;
    return v.f;
}
__host__ __device__ inline uint16_t bf16_bits(float x) {
    union { float f; uint32_t u; } v;
    v.f = x;
    return (uint16_t)((v.u + 0x7FFFu + ((v.u >> 16) & 1u)) >> 16);
}
// x ~= h + m + l with each piece exactly representable in bf16 (24 significant bits in total).
__host__ __device__ inline void split3(float x, float& h, float& m, float& l) {
    h = bf16_round(x);
    m = bf16_round(x - h);
    l = bf16_round(x - h - m);
}

// Per-launch description of one propagation (passed by value as a kernel argument).
struct PropArgs {
    const bf16_t* feat_ring;    // [cap][HWp][kC]        pixel-major bf16 features
    const bf16_t* coord_tab;    // [HWp/32][2][32][8]    reference-side spatial channels
    const bf16_t* lab_hi;       // [cap][HWp/32][2][64][8] labels in MFMA A-operand order (hi part)
    const bf16_t* lab_lo;       // same, low part (probability mode) or nullptr
    float* part;                // [TT][U][2+d][kBT]     per-unit partial (m, l, numerators)
    int slot[kMaxRef];          // ring slot of each sampled reference frame
    unsigned long long sparse_mask;   // bit n set: frame n uses sigma2 (the "interval" frames)
    int target_slot;
    int n_ref;
    int HW, HWp, Wd;
    int d;
    int tiles_per_frame;        // HWp / 32
    int row_splits;             // RS: workgroups per (target tile, frame)
    int tiles_per_split;
    float c;                    // temperature * log2(e)
    double g1, g2;              // 1 / (sigma^2 * temperature) for sigma1, sigma2
    double two_over_w, gamma;   // 2/W_d, 1 + 1/W_d^2
};

}  // namespace vosprop
