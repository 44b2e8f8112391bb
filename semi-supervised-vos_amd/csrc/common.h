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
constexpr int kTopkMax = 32;       // largest k of the top-k variant (list length kept per lane in pass 1)
constexpr int kTopkCap = 16 * kTopkMax;   // candidates per target pixel that can exceed the pass-1 bound (see prop_bf16.h)
constexpr int kXcd = 8;            // XCDs: blocks b and b+8 share an L2 (placement is a speed matter only)

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

// Work decomposition of one propagation ("stream-K" over reference tiles, XCD-partitioned):
//   the N*tiles_per_frame reference steps are cut into kXcd contiguous parts; the workgroups with
//   blockIdx % 8 == x walk part x for every target tile, so that part's features stay in that XCD's L2.
//   Inside an XCD the (target tile, step) space of TT * |part| steps is cut evenly over `wg_per_xcd`
//   workgroups, target-major, so a workgroup streams a contiguous run of reference tiles against one
//   (rarely two or three) target tiles.  Both the kernel and combine_kernel evaluate this map.
struct WorkMap {
    int TT;           // target tiles (kBT pixels each)
    int NT;           // reference steps = n_ref * tiles_per_frame
    int wg_per_xcd;   // I
    int max_parts;    // partial slots reserved per workgroup and phase
    int phases;       // P: every XCD's part of the reference stream is walked in P consecutive pieces, ALL its workgroups on
                      // the same piece at the same time, so the piece (|stream| / (8 P) ~ 2 MB at 480p, P = 2) stays in that
                      // XCD's 4 MB L2.  With P = 1 the workgroups sit at 32 evenly spread offsets of a 4.2 MB cyclic stream:
                      // reuse distance = the whole stream, 47 % of the L2 requests miss (measured, 500 MB per launch).

    __host__ __device__ inline void xcd_range(int x, int& r0, int& r1) const {   // phases == 1 (v5 / v6 kernels)
        r0 = (int)((long long)x * NT / kXcd);
        r1 = (int)((long long)(x + 1) * NT / kXcd);
    }
    __host__ __device__ inline void part_range(int x, int ph, int& r0, int& r1) const {
        const long long k = (long long)x * phases + ph, n = (long long)kXcd * phases;
        r0 = (int)(k * NT / n);
        r1 = (int)((k + 1) * NT / n);
    }
    // [q0, q1) of workgroup i in XCD x, in units of steps of the flattened (tt, step) space
    __host__ __device__ inline void wg_range(int rx, int i, long long& q0, long long& q1) const {
        const long long Q = (long long)TT * rx;
        q0 = Q * i / wg_per_xcd;
        q1 = Q * (i + 1) / wg_per_xcd;
    }
};

// Per-launch description of one propagation (passed by value as a kernel argument).
struct PropArgs {
    const bf16_t* feat_ring;    // [cap][HWp][kC]        pixel-major bf16 features
    const bf16_t* coord_tab;    // [HWp/32][2][32][8]    reference-side spatial channels
    const bf16_t* lab_hi;       // [cap][HWp/32][2][64][8] labels in MFMA A-operand order (hi part)
    const bf16_t* lab_lo;       // same, low part (probability mode) or nullptr
    float* part;                // [8*I][max_parts][2+d][kBT]  per-workgroup partial (m, l, numerators)
    int slot[kMaxRef];          // ring slot of each sampled reference frame
    unsigned long long sparse_mask;   // bit n set: frame n uses sigma2 (the "interval" frames)
    WorkMap map;
    int target_slot;
    int n_ref;
    int HW, HWp, Wd;
    int d;
    int tiles_per_frame;        // HWp / 32
    float c;                    // temperature * log2(e)
    double g1, g2;              // 1 / (sigma^2 * temperature) for sigma1, sigma2
    double two_over_w, gamma;   // 2/W_d, 1 + 1/W_d^2
    int part_rows;              // rows of one partial slot: 2 + d (dense), 1 + 2*kTopkMax (top-k pass 1), 2 (top-k pass 2)
    // top-k variant (two passes, see prop_bf16.h)
    const float* tk_thr;        // [HWp] pass 2: lower bound of the k-th largest weighted exponent of each target pixel
    const float* tk_m;          // [HWp] pass 2: column max of the raw scores (exact softmax max, from pass 1)
    unsigned* tk_cnt;           // [HWp] pass 2: number of candidates appended per target pixel
    uint2* tk_cand;             // [HWp][kTopkCap] pass 2: (exponent bits, reference row id = n*HWp + p)
    unsigned long long* dbg;    // diagnostic builds only (-DVOSPROP_STAMP): per-wave cycle sums; else nullptr
};

}  // namespace vosprop
