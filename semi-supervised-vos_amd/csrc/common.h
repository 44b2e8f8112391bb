// Shared device/host definitions of the vosprop engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

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
constexpr int kXcd = 8;            // XCDs: blocks b and b+8 share an L2 (placement is a speed matter only)
constexpr int kTkListCap = 1024;   // top-k pass 2: marked reference tiles one workgroup walks at most (its share of a target tile's)
constexpr float kTkDummy = -3.0e38f;   // "no group": below every real weighted exponent (masked rows sit at -1e30 c)

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
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

// Work decomposition of one propagation.  The (target tile, reference tile) space is cut into SEGMENTS - a run of
// consecutive reference tiles against one target tile - and every workgroup walks a short list of them (built on the host,
// engine.hip get_plan; the kernels only read their list).  The reference stream is cut into kXcd contiguous parts and the
// workgroups with blockIdx % 8 == x (they run on XCD x: tools/xcc_probe.hip) only touch part x, so each L2 sees 1/8 of the
// features.  Inside an XCD the workgroups walk their part IN LOCKSTEP (same reference tiles at the same time, each against
// its own target tile) so a tile is fetched from HBM once and then hit in L2 by the other workgroups; the target tiles left
// over when their number is not a multiple of the workgroup count are split between "primary" workgroups (head of the part)
// and "extra" ones (the tail of the part, a sub-megabyte region that stays in L2 as well).
struct Segment {
    int tt;        // target tile (kBT pixels)
    int r_lo;      // first reference tile; the stream is walked pixel tile by pixel tile with the N sampled frames inner:
                   // r = pixel_tile * N + frame  (so a wave reuses its spatial-prior tile across the frames of a pixel tile)
    int n_steps;   // reference tiles in the run
    int slot;      // partial slot written by this segment: part[slot][part_rows][kBT]
};

// Per-launch description of one propagation (passed by value as a kernel argument).
struct PropArgs {
    const bf16_t* feat_ring;    // [cap][HWp][kC]        pixel-major bf16 features (VOSPROP_PREC_BF16)
    const float* feat_f32;      // [cap][HWp][kC]        pixel-major f32 features (VOSPROP_PREC_F32; feat_ring is null then)
    const float2* coord_f32;    // [HWp] (row = p / W as f32 true division, col = p % W): the reference's coordinates (f32 path)
    float sig1_sq, sig2_sq;     // sigma^2 as f32 (f32 path: w = exp(-d2 / sigma^2), reference predict.py:173)
    const bf16_t* coord_tab;    // [HWp/32][2][32][8]    reference-side spatial channels
    const bf16_t* lab_hi;       // [cap][HWp/32][2][64][8] labels in MFMA A-operand order (hi part)
    const bf16_t* lab_lo;       // same, low part (probability mode) or nullptr
    const bf16_t* lab16;        // [cap][HWp/32][64][8] one-hot labels of <= 16 classes as ONE 16x16x32 MFMA A fragment per tile (prop_mask.h)
    bf16_t* smat;               // materialised-affinity variant only: [N * tiles][column blocks][64][16] bf16 score tiles
    float* part;                // [n_segments][part_rows][kBT]  per-segment partial (m, l, numerators)
    int slot[kMaxRef];          // ring slot of each sampled reference frame
    unsigned long long sparse_mask;   // bit n set: frame n uses sigma2 (the "interval" frames)
    const Segment* segs;        // segment table of this launch
    const int* seg_off;         // [grid + 1] segments of workgroup b: segs[seg_off[b]] .. segs[seg_off[b + 1] - 1]
    const Segment* seg_first;   // [grid] a copy of segs[seg_off[b]] (zeros for a workgroup without work): the first record loads beside
                                // the range, not behind it (one dependent round trip off every launch)
    int target_slot;
    const bf16_t* target_feat;  // dense bf16 kernel: the target frame's features [target_rows][kC] - its ring slot, or the caller's own
    int target_rows;            // channels-last bf16 buffer (HW rows) when the ring copy rides in combine_kernel (engine.hip vosprop_step)
    int target_f16;             // [r3] the caller's buffer holds f16 (an f16 encoder's output): converted to bf16 as it is loaded
    int n_ref;
    int HW, HWp, Wd;
    int d;
    int tiles_per_frame;        // HWp / 32
    float c;                    // temperature * log2(e)
    double g1, g2;              // 1 / (sigma^2 * temperature) for sigma1, sigma2
    double two_over_w, gamma;   // 2/W_d, 1 + 1/W_d^2
    int part_rows;              // rows of one partial slot: 2 + d (dense), 1 + 2*kTopkMax (top-k pass 1), 2 (top-k pass 2)
    // top-k variant (two passes on the dense kernel's pipeline, prop_dense.h TK = 1 / 2)
    int tk_k;                   // k of the top-k variant (pass 1 keeps KS = ceil(k / 8) * 8 list slots per lane)
    const float* tk_thr;        // [TT*256] pass 2: a group whose packed maximum reaches this holds candidates of the target pixel
    // [r3] top-k on the dense kernel's pipeline (prop_dense.h TK = 1 / 2, aux_kernels.h topk_select2 / topk_combine2)
    const int* tk_off;          // [TT + 1] partial slots of target tile tt: tk_off[tt] .. tk_off[tt + 1] - 1 (pass 1 lays its lists out
                                //   per target pixel: [tt][column][slot rank][half][KS])
    int tk_idx_bits;            // low mantissa bits of a packed group maximum that hold (stream index << 1 | half)
    unsigned* tk_bitmap;        // [TT][tk_words] one bit per reference tile (stream index): some column of the target tile has a
    int tk_words;               //   candidate group there; cleared by pass 1, marked by topk_select2_kernel, walked by pass 2
    int tk_bitmap_words;        // TT * tk_words
    int tk_chunks;              // pass 2: workgroups that share one target tile's marked tiles
    int tk_cap;                 // pass 2: groups one lane can dump per share
    float* tk_dump;             // [TT*256][2][tk_chunks][tk_cap][16] weighted exponents of the dumped groups
    unsigned* tk_dump_r;        // [TT*256][2][tk_chunks][tk_cap]     their tiles: frame << 16 | pixel tile
    unsigned* tk_cnt;           // [TT*256][2][tk_chunks]             groups dumped
    unsigned* tk_over;          // [3] capacity clamps that fired since vosprop_begin_video: dump slots of a lane, groups of a pixel in the
                                //   combine kernel, candidates of a pixel in the select kernel (vosprop_topk_overflows)
    // prop_mask_kernel (prop_mask.h): target-side constants of the prior MFMA, built once per engine (engine.hip build_target_consts)
    const void* tc_b;           // [2 sigma][HWp][2 k halves] bf16x8: B fragment of the prior MFMA with c folded in and the 3-way split of -g Q_t c
    const float* tc_kq;         // [2 sigma][HWp] g Q_t c
    unsigned long long* dbg;    // debug hook only (vosprop_debug_mask_stamps): [grid][8] wall-clock stamps of workgroup phases, else nullptr
};

}  // namespace vosprop
