// vosprop engine: host side of the C ABI declared in include/vosprop.h.
// Owns the feature/label ring in HBM, plans each propagation, launches the HIP kernels.
#include "../../include/vosprop.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "aux_kernels.h"
#include "common.h"
#include "prop_bf16.h"
#include "prop_dense.h"
#include "prop_mask.h"
#include "prop_f32.h"
#include "pointwise.h"
#include "encoder_ops.h"

using namespace vosprop;

namespace {

struct Ring {
    bf16_t* feat = nullptr;    // [cap][HWp][kC]          (VOSPROP_PREC_BF16)
    float* featf = nullptr;    // [cap][HWp][kC] f32      (VOSPROP_PREC_F32: the features are never rounded)
    bf16_t* lab_hi = nullptr;  // [cap][tiles][2][64][8]
    bf16_t* lab_lo = nullptr;
    bf16_t* lab16 = nullptr;   // [cap][tiles][64][8]: the 16x16x32 label fragment (one-hot, <= 16 classes; prop_mask.h)
    uint8_t* cls = nullptr;    // [cap][HWp]
    int cap = 0;
};

struct LastProp {
    bool valid = false;
    PropArgs args;        // dense: the launch; top-k: pass 2 (pass 1 differs in part_rows only)
    int grid = 0;
    bool prob = false, lab_lo = false;
    bool materialise = false;
    bool no_l = false;    // dense label mode, prediction not requested: prop_mask_kernel
    int plan_last_active = 0;      // which plan of (NT) the launch walks (Plan::last_active)
    int topk = 0;
    // top-k (prop_dense.h TK 1 / 2): pass 1 = `args` with the list partials, then topk_select2_kernel, then pass 2 on its own grid
    int tk_ks = 0;                 // list slots per lane: ceil(k / 8) * 8
    int tk_grid2 = 0;              // pass 2: TT * tk_chunks workgroups
    bool tk_norm = false;          // the prediction is wanted: a dense launch first (its partials carry the softmax max / denominator)
    float* tk_part = nullptr;      // pass-1 lists [segments][2 * KS][kBT]
    float* dense_part = nullptr;   // the dense partials when tk_norm
    int dense_rows = 0;
    TopkSelectArgs sel;
};

// Device copy of the partial-slot lists of one work decomposition (depends only on TT, NT).
struct Plan {
    int NT = -1;
    int last_active = 0;      // build_segments' last_active this plan was built for (0: every step costs the same)
    int grid = 0;             // workgroups = 8 * wg_per_xcd
    int wg_per_xcd = 0;
    int steps_per_wg = 0;     // reference tiles the busiest workgroup walks (stats)
    int max_seg_steps = 0;    // the longest segment (prop_mask_kernel builds one control table per segment: kMaskMaxSteps)
    int n_parts = 0;          // total partial slots = number of segments
    int* d_off = nullptr;     // [TT + 1]
    std::vector<int> h_off;   // host copy of d_off
    int* d_list = nullptr;    // slot indices, CSR by target tile
    Segment* d_segs = nullptr;
    int* d_seg_off = nullptr; // [grid + 1]
    Segment* d_first = nullptr; // [grid] each workgroup's first record again (PropArgs::seg_first)
};

}  // namespace

struct vosprop_ctx {
    vosprop_config cfg;
    int HW = 0, HWp = 0, tiles = 0, TT = 0;
    Ring ring;            // video state
    Ring scratch;         // stateless vosprop_predict
    bf16_t* coord_tab = nullptr;
    float2* coord_f32 = nullptr;   // [HWp] the reference's f32 pixel coordinates (f32 path)
    uint16_t* tc_b = nullptr;      // prop_mask_kernel: target-side prior constants (build_target_consts) ...
    float* tc_kq = nullptr;
    float tc_sig1 = 0.f, tc_sig2 = 0.f, tc_temp = 0.f;   // ... and what they were built for
    float* part = nullptr;
    size_t part_bytes = 0;
    bf16_t* smat = nullptr;        // materialised-affinity variant: score tiles in HBM (grown on demand)
    size_t smat_bytes = 0;
    float* pred_buf = nullptr;     // (kMaxClasses, HW) f32
    uint8_t* cls_tmp = nullptr;    // (HWp)
    uint8_t* stage_host = nullptr; // pinned (HWp): first labels of a video on their way to the GPU (stream-ordered upload)
    hipEvent_t stage_ev = nullptr;
    bool stage_busy = false;
    // in-situ kernel timing (vosprop_timing_begin / _read): event pairs around every propagation-kernel launch
    bool timing = false;
    std::vector<hipEvent_t> tev;   // pairs (start, stop), created on demand
    size_t tev_used = 0;
    // top-k scratch (grown on demand by propagate(): sizes depend on the number of sampled frames)
    float* tk_thr = nullptr;       // [TT*256] group threshold (pass 2)
    float* tk_thr_elem = nullptr;  // [TT*256] element threshold (combine)
    unsigned* tk_bitmap = nullptr; // [TT][words]
    size_t tk_bitmap_bytes = 0;
    float* tk_part = nullptr;      // pass-1 lists
    size_t tk_part_bytes = 0;
    float* tk_dump = nullptr;      // [TT*256][2][chunks][cap][16]
    unsigned* tk_dump_r = nullptr;
    unsigned* tk_cnt = nullptr;
    unsigned* tk_over = nullptr;   // [3] capacity-clamp counters of the top-k kernels (vosprop_topk_overflows)
    size_t tk_dump_groups = 0;     // capacity in groups (= TT*256*2*chunks*cap)
    size_t tk_cnt_units = 0;
    // video state
    bool in_video = false;
    int frame_idx = 0;
    int d = 0, H = 0, W = 0;
    // nearest up-sampling fused into combine_kernel (dense path): inverse index tables for the current output size
    int* up_tab = nullptr;         // device: y0[feat_h + 1], x0[feat_w + 1]
    std::vector<int> up_host;
    int up_H = 0, up_W = 0;
    float up_sy = 0.f, up_sx = 0.f;
    uint8_t* fuse_mask = nullptr;  // set by vosprop_step around propagate(): the mask combine_kernel should write
    unsigned tseq = 0;                 // eligible launches seen since vosprop_timing_begin
    bool mask_only = false;            // set by vosprop_step around propagate(): the caller did not ask for the prediction
    bool fuse_f16 = false;             // ... and they are f16 (converted where they are read)
    const void* fuse_push = nullptr;   // set by vosprop_step around propagate(): channels-last bf16 features of the target frame that
                                       // the propagation reads in place and combine_kernel copies into the target's ring slot
    LastProp last;
    std::vector<Plan> plans;   // cache keyed by NT (n_ref * tiles)
    vosprop_stats stats;
    std::string err;
};

namespace {

int fail(vosprop_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                               \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return fail(ctx, VOSPROP_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

int ring_alloc(vosprop_ctx* ctx, Ring& r, int cap) {
    const bool f32 = ctx->cfg.precision == VOSPROP_PREC_F32;
    const size_t feat_b = (size_t)cap * ctx->HWp * kC * (f32 ? sizeof(float) : sizeof(bf16_t));
    const size_t lab_b = (size_t)cap * ctx->tiles * 2 * 64 * 8 * sizeof(bf16_t);
    const size_t cls_b = (size_t)cap * ctx->HWp;
    void* fp = nullptr;
    HIP_TRY(ctx, hipMalloc(&fp, feat_b));
    if (f32) r.featf = (float*)fp;
    else r.feat = (bf16_t*)fp;
    HIP_TRY(ctx, hipMalloc((void**)&r.lab_hi, lab_b));
    HIP_TRY(ctx, hipMalloc((void**)&r.lab_lo, lab_b));
    HIP_TRY(ctx, hipMalloc((void**)&r.lab16, lab_b / 2));
    HIP_TRY(ctx, hipMemset(r.lab16, 0, lab_b / 2));
    HIP_TRY(ctx, hipMalloc((void**)&r.cls, cls_b));
    HIP_TRY(ctx, hipMemset(fp, 0, feat_b));
    HIP_TRY(ctx, hipMemset(r.lab_hi, 0, lab_b));
    HIP_TRY(ctx, hipMemset(r.lab_lo, 0, lab_b));
    HIP_TRY(ctx, hipMemset(r.cls, 0, cls_b));
    r.cap = cap;
    return VOSPROP_OK;
}

void ring_free(Ring& r) {
    if (r.feat) (void)hipFree(r.feat);
    if (r.featf) (void)hipFree(r.featf);
    if (r.lab_hi) (void)hipFree(r.lab_hi);
    if (r.lab_lo) (void)hipFree(r.lab_lo);
    if (r.lab16) (void)hipFree(r.lab16);
    if (r.cls) (void)hipFree(r.cls);
    r = Ring();
}

// Reference-side spatial channels, [tile][half][row][8]: per pixel p=(a,b):
//   ch0-2: a   ch3-5: b   ch6-8: Qh   ch9-10: Qm   ch11: Ql,   Q = a^2 + (2/W) a b + (1 + 1/W^2) b^2;
//   ch12-14: 1 (prop_mask.h: they carry the 3-way split of the column constant -(g Q_t c + M) on the target side; the other kernels
//   put 0 there);  ch15: 1 for the PADDED rows of the last tile (all their other channels are 0) - against a target-side -1e30 it
//   takes them out of the softmax.
int build_coord_table(vosprop_ctx* ctx) {
    const int W = ctx->cfg.feat_w;
    std::vector<uint16_t> tab((size_t)ctx->HWp * kCoordCh, 0);
    const double tw = 2.0 / W, gm = 1.0 + 1.0 / ((double)W * W);
    for (int p = 0; p < ctx->HW; ++p) {
        const double a = p / W, b = p % W;
        const double Q = a * a + tw * a * b + gm * b * b;
        const float qh = bf16_round((float)Q);
        const float qm = bf16_round((float)(Q - qh));
        const float ql = bf16_round((float)(Q - qh - qm));
        float ch[kCoordCh] = {(float)a, (float)a, (float)a, (float)b, (float)b, (float)b, qh, qh, qh, qm, qm, ql, 1, 1, 1, 0};
        const int tile = p / kTileR, row = p % kTileR;
        for (int c = 0; c < kCoordCh; ++c)
            tab[(((size_t)tile * 2 + c / 8) * 32 + row) * 8 + (c % 8)] = bf16_bits(ch[c]);
    }
    for (int p = ctx->HW; p < ctx->HWp; ++p) {
        const int tile = p / kTileR, row = p % kTileR;
        tab[(((size_t)tile * 2 + 1) * 32 + row) * 8 + 7] = bf16_bits(1.0f);
    }
    HIP_TRY(ctx, hipMalloc((void**)&ctx->coord_tab, tab.size() * 2));
    HIP_TRY(ctx, hipMemcpy(ctx->coord_tab, tab.data(), tab.size() * 2, hipMemcpyHostToDevice));
    return VOSPROP_OK;
}

// The reference's own coordinates (src/model/predict.py:167-168): row = index.div(float(W)) - a TRUE division of the flat index,
// in f32 - and col = index % W, for the f32 path, which evaluates the prior from them op for op.  Host f32 arithmetic is IEEE.
int build_coord_f32(vosprop_ctx* ctx) {
    const int W = ctx->cfg.feat_w;
    std::vector<float2> tab((size_t)ctx->HWp, make_float2(0.f, 0.f));
    for (int p = 0; p < ctx->HW; ++p) tab[(size_t)p] = make_float2((float)p / (float)W, (float)(p % W));
    HIP_TRY(ctx, hipMalloc((void**)&ctx->coord_f32, tab.size() * sizeof(float2)));
    HIP_TRY(ctx, hipMemcpy(ctx->coord_f32, tab.data(), tab.size() * sizeof(float2), hipMemcpyHostToDevice));
    return VOSPROP_OK;
}

// Target-side constants of prop_mask_kernel's prior MFMA for one (sigma1, sigma2, temperature): per target pixel t = (a_t, b_t)
// and sigma class, the B fragment that pairs with build_coord_table's channels - with c = temperature log2(e) folded in, so the MFMA
// yields log2 w + (the column constant) directly:
//   k half 0: split3(c g (2 a_t + (2/W) b_t)), split3(c g (2 gamma b_t + (2/W) a_t)), the first two parts of split3(-c g)
//   k half 1: the third part and again the first two of split3(-c g) (against Qh, Qm, Qm, Ql), split3(-c g Q_t), -1e30
// and kq = c g Q_t as f32 (the kernel re-splits -(kq + M) when a column's reference level M moves).  g = 1 / (sigma^2 temperature).
int build_target_consts(vosprop_ctx* ctx, float sigma1, float sigma2, float temperature) {
    if (ctx->tc_b && ctx->tc_sig1 == sigma1 && ctx->tc_sig2 == sigma2 && ctx->tc_temp == temperature) return VOSPROP_OK;
    const int W = ctx->cfg.feat_w, HWp = ctx->HWp;
    std::vector<uint16_t> tb((size_t)2 * HWp * 16, 0);
    std::vector<float> kq((size_t)2 * HWp, 0.f);
    const double tw = 2.0 / W, gm = 1.0 + 1.0 / ((double)W * W);
    const double c = (double)temperature * 1.4426950408889634;
    for (int sg = 0; sg < 2; ++sg) {
        const double sigma = sg ? sigma2 : sigma1;
        const double g = 1.0 / (sigma * sigma * (double)temperature);
        for (int t = 0; t < HWp; ++t) {
            const int tq = t < ctx->HW ? t : ctx->HW - 1;
            const double at = tq / W, bt = tq % W;
            const double qt = at * at + tw * at * bt + gm * bt * bt;
            float ah, am, al, bh, bm, bl, kh, km, kl, sh, sm, sl;
            split3((float)(c * g * (2.0 * at + tw * bt)), ah, am, al);
            split3((float)(c * g * (2.0 * gm * bt + tw * at)), bh, bm, bl);
            split3((float)(-c * g), kh, km, kl);
            const float kqf = (float)(c * g * qt);
            split3(-kqf, sh, sm, sl);
            const float lo[8] = {ah, am, al, bh, bm, bl, kh, km};
            const float hi[8] = {kl, kh, km, kh, sh, sm, sl, -1.0e30f};
            uint16_t* o = &tb[((size_t)sg * HWp + t) * 16];
            for (int e = 0; e < 8; ++e) {
                o[e] = bf16_bits(lo[e]);
                o[8 + e] = bf16_bits(hi[e]);
            }
            kq[(size_t)sg * HWp + t] = kqf;
        }
    }
    if (!ctx->tc_b) {
        HIP_TRY(ctx, hipMalloc((void**)&ctx->tc_b, tb.size() * 2));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->tc_kq, kq.size() * sizeof(float)));
    } else {
        HIP_TRY(ctx, hipDeviceSynchronize());      // a propagation in flight may still read the old constants
    }
    HIP_TRY(ctx, hipMemcpy(ctx->tc_b, tb.data(), tb.size() * 2, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->tc_kq, kq.data(), kq.size() * sizeof(float), hipMemcpyHostToDevice));
    ctx->tc_sig1 = sigma1;
    ctx->tc_sig2 = sigma2;
    ctx->tc_temp = temperature;
    return VOSPROP_OK;
}

int ensure_part(vosprop_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->part_bytes) return VOSPROP_OK;
    if (ctx->part) (void)hipFree(ctx->part);
    ctx->part = nullptr;
    ctx->part_bytes = 0;
    HIP_TRY(ctx, hipMalloc((void**)&ctx->part, bytes));
    ctx->part_bytes = bytes;
    return VOSPROP_OK;
}

// Work decomposition for NT reference steps (see WorkMap in common.h) + the CSR list, per target tile, of the
// partial slots that will hold it.  Cached per NT; built on first use (a few microseconds of host work).
// Pure host function: the segment lists of every workgroup (index b = i * 8 + x runs on XCD x) for TT target tiles and NT
// reference tiles.  Returns the workgroups per XCD.
// last_active (0 = not told): how many of the LAST target tile's 8 waves hold a column of the map.  prop_mask_kernel runs the others in
// its staging-only form, and a workgroup on that tile then steps faster (480p, 1 active wave: 594 ns against 848 ns) - its primary
// is given a longer head for the same finishing time.
int build_segments(int TT, int NT, std::vector<std::vector<Segment>>& per_wg, int last_active = 0) {
    long long per_xcd = ((long long)TT * NT + kXcd - 1) / kXcd;
    long long Iq = per_xcd / 4;   // at least ~4 tile steps per workgroup
    const int I = (int)(Iq < 1 ? 1 : (Iq > 32 ? 32 : Iq));
    per_wg.assign((size_t)(kXcd * I), {});
    auto add = [&](int x, int i, int tt, int r_lo, int n) {
        if (n > 0) per_wg[(size_t)(i * kXcd + x)].push_back(Segment{tt, r_lo, n, 0});
    };
    const double w_last = last_active > 0 && last_active < kWaves ? 0.66 + 0.34 * last_active / kWaves : 1.0;   // step time on the last tile
    for (int x = 0; x < kXcd; ++x) {
        const int r0 = (int)((long long)x * NT / kXcd), r1 = (int)((long long)(x + 1) * NT / kXcd);
        const int RX = r1 - r0;
        if (RX <= 0) continue;
        // lockstep map: k full rounds (workgroup i takes target tile j*I + i over the whole part), then the leftover tiles
        const int k = TT / I, rem = TT - k * I;
        for (int j = 0; j < k; ++j)
            for (int i = 0; i < I; ++i) add(x, i, j * I + i, r0, RX);
        if (rem > 0) {
            const int E = I - rem;                                        // extra workgroups (>= 1)
            // head length U: the primaries walk U tiles in one segment (the last tile's primary U / w_last), the E extras share the
            // tails in several short segments; a segment start (target fragments, staging prologue, partial write) costs about as
            // much as `seg_cost` tile steps, so U is chosen to equalise  U + c  and  tail share + c * segments
            static const char* sc_env = getenv("VOSPROP_SEGCOST");
            const double seg_cost = sc_env ? atof(sc_env) : 16.0;   // measured: a segment start costs ~12.8 us = ~15 steps of prop_mask_kernel
                                                                    // (profiles/r04_segcost.txt: 480p 184.7 us at 9, 176.2 at 17, 176.1 at 21, 181.0 at 26)
            auto head_last = [&](int u) { const int ul = (int)(u / w_last); return ul < RX ? ul : RX; };
            int U = RX;
            double best = 1e30;
            for (int u = 1; u <= RX; ++u) {
                const int tl = RX - u, tll = RX - head_last(u);
                const double Q = (double)(rem - 1) * tl + tll;
                const int ntails = (tl > 0 ? rem - 1 : 0) + (tll > 0 ? 1 : 0);
                const double share = Q / E;
                const double nseg = Q > 0 ? (double)ntails / E + 1.0 : 0.0;
                const double cost = std::max(u + seg_cost, share + seg_cost * nseg);
                if (cost < best) { best = cost; U = u; }
            }
            // Small maps (few steps per workgroup): giving every leftover tile a WHOLE number of workgroups (1 + E / rem, equal pieces,
            // one segment each) beats the shared tail, whose second segment start is not amortised there (240p: 48 of 256
            // workgroups ran two segments of ~6 steps and finished 8 us behind the rest)
            {
                const int pieces_min = 1 + E / rem;
                const double cost_whole = (double)((RX + pieces_min - 1) / pieces_min) + seg_cost;
                if (cost_whole < best) {
                    int wg = 0;
                    for (int t = 0; t < rem; ++t) {
                        const int pieces = 1 + E / rem + (t < E % rem ? 1 : 0);
                        for (int pc = 0; pc < pieces; ++pc, ++wg) {
                            const int lo = (int)((long long)RX * pc / pieces), hi = (int)((long long)RX * (pc + 1) / pieces);
                            add(x, wg, k * I + t, r0 + lo, hi - lo);
                        }
                    }
                    continue;
                }
            }
            std::vector<int> head((size_t)rem, U);
            head[(size_t)rem - 1] = head_last(U);
            for (int i = 0; i < rem; ++i) add(x, i, k * I + i, r0, head[(size_t)i]);   // primaries: head of their own tile, in lockstep
            // the tails, end to end, walked by the extras: position q lies in tile t at offset q - pre[t]
            std::vector<long long> pre((size_t)rem + 1, 0);
            for (int t = 0; t < rem; ++t) pre[(size_t)t + 1] = pre[(size_t)t] + (RX - head[(size_t)t]);
            const long long Q = pre[(size_t)rem];
            int t = 0;
            for (int e = 0; e < E && Q > 0; ++e) {
                long long q = Q * e / E;
                const long long q1 = Q * (e + 1) / E;
                while (q < q1) {
                    while (pre[(size_t)t + 1] <= q) ++t;
                    long long qe = pre[(size_t)t + 1];
                    if (qe > q1) qe = q1;
                    add(x, rem + e, k * I + t, r0 + head[(size_t)t] + (int)(q - pre[(size_t)t]), (int)(qe - q));
                    q = qe;
                }
            }
        }
    }
    return I;
}

int get_plan(vosprop_ctx* ctx, int NT, const Plan** out, int last_active = 0) {
    for (const Plan& p : ctx->plans)
        if (p.NT == NT && p.last_active == last_active) { *out = &p; return VOSPROP_OK; }
    Plan p;
    p.NT = NT;
    p.last_active = last_active;
    const int TT = ctx->TT;
    std::vector<std::vector<Segment>> per_wg;
    p.wg_per_xcd = build_segments(TT, NT, per_wg, last_active);
    p.grid = kXcd * p.wg_per_xcd;
    // Partial slots are numbered so that the slots of one target tile are CONSECUTIVE: slot = off[tt] + (its rank among the tile's
    // segments).  The kernels that merge partials then need `off` only - the slot list is the identity and they do not load it (one
    // dependent round trip fewer in combine_kernel / topk_select_kernel / topk_combine_kernel).
    std::vector<int> off((size_t)TT + 1, 0);
    for (int b = 0; b < p.grid; ++b)
        for (const Segment& sg : per_wg[(size_t)b]) ++off[(size_t)sg.tt + 1];
    for (int tt = 0; tt < TT; ++tt) off[(size_t)tt + 1] += off[(size_t)tt];
    p.h_off = off;
    std::vector<int> next(off.begin(), off.end() - 1);
    std::vector<Segment> segs;
    std::vector<int> seg_off((size_t)p.grid + 1, 0);
    for (int b = 0; b < p.grid; ++b) {
        int steps = 0;
        for (Segment sg : per_wg[(size_t)b]) {
            sg.slot = next[(size_t)sg.tt]++;
            segs.push_back(sg);
            steps += sg.n_steps;
            if (sg.n_steps > p.max_seg_steps) p.max_seg_steps = sg.n_steps;
        }
        if (steps > p.steps_per_wg) p.steps_per_wg = steps;
        seg_off[(size_t)b + 1] = (int)segs.size();
    }
    p.n_parts = (int)segs.size();
    std::vector<int> flat((size_t)p.n_parts);
    for (int i = 0; i < p.n_parts; ++i) flat[(size_t)i] = i;      // identity (kept for the kernels' signatures and the tests)
    HIP_TRY(ctx, hipMalloc((void**)&p.d_off, off.size() * sizeof(int)));
    HIP_TRY(ctx, hipMalloc((void**)&p.d_list, (flat.size() + 1) * sizeof(int)));
    HIP_TRY(ctx, hipMalloc((void**)&p.d_segs, (segs.size() + 1) * sizeof(Segment)));
    HIP_TRY(ctx, hipMalloc((void**)&p.d_seg_off, seg_off.size() * sizeof(int)));
    std::vector<Segment> first((size_t)p.grid, Segment{0, 0, 0, 0});
    for (int b = 0; b < p.grid; ++b)
        if (seg_off[(size_t)b] < seg_off[(size_t)b + 1]) first[(size_t)b] = segs[(size_t)seg_off[(size_t)b]];
    HIP_TRY(ctx, hipMalloc((void**)&p.d_first, first.size() * sizeof(Segment)));
    HIP_TRY(ctx, hipMemcpy(p.d_first, first.data(), first.size() * sizeof(Segment), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(p.d_off, off.data(), off.size() * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(p.d_list, flat.data(), flat.size() * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(p.d_segs, segs.data(), segs.size() * sizeof(Segment), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(p.d_seg_off, seg_off.data(), seg_off.size() * sizeof(int), hipMemcpyHostToDevice));
    ctx->plans.push_back(p);
    *out = &ctx->plans.back();
    return VOSPROP_OK;
}

template <typename T, typename Out>
void launch_push(const void* src, Out* dst, int HW, hipStream_t s) {
    hipLaunchKernelGGL((push_kernel<T, Out>), dim3((HW + 63) / 64, kC / 64), dim3(256), 0, s, (const T*)src, dst, HW);
}

template <typename Out>
int push_features_to(vosprop_ctx* ctx, const void* src, int dtype, Out* dst, int same_dt, hipStream_t s) {
    if (dtype & VOSPROP_LAYOUT_HWC) {   // channels-last source: rows are in ring order already
        const int n8 = ctx->HW * kC / 8;
        const dim3 grid((n8 + 255) / 256), block(256);
        const int dt = dtype & ~VOSPROP_LAYOUT_HWC;
        if (dt == same_dt) {
            HIP_TRY(ctx, hipMemcpyAsync(dst, src, (size_t)ctx->HW * kC * sizeof(Out), hipMemcpyDeviceToDevice, s));
            return VOSPROP_OK;
        }
        switch (dt) {
            case VOSPROP_DT_F32: hipLaunchKernelGGL((push_hwc_kernel<float, Out>), grid, block, 0, s, (const float*)src, dst, n8); break;
            case VOSPROP_DT_F16: hipLaunchKernelGGL((push_hwc_kernel<__half, Out>), grid, block, 0, s, (const __half*)src, dst, n8); break;
            case VOSPROP_DT_BF16: hipLaunchKernelGGL((push_hwc_kernel<bf16_t, Out>), grid, block, 0, s, (const bf16_t*)src, dst, n8); break;
            default: return fail(ctx, VOSPROP_E_INVALID, "unknown feature dtype");
        }
        HIP_TRY(ctx, hipGetLastError());
        return VOSPROP_OK;
    }
    switch (dtype) {
        case VOSPROP_DT_F32: launch_push<float, Out>(src, dst, ctx->HW, s); break;
        case VOSPROP_DT_F16: launch_push<__half, Out>(src, dst, ctx->HW, s); break;
        case VOSPROP_DT_BF16: launch_push<bf16_t, Out>(src, dst, ctx->HW, s); break;
        default: return fail(ctx, VOSPROP_E_INVALID, "unknown feature dtype");
    }
    HIP_TRY(ctx, hipGetLastError());
    return VOSPROP_OK;
}

// features of one frame -> slot `slot` of the ring (bf16, or f32 unrounded on the parity path)
int push_features(vosprop_ctx* ctx, const void* src, int dtype, Ring& r, int slot, hipStream_t s) {
    const size_t off = (size_t)slot * ctx->HWp * kC;
    if (r.featf) return push_features_to<float>(ctx, src, dtype, r.featf + off, VOSPROP_DT_F32, s);
    return push_features_to<bf16_t>(ctx, src, dtype, r.feat + off, VOSPROP_DT_BF16, s);
}

// e0 / e1 (optional): HIP events attached to the dispatch itself (hipExtLaunchKernelGGL) - they carry the kernel's own start and
// end timestamps, like a profiler's, without the queue latency a pair of hipEventRecord calls around the launch would include
void launch_prop_mode(const LastProp& lp, const PropArgs& a_in, int mode, hipStream_t s, hipEvent_t e0 = nullptr,
                      hipEvent_t e1 = nullptr) {
    const dim3 grid(lp.grid), block(kWaves * 64);
    const PropArgs& a = a_in;
    if (mode == 1) {      // top-k pass 1 (list partials): one instantiation per list length
        switch (lp.tk_ks) {
            case 8: hipLaunchKernelGGL((prop_dense_kernel<false, false, 0, false, 1, 8>), grid, block, 0, s, a); break;
            case 16: hipLaunchKernelGGL((prop_dense_kernel<false, false, 0, false, 1, 16>), grid, block, 0, s, a); break;
            case 24: hipLaunchKernelGGL((prop_dense_kernel<false, false, 0, false, 1, 24>), grid, block, 0, s, a); break;
            default: hipLaunchKernelGGL((prop_dense_kernel<false, false, 0, false, 1, 32>), grid, block, 0, s, a); break;
        }
        return;
    }
    if (mode == 2) {      // top-k pass 2 (marked tiles only)
        hipLaunchKernelGGL((prop_dense_kernel<false, false, 0, false, 2, 8>), dim3(lp.tk_grid2), block, 0, s, a);
        return;
    }
    if (a.feat_f32) {   // VOSPROP_PREC_F32: the parity kernel (prop_f32.h)
        if (lp.prob) {
            if (lp.lab_lo) hipLaunchKernelGGL((prop_f32_kernel<true, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((prop_f32_kernel<true, false>), grid, block, 0, s, a);
        } else {
            if (lp.lab_lo) hipLaunchKernelGGL((prop_f32_kernel<false, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((prop_f32_kernel<false, false>), grid, block, 0, s, a);
        }
        return;
    }
    // dense path: the in-wave pipelined kernel (prop_dense.h); the mask-only label-mode step is the hand-ordered prop_mask_kernel
    if (lp.materialise) {   // the materialised-affinity variant: score tiles out to HBM, then back in (prop_dense.h MAT 1 / 2)
        if (lp.prob) {
            hipLaunchKernelGGL((prop_dense_kernel<true, true, 1>), grid, block, 0, s, a);
            hipLaunchKernelGGL((prop_dense_kernel<true, true, 2>), grid, block, 0, s, a);
        } else {
            hipLaunchKernelGGL((prop_dense_kernel<false, false, 1>), grid, block, 0, s, a);
            if (lp.lab_lo) hipLaunchKernelGGL((prop_dense_kernel<false, true, 2>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((prop_dense_kernel<false, false, 2>), grid, block, 0, s, a);
        }
        return;
    }
    if (lp.prob) {
        if (lp.lab_lo) hipLaunchKernelGGL((prop_dense_kernel<true, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((prop_dense_kernel<true, false>), grid, block, 0, s, a);
    } else {
        if (lp.lab_lo) hipLaunchKernelGGL((prop_dense_kernel<false, true>), grid, block, 0, s, a);
        else if (lp.no_l) {      // the mask-only form (prop_mask.h)
            if (e0) hipExtLaunchKernelGGL(prop_mask_kernel, grid, block, 0, s, e0, e1, 0, a);
            else hipLaunchKernelGGL(prop_mask_kernel, grid, block, 0, s, a);
        } else if (e0) hipExtLaunchKernelGGL((prop_dense_kernel<false, false>), grid, block, 0, s, e0, e1, 0, a);
        else hipLaunchKernelGGL((prop_dense_kernel<false, false>), grid, block, 0, s, a);
    }
}

// The propagation kernel(s) of one step: dense = one launch; top-k = pass 1, select, pass 2 (and, in front of them, a dense
// launch for the softmax max / denominators when the caller wants the normalised prediction, not just the mask).
void launch_prop(const vosprop_ctx* ctx, const LastProp& lp, hipStream_t s, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
    if (lp.topk == 0) { launch_prop_mode(lp, lp.args, 0, s, e0, e1); return; }
    if (lp.tk_norm) {
        LastProp dl = lp;
        dl.topk = 0;
        dl.no_l = false;
        PropArgs ad = lp.args;
        ad.part = lp.dense_part;
        ad.part_rows = lp.dense_rows;
        launch_prop_mode(dl, ad, 0, s);
    }
    PropArgs a1 = lp.args;
    a1.part = lp.tk_part;
    a1.part_rows = 2 * lp.tk_ks;
    launch_prop_mode(lp, a1, 1, s);
    hipLaunchKernelGGL(topk_select2_kernel, dim3(ctx->TT * (kBT / kTkSelCols)), dim3(kTkSelCols * 64), 0, s, lp.sel);
    launch_prop_mode(lp, lp.args, 2, s);
}

// device buffer of at least `bytes` (grow-only)
template <typename T>
int ensure_buf(vosprop_ctx* ctx, T** p, size_t* have, size_t bytes, hipStream_t s) {
    if (bytes <= *have) return VOSPROP_OK;
    HIP_TRY(ctx, hipStreamSynchronize(s));      // work in flight may still use the old buffer
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *have = 0;
    HIP_TRY(ctx, hipMalloc((void**)p, bytes));
    *have = bytes;
    return VOSPROP_OK;
}

// One propagation: sampled frames `idx` (history indices, ring slot = idx % cap) against the target slot.
int propagate(vosprop_ctx* ctx, Ring& ring, const int* slots, int n_ref, int frame_idx, int target_slot, int d,
              bool prob, bool lab_lo, float sigma1, float sigma2, float temperature, float* pred, uint8_t* cls,
              bf16_t* new_lab_hi, bf16_t* new_lab_lo, hipStream_t s, bf16_t* new_lab16 = nullptr) {
    const int topk = ctx->cfg.topk;
    if (n_ref < 1 || n_ref > kMaxRef) return fail(ctx, VOSPROP_E_INVALID, "n_ref out of range");
    if (d < 1 || d > kMaxClasses) return fail(ctx, VOSPROP_E_UNSUPPORTED, "d > VOSPROP_MAX_CLASSES");
    if (!(temperature > 0.0f)) return fail(ctx, VOSPROP_E_UNSUPPORTED, "temperature must be > 0");
    const bool f32 = ctx->cfg.precision == VOSPROP_PREC_F32;
    if (f32 && topk != 0) return fail(ctx, VOSPROP_E_UNSUPPORTED, "top-k is built on the bf16 path only");
    if (topk != 0 && prob) return fail(ctx, VOSPROP_E_UNSUPPORTED, "top-k is built for label propagation only");
    LastProp lp;
    PropArgs& a = lp.args;
    memset(&a, 0, sizeof(a));
    a.feat_ring = ring.feat;
    a.feat_f32 = ring.featf;
    a.coord_f32 = ctx->coord_f32;
    a.sig1_sq = (float)((double)sigma1 * (double)sigma1);
    a.sig2_sq = (float)((double)sigma2 * (double)sigma2);
    a.coord_tab = ctx->coord_tab;
    a.lab_hi = ring.lab_hi;
    a.lab_lo = lab_lo ? ring.lab_lo : nullptr;
    a.lab16 = ring.lab16;
    for (int n = 0; n < n_ref; ++n) a.slot[n] = slots[n];
    a.sparse_mask = 0;
    if (!prob && frame_idx > 15)   // reference src/model/predict.py:59-64
        for (int n = 0; n < n_ref - kContinuousFrame; ++n) a.sparse_mask |= 1ull << n;
    a.target_slot = target_slot;
    bf16_t* const target_in_ring = ring.feat ? ring.feat + (size_t)target_slot * ctx->HWp * kC : nullptr;
    a.target_feat = ctx->fuse_push ? (const bf16_t*)ctx->fuse_push : target_in_ring;
    a.target_rows = ctx->fuse_push ? ctx->HW : ctx->HWp;
    a.target_f16 = ctx->fuse_push && ctx->fuse_f16 ? 1 : 0;
    a.n_ref = n_ref;
    a.HW = ctx->HW;
    a.HWp = ctx->HWp;
    a.Wd = ctx->cfg.feat_w;
    a.d = d;
    a.tiles_per_frame = ctx->tiles;
    // the mask-only label-mode step is prop_mask_kernel (prop_mask.h) when its limits hold; its plan knows that the last target
    // tile's waves beyond the map only stage (build_segments' last_active)
    const bool mask_form = ctx->mask_only && !prob && !lab_lo && !topk && !f32 && ctx->cfg.materialise == 0 && d <= kMaskMaxClasses;
    const int last_cols = ctx->HW - (ctx->TT - 1) * kBT;
    const int last_active = (last_cols + kColsPerWave - 1) / kColsPerWave;
    const Plan* plan = nullptr;
    int rc = get_plan(ctx, n_ref * ctx->tiles, &plan, mask_form && last_active < kWaves ? last_active : 0);
    if (!rc && mask_form && plan->max_seg_steps > kMaskMaxSteps) rc = get_plan(ctx, n_ref * ctx->tiles, &plan);   // -> prop_dense_kernel
    if (rc) return rc;
    lp.plan_last_active = plan->last_active;
    a.segs = plan->d_segs;
    a.seg_off = plan->d_seg_off;
    a.seg_first = plan->d_first;
    a.c = (float)((double)temperature * 1.4426950408889634);
    a.g1 = 1.0 / ((double)sigma1 * sigma1 * temperature);
    a.g2 = 1.0 / ((double)sigma2 * sigma2 * temperature);
    a.two_over_w = 2.0 / ctx->cfg.feat_w;
    a.gamma = 1.0 + 1.0 / ((double)ctx->cfg.feat_w * ctx->cfg.feat_w);
    lp.grid = plan->grid;
    lp.prob = prob;
    lp.lab_lo = topk ? false : lab_lo;
    lp.topk = topk;
    rc = ensure_part(ctx, (size_t)plan->n_parts * (2 + d) * kBT * sizeof(float));
    if (rc) return rc;
    a.part = ctx->part;
    a.part_rows = 2 + d;
    if (topk) {
        // ---- top-k (prop_dense.h TK 1 / 2): sizes follow the number of reference tiles NT = n_ref * tiles ----
        const int NT = n_ref * ctx->tiles;
        if (NT > 65536 || ctx->tiles > 65535) return fail(ctx, VOSPROP_E_UNSUPPORTED, "top-k: more than 65 536 reference tiles");
        const int KS = (topk + 7) / 8 * 8;
        int bits = 1;
        while ((1 << bits) < 2 * NT) ++bits;                       // (stream index << 1 | half)
        const int words = (NT + 31) / 32;
        int chunks = 256 / ctx->TT;                                 // workgroups of pass 2 per target tile: fill the chip once ...
        if (chunks > 32) chunks = 32;                               // (topk_combine2_kernel: one lane per (half, share) unit)
        // (tried: a multiple of 8 shares per target tile, so that share s of every target tile runs on XCD s - the marked tiles of a
        // target tile cluster around ITS pixels, so equal shares of distant target tiles have nothing in common: pass-2 traffic on
        // flat logits 136 -> 159 MB, L2 hit 0.46 -> 0.36, +4 us.  With 9 shares neighbouring shares of neighbouring target tiles meet
        // on an XCD, which is the better accident.)
        const int need_chunks = (NT + kTkListCap - 1) / kTkListCap; // ... and never more marked tiles per workgroup than its list holds
        if (chunks < need_chunks) chunks = need_chunks;
        if (chunks < 1) chunks = 1;
        if (need_chunks > 32) return fail(ctx, VOSPROP_E_UNSUPPORTED, "top-k: more than 32 768 reference tiles");
        const size_t cols = (size_t)ctx->TT * kBT;
        const size_t units = cols * 2 * chunks, groups = units * KS;
        if (!ctx->tk_over) {
            HIP_TRY(ctx, hipMalloc((void**)&ctx->tk_over, 3 * sizeof(unsigned)));
            HIP_TRY(ctx, hipMemsetAsync(ctx->tk_over, 0, 3 * sizeof(unsigned), s));
        }
        if (!ctx->tk_thr) {
            HIP_TRY(ctx, hipMalloc((void**)&ctx->tk_thr, cols * sizeof(float)));
            HIP_TRY(ctx, hipMalloc((void**)&ctx->tk_thr_elem, cols * sizeof(float)));
        }
        rc = ensure_buf(ctx, &ctx->tk_bitmap, &ctx->tk_bitmap_bytes, (size_t)ctx->TT * words * 4, s);
        if (!rc) rc = ensure_buf(ctx, &ctx->tk_part, &ctx->tk_part_bytes, (size_t)plan->n_parts * 2 * KS * kBT * sizeof(float), s);
        if (rc) return rc;
        if (groups > ctx->tk_dump_groups || units > ctx->tk_cnt_units) {
            HIP_TRY(ctx, hipStreamSynchronize(s));
            if (ctx->tk_dump) (void)hipFree(ctx->tk_dump);
            if (ctx->tk_dump_r) (void)hipFree(ctx->tk_dump_r);
            if (ctx->tk_cnt) (void)hipFree(ctx->tk_cnt);
            ctx->tk_dump = nullptr; ctx->tk_dump_r = nullptr; ctx->tk_cnt = nullptr;
            ctx->tk_dump_groups = ctx->tk_cnt_units = 0;
            HIP_TRY(ctx, hipMalloc((void**)&ctx->tk_dump, groups * 16 * sizeof(float)));
            HIP_TRY(ctx, hipMalloc((void**)&ctx->tk_dump_r, groups * sizeof(unsigned)));
            HIP_TRY(ctx, hipMalloc((void**)&ctx->tk_cnt, units * sizeof(unsigned)));
            ctx->tk_dump_groups = groups;
            ctx->tk_cnt_units = units;
        }
        lp.tk_ks = KS;
        lp.tk_grid2 = ctx->TT * chunks;
        lp.tk_norm = !ctx->mask_only;
        lp.tk_part = ctx->tk_part;
        lp.dense_part = ctx->part;
        lp.dense_rows = 2 + d;
        a.tk_off = plan->d_off;
        a.tk_idx_bits = bits;
        a.tk_bitmap = ctx->tk_bitmap;
        a.tk_words = words;
        a.tk_bitmap_words = ctx->TT * words;
        a.tk_chunks = chunks;
        a.tk_cap = KS;
        a.tk_dump = ctx->tk_dump;
        a.tk_dump_r = ctx->tk_dump_r;
        a.tk_cnt = ctx->tk_cnt;
        a.tk_over = ctx->tk_over;
        memset(&lp.sel, 0, sizeof(lp.sel));
        lp.sel.part = ctx->tk_part;
        lp.sel.plist_off = plan->d_off;
        lp.sel.k = topk; lp.sel.ks = KS; lp.sel.HW = ctx->HW; lp.sel.bits = bits; lp.sel.words = words;
        lp.sel.thr_grp = ctx->tk_thr; lp.sel.thr_elem = ctx->tk_thr_elem; lp.sel.bitmap = ctx->tk_bitmap;
        lp.sel.over = ctx->tk_over;
    }
    lp.materialise = ctx->cfg.materialise != 0;
    lp.no_l = mask_form && plan->max_seg_steps <= kMaskMaxSteps;      // (more classes or longer segments: prop_dense_kernel, with its denominators)
    if (lp.no_l) {
        rc = build_target_consts(ctx, sigma1, sigma2, temperature);
        if (rc) return rc;
        a.tc_b = ctx->tc_b;
        a.tc_kq = ctx->tc_kq;
    }
    if (lp.materialise) {
        if (f32 || topk) return fail(ctx, VOSPROP_E_UNSUPPORTED, "the materialised-affinity variant is dense, bf16 path only");
        const size_t need = (size_t)n_ref * ctx->tiles * ((size_t)ctx->TT * kWaves) * 64 * 16 * sizeof(bf16_t);
        if (need > ctx->smat_bytes) {
            HIP_TRY(ctx, hipStreamSynchronize(s));
            if (ctx->smat) (void)hipFree(ctx->smat);
            ctx->smat = nullptr;
            ctx->smat_bytes = 0;
            HIP_TRY(ctx, hipMalloc((void**)&ctx->smat, need));
            ctx->smat_bytes = need;
        }
        a.smat = ctx->smat;
    }
    a.tk_k = topk;
    a.tk_thr = ctx->tk_thr;
    // in-situ timing: event pairs ride on every `stride`-th eligible launch (VOSPROP_TIMING_STRIDE, default 1 = all of them)
    static const int tstride = getenv("VOSPROP_TIMING_STRIDE") ? std::max(1, atoi(getenv("VOSPROP_TIMING_STRIDE"))) : 1;
    const bool eligible = ctx->timing && !f32 && !topk && !prob && !lab_lo && !lp.materialise;   // (either NEED_L form)
    const bool timed = eligible && (ctx->tseq++ % tstride) == 0 && ctx->tev_used + 2 <= 2 * 4096;
    if (timed) {
        while (ctx->tev.size() < ctx->tev_used + 2) {
            hipEvent_t e;
            HIP_TRY(ctx, hipEventCreate(&e));
            ctx->tev.push_back(e);
        }
        launch_prop(ctx, lp, s, ctx->tev[ctx->tev_used], ctx->tev[ctx->tev_used + 1]);
        ctx->tev_used += 2;
    } else {
        launch_prop(ctx, lp, s);
    }
    HIP_TRY(ctx, hipGetLastError());
    const dim3 cgrid(ctx->HWp / 64 + (ctx->HWp % 64 ? 1 : 0));
    if (topk) {
        TopkCombineArgs ca;
        memset(&ca, 0, sizeof(ca));
        ca.thr_elem = ctx->tk_thr_elem; ca.dump = ctx->tk_dump; ca.dump_r = ctx->tk_dump_r; ca.cnt = ctx->tk_cnt;
        ca.cls_ring = ring.cls;
        ca.over = ctx->tk_over;
        ca.norm_part = lp.tk_norm ? ctx->part : nullptr;
        ca.plist_off = plan->d_off;
        ca.norm_rows = 2 + d;
        for (int n = 0; n < n_ref; ++n) ca.slot[n] = slots[n];
        ca.k = topk; ca.d = d; ca.HW = ctx->HW; ca.HWp = ctx->HWp; ca.n_ref = n_ref; ca.chunks = a.tk_chunks; ca.cap = a.tk_cap;
        ca.c = a.c;
        // VOSPROP_TK_FORCE_RADIX=1 (tests/test_gpu_topk.py): every column takes the radix selection over ALL its keys - the path a
        // column falls back to when its candidates overflow the compaction buffer (counted in vosprop_topk_overflows)
        static const int tk_force_radix = getenv("VOSPROP_TK_FORCE_RADIX") ? atoi(getenv("VOSPROP_TK_FORCE_RADIX")) : 0;
        ca.force_radix = tk_force_radix;
        hipLaunchKernelGGL(topk_combine2_kernel, dim3((ctx->HWp + kTkComWaves - 1) / kTkComWaves), dim3(kTkComWaves * 64), 0, s, ca, pred, cls, new_lab_hi,
                           new_lab_lo);
    } else {
        UpArgs up;
        memset(&up, 0, sizeof(up));
        if (ctx->fuse_mask) {
            up.mask = ctx->fuse_mask;
            up.y0 = ctx->up_tab;
            up.x0 = ctx->up_tab + ctx->cfg.feat_h + 1;
            up.H = ctx->H; up.W = ctx->W; up.Hd = ctx->cfg.feat_h; up.Wd = ctx->cfg.feat_w;
            up.sx = ctx->up_sx;
        }
        const int cp_n = ctx->fuse_push ? (int)((size_t)ctx->HW * kC * sizeof(bf16_t) / 16) : 0;
        hipLaunchKernelGGL(combine_kernel, cgrid, dim3(256), 0, s, ctx->part, plan->d_off, plan->d_list, d, ctx->HW, a.c, pred,
                           cls, new_lab_hi, new_lab_lo, prob ? 1 : 0, up, (const uint4*)ctx->fuse_push, (uint4*)target_in_ring, cp_n,
                           lp.no_l ? 1 : 0, ctx->fuse_push && ctx->fuse_f16 ? 1 : 0, new_lab16);
        // re-runs of this propagation (vosprop_time_last_propagation, debug hooks) read the target from the ring: the caller's
        // buffer is only promised until the work enqueued by this call has run
        a.target_feat = target_in_ring;
        a.target_rows = ctx->HWp;
        a.target_f16 = 0;
    }
    HIP_TRY(ctx, hipGetLastError());
    lp.valid = true;
    ctx->last = lp;
    vosprop_stats& st = ctx->stats;
    const double HW = ctx->HW;
    st.n_ref = n_ref;
    st.hw = ctx->HW;
    st.workgroups = lp.grid;
    st.tiles_per_wg = (int)(((long long)ctx->TT * n_ref * ctx->tiles + lp.grid - 1) / lp.grid);
    st.kernel_id = f32 ? VOSPROP_KERNEL_F32 : topk ? VOSPROP_KERNEL_TOPK : lp.materialise ? VOSPROP_KERNEL_MATERIALISED
                   : lp.no_l ? VOSPROP_KERNEL_MASK : VOSPROP_KERNEL_DENSE;
    // ALGORITHMIC work, counted once whatever the number of passes the implementation makes: the top-k
    // variant scores every (reference, target) pair twice today, which is its cost, not its work
    st.flops = 2.0 * n_ref * HW * HW * kC + (topk ? 2.0 * d * topk * HW : 2.0 * d * n_ref * HW * HW);
    st.bytes = n_ref * HW * kC * 2.0 + HW * kC * 2.0 + n_ref * HW + d * HW * 4.0;
    if (lp.materialise) st.bytes += 2.0 * n_ref * HW * HW * 2.0;   // the bf16 affinity written once and read once
    return VOSPROP_OK;
}

int pack_labels_from_cls(vosprop_ctx* ctx, const uint8_t* cls, bf16_t* lab_hi, bf16_t* lab16, hipStream_t s) {
    const int n = ctx->tiles * 128;
    hipLaunchKernelGGL(pack_cls_kernel, dim3((n + 255) / 256), dim3(256), 0, s, cls, lab_hi, lab16, ctx->HW, ctx->tiles);
    HIP_TRY(ctx, hipGetLastError());
    return VOSPROP_OK;
}

int pack_labels_from_f32(vosprop_ctx* ctx, const float* L, size_t ld, int d, bf16_t* lab_hi, bf16_t* lab_lo,
                         uint8_t* cls, hipStream_t s) {
    const int n = ctx->tiles * 128;
    hipLaunchKernelGGL(pack_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, s, L, ld, d, lab_hi, lab_lo, cls, ctx->HW,
                       ctx->tiles);
    HIP_TRY(ctx, hipGetLastError());
    return VOSPROP_OK;
}

size_t dtype_size(int dt) { return (dt & ~VOSPROP_LAYOUT_HWC) == VOSPROP_DT_F32 ? 4 : 2; }

// ATen nearest-neighbour source index (see aux_kernels.h upsample_kernel)
inline int nearest_src(int dst, int in_size, int out_size) {
    const float scale = (float)in_size / (float)out_size;
    int s = (int)floorf((float)dst * scale);
    return s < in_size - 1 ? s : in_size - 1;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
extern "C" {

const char* vosprop_version(void) { return "vosprop 0.2 (gfx950, bf16 MFMA + f32 MFMA parity path)"; }

void vosprop_default_config(vosprop_config* cfg, int feat_h, int feat_w) {
    memset(cfg, 0, sizeof(*cfg));
    cfg->abi_version = VOSPROP_ABI_VERSION;
    cfg->device = 0;
    cfg->feat_h = feat_h;
    cfg->feat_w = feat_w;
    cfg->channels = kC;
    cfg->ref_num = 9;
    cfg->frame_range = 40;
    cfg->sigma1 = 8.0f;
    cfg->sigma2 = 21.0f;
    cfg->temperature = 1.0f;
    cfg->probability = 0;
    cfg->topk = 0;
    cfg->precision = VOSPROP_PREC_BF16;
    cfg->ring_capacity = 0;
}

int vosprop_bias_act(void* y, const void* bias, const void* residual, long long pixels, int channels, int relu, int dtype,
                     void* stream) {
    if (!y || !bias || pixels < 0 || channels < 8 || channels % 8) return VOSPROP_E_INVALID;
    if (pixels == 0) return VOSPROP_OK;
    hipStream_t s = (hipStream_t)stream;
    const long long n = pixels * channels;
    const long long groups = dtype == VOSPROP_DT_F32 ? n / 4 : n / 8;
    long long blocks = (groups + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;   // grid-stride: 16 workgroups per CU
    const dim3 grid((unsigned)blocks), block(256);
#define VOSPROP_BA(KER, T)                                                                                                  \
    do {                                                                                                                    \
        if (relu && residual) hipLaunchKernelGGL((KER<T, true, true>), grid, block, 0, s, (T*)y, (const T*)bias, (const T*)residual, groups, channels);   \
        else if (relu) hipLaunchKernelGGL((KER<T, true, false>), grid, block, 0, s, (T*)y, (const T*)bias, (const T*)nullptr, groups, channels);          \
        else if (residual) hipLaunchKernelGGL((KER<T, false, true>), grid, block, 0, s, (T*)y, (const T*)bias, (const T*)residual, groups, channels);     \
        else hipLaunchKernelGGL((KER<T, false, false>), grid, block, 0, s, (T*)y, (const T*)bias, (const T*)nullptr, groups, channels);                   \
    } while (0)
    if (dtype == VOSPROP_DT_BF16) VOSPROP_BA(bias_act_kernel, bf16_t);
    else if (dtype == VOSPROP_DT_F16) VOSPROP_BA(bias_act_kernel, _Float16);
    else if (dtype == VOSPROP_DT_F32) {
        float* yf = (float*)y;
        const float *bf = (const float*)bias, *rf = (const float*)residual;
        if (relu && rf) hipLaunchKernelGGL((bias_act_f32_kernel<true, true>), grid, block, 0, s, yf, bf, rf, groups, channels);
        else if (relu) hipLaunchKernelGGL((bias_act_f32_kernel<true, false>), grid, block, 0, s, yf, bf, rf, groups, channels);
        else if (rf) hipLaunchKernelGGL((bias_act_f32_kernel<false, true>), grid, block, 0, s, yf, bf, rf, groups, channels);
        else hipLaunchKernelGGL((bias_act_f32_kernel<false, false>), grid, block, 0, s, yf, bf, rf, groups, channels);
    } else return VOSPROP_E_INVALID;
#undef VOSPROP_BA
    return hipGetLastError() == hipSuccess ? VOSPROP_OK : VOSPROP_E_HIP;
}

int vosprop_bias_relu_maxpool(const void* x, const void* bias, void* y, int n, int h, int w, int channels, int dtype,
                              void* stream) {
    if (!x || !bias || !y || n < 0 || h <= 0 || w <= 0 || channels < 8 || channels % 8) return VOSPROP_E_INVALID;
    if (n == 0) return VOSPROP_OK;
    hipStream_t s = (hipStream_t)stream;
    const int oh = (h - 1) / 2 + 1, ow = (w - 1) / 2 + 1;
    const int G = dtype == VOSPROP_DT_F32 ? 4 : 8;
    const long long total = (long long)n * oh * ow * (channels / G);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    const dim3 grid((unsigned)blocks), block(256);
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    if (dtype == VOSPROP_DT_BF16)
        hipLaunchKernelGGL((bias_relu_maxpool_kernel<bf16_t, Vec8<bf16_t>::type, 8>), grid, block, 0, s, (const bf16_t*)x,
                           (const bf16_t*)bias, (bf16_t*)y, n, h, w, channels, oh, ow);
    else if (dtype == VOSPROP_DT_F16)
        hipLaunchKernelGGL((bias_relu_maxpool_kernel<_Float16, Vec8<_Float16>::type, 8>), grid, block, 0, s, (const _Float16*)x,
                           (const _Float16*)bias, (_Float16*)y, n, h, w, channels, oh, ow);
    else if (dtype == VOSPROP_DT_F32)
        hipLaunchKernelGGL((bias_relu_maxpool_kernel<float, f32x4v, 4>), grid, block, 0, s, (const float*)x, (const float*)bias,
                           (float*)y, n, h, w, channels, oh, ow);
    else return VOSPROP_E_INVALID;
    return hipGetLastError() == hipSuccess ? VOSPROP_OK : VOSPROP_E_HIP;
}

int vosprop_pointwise_conv(const void* x, const void* weight, const void* bias, const void* residual, void* y, long long pixels,
                           int cin, int cout, int relu, int dtype, void* stream) {
    hipDataType dt;
    switch (dtype) {
        case VOSPROP_DT_BF16: dt = HIP_R_16BF; break;
        case VOSPROP_DT_F16: dt = HIP_R_16F; break;
        case VOSPROP_DT_F32: dt = HIP_R_32F; break;
        default: return VOSPROP_E_INVALID;
    }
    switch (pointwise_conv(x, weight, bias, residual, y, pixels, cin, cout, relu, dt, dtype, (hipStream_t)stream)) {
        case 0: return VOSPROP_OK;
        case 1: return VOSPROP_E_INVALID;
        case 3: return VOSPROP_E_UNSUPPORTED;
        default: return VOSPROP_E_HIP;
    }
}

int vosprop_set_deterministic(int on) {
    std::lock_guard<std::mutex> lock(pw_mutex());
    const int prev = pw_deterministic_flag();
    pw_deterministic_flag() = on ? 1 : 0;
    return prev;
}

/* test hook (GPU): every candidate algorithm the library returns for one pointwise-convolution problem, each timed and checked
 * against the f32 reference of csrc/pointwise.h on a zeroed and on a 0xFF-filled workspace.  rows: vosprop::PwCandidateReport
 * (index, workspace bytes, microseconds, worst err/tol clean, worst err/tol dirty, repeats, repeats_bad, name[160]).  `repeats`
 * further launches per candidate are each checked too (a racy algorithm fails only some); full != 0 checks EVERY output row
 * instead of the 1 024 sampled ones.  Returns the number of rows, or a negative VOSPROP_E_* code.  Changes no plan. */
int vosprop_debug_pointwise_candidates(const void* x, const void* weight, const void* bias, const void* residual, void* y,
                                       long long pixels, int cin, int cout, int relu, int dtype, void* stream, void* rows,
                                       int cap_rows, int repeats, int full) {
    hipDataType dt;
    switch (dtype) {
        case VOSPROP_DT_BF16: dt = HIP_R_16BF; break;
        case VOSPROP_DT_F16: dt = HIP_R_16F; break;
        case VOSPROP_DT_F32: dt = HIP_R_32F; break;
        default: return VOSPROP_E_INVALID;
    }
    int n = 0;
    const int rc = pointwise_conv(x, weight, bias, residual, y, pixels, cin, cout, relu, dt, dtype, (hipStream_t)stream,
                                  (PwCandidateReport*)rows, cap_rows, &n, repeats, full != 0);
    if (rc == 0) return n;
    return rc == 1 ? VOSPROP_E_INVALID : rc == 3 ? VOSPROP_E_UNSUPPORTED : VOSPROP_E_HIP;
}

/* test hook (no GPU needed): segment table for TT target tiles x NT reference tiles; rows of out = (workgroup, tt, r_lo, n_steps) */
int vosprop_debug_plan(int TT, int NT, int* out, int cap_rows, int last_active) {
    std::vector<std::vector<Segment>> per_wg;
    build_segments(TT, NT, per_wg, last_active);
    int n = 0;
    for (size_t b = 0; b < per_wg.size(); ++b)
        for (const Segment& sg : per_wg[b]) {
            if (out && n < cap_rows) {
                out[4 * n] = (int)b; out[4 * n + 1] = sg.tt; out[4 * n + 2] = sg.r_lo; out[4 * n + 3] = sg.n_steps;
            }
            ++n;
        }
    return n;
}

int vosprop_sample_frames(int frame_idx, int frame_range, int num_refs, int* out) {
    // reference src/model/predict.py:74-89; float64 linspace + truncation, as numpy does it
    int n = 0;
    if (frame_idx <= num_refs) {
        for (int i = 0; i < frame_idx; ++i) out[n++] = i;
        return n;
    }
    const int dense_num = kContinuousFrame - 1;
    const int sparse_num = num_refs - dense_num;
    if (sparse_num < 0) return VOSPROP_E_INVALID;   // the reference's np.linspace(.., num=-1 / -2) raises ValueError here
    const int ref_end = frame_idx - dense_num - 1;
    const int ref_start = ref_end - frame_range > 0 ? ref_end - frame_range : 0;
    if (sparse_num == 1) {
        out[n++] = ref_start;
    } else if (sparse_num > 1) {
        const double step = (double)(ref_end - ref_start) / (double)(sparse_num - 1);
        for (int k = 0; k < sparse_num; ++k)
            out[n++] = (k == sparse_num - 1) ? ref_end : (int)((double)ref_start + k * step);
    }
    for (int jj = 0; jj < dense_num; ++jj) out[n++] = frame_idx - dense_num + jj;
    return n;
}

int vosprop_create(vosprop_ctx** out, const vosprop_config* cfg) {
    if (!out || !cfg) return VOSPROP_E_INVALID;
    *out = nullptr;
    if (cfg->abi_version != VOSPROP_ABI_VERSION) return VOSPROP_E_INVALID;
    if (cfg->channels != kC) return VOSPROP_E_UNSUPPORTED;
    if (cfg->feat_h < 1 || cfg->feat_w < 1 || cfg->feat_h > VOSPROP_MAX_DIM || cfg->feat_w > VOSPROP_MAX_DIM)
        return VOSPROP_E_UNSUPPORTED;
    if (cfg->ref_num < 1 || cfg->ref_num > kMaxRef || cfg->frame_range < 0) return VOSPROP_E_INVALID;
    if (!(cfg->temperature > 0.0f) || !(cfg->sigma1 > 0.0f) || !(cfg->sigma2 > 0.0f)) return VOSPROP_E_INVALID;
    if (cfg->precision != VOSPROP_PREC_BF16 && cfg->precision != VOSPROP_PREC_F32) return VOSPROP_E_INVALID;
    if (cfg->precision == VOSPROP_PREC_F32 && cfg->topk != 0) return VOSPROP_E_UNSUPPORTED;   // top-k: bf16 path only
    if (cfg->topk < 0 || cfg->topk > kTopkMax) return VOSPROP_E_UNSUPPORTED;
    if (cfg->topk != 0 && cfg->probability) return VOSPROP_E_UNSUPPORTED;
    if (cfg->materialise && (cfg->topk != 0 || cfg->precision != VOSPROP_PREC_BF16)) return VOSPROP_E_UNSUPPORTED;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || cfg->device < 0 || cfg->device >= ndev) return VOSPROP_E_HIP;
    if (hipSetDevice(cfg->device) != hipSuccess) return VOSPROP_E_HIP;
    vosprop_ctx* ctx = new (std::nothrow) vosprop_ctx();
    if (!ctx) return VOSPROP_E_NOMEM;
    ctx->cfg = *cfg;
    ctx->HW = cfg->feat_h * cfg->feat_w;
    ctx->HWp = (ctx->HW + kTileR - 1) / kTileR * kTileR;
    ctx->tiles = ctx->HWp / kTileR;
    ctx->TT = (ctx->HW + kBT - 1) / kBT;
    int cap = cfg->ring_capacity;
    const int need = (cfg->frame_range + kContinuousFrame > cfg->ref_num ? cfg->frame_range + kContinuousFrame : cfg->ref_num) + 1;
    if (cap == 0) cap = need;
    if (cap < need) { delete ctx; return VOSPROP_E_INVALID; }   // vosprop_step reaches frame_idx - 4 - frame_range: fewer slots alias
    ctx->cfg.ring_capacity = cap;
    int rc = ring_alloc(ctx, ctx->ring, cap);
    if (!rc) rc = build_coord_table(ctx);
    if (!rc) rc = build_coord_f32(ctx);
    if (!rc && hipMalloc((void**)&ctx->pred_buf, (size_t)kMaxClasses * ctx->HW * sizeof(float)) != hipSuccess) rc = VOSPROP_E_HIP;
    if (!rc && hipMalloc((void**)&ctx->up_tab, (size_t)(cfg->feat_h + cfg->feat_w + 2) * sizeof(int)) != hipSuccess) rc = VOSPROP_E_HIP;
    if (!rc && hipMalloc((void**)&ctx->cls_tmp, (size_t)ctx->HWp) != hipSuccess) rc = VOSPROP_E_HIP;
    if (!rc && hipHostMalloc((void**)&ctx->stage_host, (size_t)ctx->HWp, hipHostMallocDefault) != hipSuccess) rc = VOSPROP_E_HIP;
    if (!rc && hipEventCreateWithFlags(&ctx->stage_ev, hipEventDisableTiming) != hipSuccess) rc = VOSPROP_E_HIP;
    if (rc) { vosprop_destroy(ctx); return rc; }
    *out = ctx;
    return VOSPROP_OK;
}

void vosprop_destroy(vosprop_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->cfg.device);
    (void)hipDeviceSynchronize();
    ring_free(ctx->ring);
    ring_free(ctx->scratch);
    if (ctx->coord_tab) (void)hipFree(ctx->coord_tab);
    if (ctx->tk_over) (void)hipFree(ctx->tk_over);
    if (ctx->tc_b) (void)hipFree(ctx->tc_b);
    if (ctx->tc_kq) (void)hipFree(ctx->tc_kq);
    if (ctx->coord_f32) (void)hipFree(ctx->coord_f32);
    if (ctx->up_tab) (void)hipFree(ctx->up_tab);
    if (ctx->part) (void)hipFree(ctx->part);
    if (ctx->smat) (void)hipFree(ctx->smat);
    for (Plan& p : ctx->plans) {
        if (p.d_off) (void)hipFree(p.d_off);
        if (p.d_list) (void)hipFree(p.d_list);
        if (p.d_segs) (void)hipFree(p.d_segs);
        if (p.d_seg_off) (void)hipFree(p.d_seg_off);
        if (p.d_first) (void)hipFree(p.d_first);
    }
    if (ctx->pred_buf) (void)hipFree(ctx->pred_buf);
    if (ctx->cls_tmp) (void)hipFree(ctx->cls_tmp);
    if (ctx->stage_host) (void)hipHostFree(ctx->stage_host);
    if (ctx->stage_ev) (void)hipEventDestroy(ctx->stage_ev);
    for (hipEvent_t e : ctx->tev) (void)hipEventDestroy(e);
    if (ctx->tk_thr) (void)hipFree(ctx->tk_thr);
    if (ctx->tk_thr_elem) (void)hipFree(ctx->tk_thr_elem);
    if (ctx->tk_bitmap) (void)hipFree(ctx->tk_bitmap);
    if (ctx->tk_part) (void)hipFree(ctx->tk_part);
    if (ctx->tk_dump) (void)hipFree(ctx->tk_dump);
    if (ctx->tk_dump_r) (void)hipFree(ctx->tk_dump_r);
    if (ctx->tk_cnt) (void)hipFree(ctx->tk_cnt);
    delete ctx;
}

const char* vosprop_last_error(const vosprop_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int vosprop_frame_index(const vosprop_ctx* ctx) { return ctx && ctx->in_video ? ctx->frame_idx : -1; }

// Stream-ordered start of a video: the labels go through a pinned staging buffer owned by the context, everything is enqueued
// on `s`, nothing waits for the device (the staging buffer is only waited for if the previous upload has not finished).
static int begin_with_lowres(vosprop_ctx* ctx, const std::vector<uint8_t>& cls, int d, int H, int W, hipStream_t s) {
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    if (ctx->stage_busy) HIP_TRY(ctx, hipEventSynchronize(ctx->stage_ev));
    memcpy(ctx->stage_host, cls.data(), cls.size());
    HIP_TRY(ctx, hipMemcpyAsync(ctx->ring.cls, ctx->stage_host, cls.size(), hipMemcpyHostToDevice, s));
    int rc = pack_labels_from_cls(ctx, ctx->ring.cls, ctx->ring.lab_hi, ctx->ring.lab16, s);
    if (rc) return rc;
    const size_t lab_slot_b = (size_t)ctx->tiles * 2 * 64 * 8 * sizeof(bf16_t);
    HIP_TRY(ctx, hipMemsetAsync(ctx->ring.lab_lo, 0, lab_slot_b, s));
    HIP_TRY(ctx, hipEventRecord(ctx->stage_ev, s));
    ctx->stage_busy = true;
    ctx->in_video = true;
    ctx->frame_idx = 0;
    ctx->d = d;
    ctx->H = H;
    ctx->W = W;
    if (H != ctx->up_H || W != ctx->up_W) {     // new output size: inverse nearest tables for combine_kernel's mask tail
        const int Hd = ctx->cfg.feat_h, Wd = ctx->cfg.feat_w;
        ctx->up_sy = (float)Hd / (float)H;
        ctx->up_sx = (float)Wd / (float)W;
        ctx->up_host.assign((size_t)Hd + Wd + 2, 0);
        int y = 0;
        for (int i = 0; i <= Hd; ++i) {
            while (y < H && nearest_src(y, Hd, H) < i) ++y;
            ctx->up_host[i] = y;
        }
        int x = 0;
        for (int i = 0; i <= Wd; ++i) {
            while (x < W && nearest_src(x, Wd, W) < i) ++x;
            ctx->up_host[Hd + 1 + i] = x;
        }
        HIP_TRY(ctx, hipMemcpyAsync(ctx->up_tab, ctx->up_host.data(), ctx->up_host.size() * sizeof(int), hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipStreamSynchronize(s));     // pageable source; only when the output size changes (once per job, normally)
        ctx->up_H = H;
        ctx->up_W = W;
    }
    return VOSPROP_OK;
}

int vosprop_begin_video_on(vosprop_ctx* ctx, const uint8_t* first_label_host, int H, int W, int* d_out, void* stream) {
    if (!ctx || !first_label_host || H < 1 || W < 1) return fail(ctx, VOSPROP_E_INVALID, "bad arguments");
    const int Hd = (int)std::ceil(H * 0.125), Wd = (int)std::ceil(W * 0.125);   // reference predict.py:109-110
    if (Hd != ctx->cfg.feat_h || Wd != ctx->cfg.feat_w)
        return fail(ctx, VOSPROP_E_INVALID, "image size does not match the engine's feature map");
    int mx = 0;
    for (size_t i = 0; i < (size_t)H * W; ++i) mx = first_label_host[i] > mx ? first_label_host[i] : mx;
    const int d = mx + 1;   // reference predict.py:113
    if (d > kMaxClasses) return fail(ctx, VOSPROP_E_UNSUPPORTED, "more than VOSPROP_MAX_CLASSES classes");
    std::vector<uint8_t> cls((size_t)ctx->HWp, 0);
    for (int y = 0; y < Hd; ++y) {   // reference get_labels, predict.py:92-96 (one-hot + nearest == nearest of indices)
        const int sy = nearest_src(y, H, Hd);
        for (int x = 0; x < Wd; ++x) cls[(size_t)y * Wd + x] = first_label_host[(size_t)sy * W + nearest_src(x, W, Wd)];
    }
    const int rc = begin_with_lowres(ctx, cls, d, H, W, (hipStream_t)stream);
    if (rc) return rc;
    if (d_out) *d_out = d;
    return VOSPROP_OK;
}

int vosprop_begin_video(vosprop_ctx* ctx, const uint8_t* first_label_host, int H, int W, int* d_out) {
    if (!ctx) return VOSPROP_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipDeviceSynchronize());   // work of the previous video may be in flight on any stream
    const int rc = vosprop_begin_video_on(ctx, first_label_host, H, W, d_out, nullptr);
    if (rc) return rc;
    HIP_TRY(ctx, hipDeviceSynchronize());
    return VOSPROP_OK;
}

int vosprop_begin_video_labels_on(vosprop_ctx* ctx, const uint8_t* cls_lowres_host, int d, int out_h, int out_w,
                                  void* stream) {
    if (!ctx || !cls_lowres_host || out_h < 1 || out_w < 1 || d < 1) return fail(ctx, VOSPROP_E_INVALID, "bad arguments");
    if (d > kMaxClasses) return fail(ctx, VOSPROP_E_UNSUPPORTED, "more than VOSPROP_MAX_CLASSES classes");
    std::vector<uint8_t> cls((size_t)ctx->HWp, 0);
    for (int i = 0; i < ctx->HW; ++i) {
        if (cls_lowres_host[i] >= d) return fail(ctx, VOSPROP_E_INVALID, "class index >= d in the label map");
        cls[(size_t)i] = cls_lowres_host[i];
    }
    return begin_with_lowres(ctx, cls, d, out_h, out_w, (hipStream_t)stream);
}

int vosprop_begin_video_labels(vosprop_ctx* ctx, const uint8_t* cls_lowres_host, int d, int out_h, int out_w) {
    if (!ctx) return VOSPROP_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    HIP_TRY(ctx, hipDeviceSynchronize());
    const int rc = vosprop_begin_video_labels_on(ctx, cls_lowres_host, d, out_h, out_w, nullptr);
    if (rc) return rc;
    HIP_TRY(ctx, hipDeviceSynchronize());
    return VOSPROP_OK;
}

int vosprop_step(vosprop_ctx* ctx, const void* feat_dev, int feat_dtype, float* pred_out_dev, uint8_t* mask_out_dev,
                 void* stream) {
    if (!ctx || !feat_dev) return fail(ctx, VOSPROP_E_INVALID, "bad arguments");
    if (!ctx->in_video) return fail(ctx, VOSPROP_E_STATE, "vosprop_step before vosprop_begin_video");
    hipStream_t s = (hipStream_t)stream;
    Ring& R = ctx->ring;
    const int f = ctx->frame_idx;
    const int slot = f % R.cap;
    const size_t lab_slot = (size_t)ctx->tiles * 2 * 64 * 8;
    // Channels-last bf16 features on the dense bf16 path are not copied into the ring up front: the propagation kernel reads the
    // target frame where the encoder left it and combine_kernel carries the copy (one launch and one dispatch gap fewer per frame)
    const bool hwc16 = feat_dtype == (VOSPROP_DT_BF16 | VOSPROP_LAYOUT_HWC) || feat_dtype == (VOSPROP_DT_F16 | VOSPROP_LAYOUT_HWC);
    const bool fuse_push = f > 0 && hwc16 && ctx->cfg.precision == VOSPROP_PREC_BF16 &&
                           ctx->cfg.topk == 0 && !ctx->cfg.materialise;
    int rc = fuse_push ? VOSPROP_OK : push_features(ctx, feat_dev, feat_dtype, R, slot, s);
    if (rc) return rc;
    if (f == 0) {   // reference inference_utils.py:33-48: frame 0 only seeds the history
        ctx->frame_idx = 1;
        return VOSPROP_OK;
    }
    int idx[kMaxRef], slots[kMaxRef];
    const int n_ref = vosprop_sample_frames(f, ctx->cfg.frame_range, ctx->cfg.ref_num, idx);
    if (n_ref < 1)
        return fail(ctx, VOSPROP_E_INVALID, "ref_num < 3 cannot sample past frame ref_num (the reference's np.linspace raises)");
    for (int n = 0; n < n_ref; ++n) slots[n] = idx[n] % R.cap;
    const bool prob = ctx->cfg.probability != 0;
    uint8_t* cls_slot = R.cls + (size_t)slot * ctx->HWp;
    // combine_kernel also writes the new label of this frame (reference inference_utils.py:67-71) into its ring slot
    const bool fuse_up = mask_out_dev && ctx->cfg.topk == 0;      // dense path: combine_kernel writes the mask itself
    ctx->fuse_mask = fuse_up ? mask_out_dev : nullptr;
    ctx->fuse_push = fuse_push ? feat_dev : nullptr;
    ctx->fuse_f16 = fuse_push && feat_dtype == (VOSPROP_DT_F16 | VOSPROP_LAYOUT_HWC);
    ctx->mask_only = pred_out_dev == nullptr;
    rc = propagate(ctx, R, slots, n_ref, f, slot, ctx->d, prob, prob, ctx->cfg.sigma1, ctx->cfg.sigma2,
                   ctx->cfg.temperature, ctx->pred_buf, cls_slot, R.lab_hi + slot * lab_slot, R.lab_lo + slot * lab_slot, s,
                   R.lab16 + slot * (lab_slot / 2));
    ctx->fuse_mask = nullptr;
    ctx->fuse_push = nullptr;
    ctx->mask_only = false;
    if (rc) return rc;
    if (pred_out_dev)
        HIP_TRY(ctx, hipMemcpyAsync(pred_out_dev, ctx->pred_buf, (size_t)ctx->d * ctx->HW * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (mask_out_dev && !fuse_up) {   // reference inference_utils.py:74-75
        hipLaunchKernelGGL(upsample_kernel, dim3((ctx->W + 255) / 256, ctx->H), dim3(256), 0, s, cls_slot, ctx->cfg.feat_h,
                           ctx->cfg.feat_w, mask_out_dev, ctx->H, ctx->W, ctx->up_sy, ctx->up_sx);
        HIP_TRY(ctx, hipGetLastError());
    }
    ctx->frame_idx = f + 1;
    return VOSPROP_OK;
}

int vosprop_predict(vosprop_ctx* ctx, const void* ref_dev, const void* target_dev, int feat_dtype,
                    const float* ref_label_dev, int T, int d, int frame_idx, int frame_range, int ref_num,
                    float temperature, float sigma1, float sigma2, int probability, float* out_dev, void* stream) {
    if (!ctx || !ref_dev || !target_dev || !ref_label_dev || !out_dev) return fail(ctx, VOSPROP_E_INVALID, "bad arguments");
    if (frame_idx < 1 || frame_idx > T) return fail(ctx, VOSPROP_E_INVALID, "frame_idx must be in [1, T]");
    if (ref_num < 1 || ref_num > kMaxRef) return fail(ctx, VOSPROP_E_INVALID, "ref_num out of range");
    if (d < 1 || d > kMaxClasses) return fail(ctx, VOSPROP_E_UNSUPPORTED, "d > VOSPROP_MAX_CLASSES");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(ctx, hipSetDevice(ctx->cfg.device));
    std::vector<int> idx((size_t)(frame_idx > ref_num ? frame_idx : ref_num) + kContinuousFrame);
    const int n_ref = vosprop_sample_frames(frame_idx, frame_range, ref_num, idx.data());
    if (n_ref < 1)
        return fail(ctx, VOSPROP_E_INVALID, "ref_num < 3 cannot sample past frame ref_num (the reference's np.linspace raises)");
    if (n_ref > kMaxRef) return fail(ctx, VOSPROP_E_INVALID, "more than VOSPROP_MAX_REF sampled frames");
    if (ctx->scratch.cap < n_ref + 1) {      // sized by what the sampler RETURNS (never by ref_num alone)
        HIP_TRY(ctx, hipDeviceSynchronize());
        ring_free(ctx->scratch);
        int rc = ring_alloc(ctx, ctx->scratch, (n_ref > ref_num ? n_ref : ref_num) + 1);
        if (rc) return rc;
    }
    Ring& R = ctx->scratch;
    const size_t esz = dtype_size(feat_dtype);
    const size_t frame_elems = (size_t)kC * ctx->HW;
    const size_t lab_slot = (size_t)ctx->tiles * 2 * 64 * 8;
    int slots[kMaxRef];
    for (int n = 0; n < n_ref; ++n) {
        slots[n] = n;
        int rc = push_features(ctx, (const unsigned char*)ref_dev + (size_t)idx[n] * frame_elems * esz, feat_dtype, R, n, s);
        if (rc) return rc;
        // ref_label (d, T, HW): class stride T*HW floats, frame idx[n]
        rc = pack_labels_from_f32(ctx, ref_label_dev + (size_t)idx[n] * ctx->HW, (size_t)T * ctx->HW, d,
                                  R.lab_hi + n * lab_slot, R.lab_lo + n * lab_slot, R.cls + (size_t)n * ctx->HWp, s);
        if (rc) return rc;
    }
    int rc = push_features(ctx, target_dev, feat_dtype, R, n_ref, s);
    if (rc) return rc;
    return propagate(ctx, R, slots, n_ref, frame_idx, n_ref, d, probability != 0, true, sigma1, sigma2, temperature,
                     out_dev, ctx->cls_tmp, nullptr, nullptr, s);
}

int vosprop_topk_overflows(vosprop_ctx* ctx, unsigned* out3, void* stream) {
    if (!ctx || !out3) return VOSPROP_E_INVALID;
    out3[0] = out3[1] = out3[2] = 0u;
    if (!ctx->tk_over) return VOSPROP_OK;      // no top-k step has run on this context
    HIP_TRY(ctx, hipMemcpyAsync(out3, ctx->tk_over, 3 * sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)stream));
    return VOSPROP_OK;
}

const char* vosprop_kernel_name(int kernel_id) {
    switch (kernel_id) {
        case VOSPROP_KERNEL_DENSE: return "prop_dense_kernel";
        case VOSPROP_KERNEL_MASK: return "prop_mask_kernel";
        case VOSPROP_KERNEL_TOPK: return "prop_dense_kernel<TK=1>+topk_select2_kernel+prop_dense_kernel<TK=2>";
        case VOSPROP_KERNEL_F32: return "prop_f32_kernel";
        case VOSPROP_KERNEL_MATERIALISED: return "prop_dense_kernel<MAT=1>+prop_dense_kernel<MAT=2>";
        default: return "?";
    }
}

int vosprop_last_stats(const vosprop_ctx* ctx, vosprop_stats* out) {
    if (!ctx || !out) return VOSPROP_E_INVALID;
    if (!ctx->last.valid) return VOSPROP_E_STATE;
    *out = ctx->stats;
    return VOSPROP_OK;
}

int vosprop_timing_begin(vosprop_ctx* ctx) {
    if (!ctx) return VOSPROP_E_INVALID;
    ctx->timing = true;
    ctx->tev_used = 0;
    ctx->tseq = 0;
    return VOSPROP_OK;
}

int vosprop_timing_read(vosprop_ctx* ctx, double* mean_us, int* launches) {
    if (!ctx || !mean_us || !launches) return fail(ctx, VOSPROP_E_INVALID, "bad arguments");
    ctx->timing = false;
    double sum = 0.0;
    int n = 0;
    for (size_t i = 0; i + 1 < ctx->tev_used; i += 2) {
        HIP_TRY(ctx, hipEventSynchronize(ctx->tev[i + 1]));
        float ms = 0.0f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->tev[i], ctx->tev[i + 1]));
        sum += (double)ms * 1000.0;
        ++n;
    }
    ctx->tev_used = 0;
    *mean_us = n ? sum / n : 0.0;
    *launches = n;
    return VOSPROP_OK;
}

int vosprop_time_last_propagation(vosprop_ctx* ctx, int iters, void* stream, double* mean_us) {
    if (!ctx || !mean_us || iters < 1) return fail(ctx, VOSPROP_E_INVALID, "bad arguments");
    if (!ctx->last.valid) return fail(ctx, VOSPROP_E_STATE, "no propagation has run on this context");
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    HIP_TRY(ctx, hipEventCreate(&e0));
    HIP_TRY(ctx, hipEventCreate(&e1));
    launch_prop(ctx, ctx->last, s);   // warm
    HIP_TRY(ctx, hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) launch_prop(ctx, ctx->last, s);
    HIP_TRY(ctx, hipEventRecord(e1, s));
    HIP_TRY(ctx, hipEventSynchronize(e1));
    float ms = 0.0f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    HIP_TRY(ctx, hipGetLastError());
    *mean_us = (double)ms * 1000.0 / iters;
    return VOSPROP_OK;
}

// Debug hook (not part of include/vosprop.h; tools/mask_stamps.py): re-run the last mask-only propagation with wall-clock stamps of
// every workgroup's phases (prop_mask.h VOSPROP_MASK_STAMP; 100 MHz s_memrealtime).  out_host: [grid][8] u64.  Returns the grid size.
int vosprop_debug_mask_stamps(vosprop_ctx* ctx, unsigned long long* out_host, int max_words) {
    if (!ctx || !ctx->last.valid || !ctx->last.no_l) return VOSPROP_E_STATE;
    LastProp lp = ctx->last;
    const size_t n = (size_t)lp.grid * 8;
    if ((size_t)max_words < n) return VOSPROP_E_INVALID;
    unsigned long long* d = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d, n * 8));
    HIP_TRY(ctx, hipMemset(d, 0, n * 8));
    for (int i = 0; i < 3; ++i) launch_prop(ctx, ctx->last, nullptr);      // warm, back to back like the timing loop
    lp.args.dbg = d;
    launch_prop(ctx, lp, nullptr);
    launch_prop(ctx, ctx->last, nullptr);
    HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, hipMemcpy(out_host, d, n * 8, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return lp.grid;
}

// Debug hook (not part of include/vosprop.h; tools/dbg_mask.py): re-run the last DENSE label-mode propagation with prop_mask_kernel
// (which = 1) or prop_dense_kernel with denominators (which = 0) and copy the per-segment partials [n_parts][2 + d][256] to the host.
// Both kernels walk the same segment table, so the two copies compare slot by slot.  Returns the number of floats.
int vosprop_debug_partials(vosprop_ctx* ctx, int which, float* out_host, int max_floats) {
    if (!ctx || !ctx->last.valid) return VOSPROP_E_STATE;
    LastProp lp = ctx->last;
    if (lp.topk || lp.prob || lp.lab_lo || lp.materialise || lp.args.feat_f32) return VOSPROP_E_UNSUPPORTED;
    const Plan* plan = nullptr;
    int rc = get_plan(ctx, lp.args.n_ref * ctx->tiles, &plan, lp.plan_last_active);
    if (rc) return rc;
    const size_t n = (size_t)plan->n_parts * lp.args.part_rows * kBT;
    if ((size_t)max_floats < n) return VOSPROP_E_INVALID;
    lp.no_l = which != 0;
    if (lp.no_l) {
        rc = build_target_consts(ctx, ctx->cfg.sigma1, ctx->cfg.sigma2, ctx->cfg.temperature);
        if (rc) return rc;
        lp.args.tc_b = ctx->tc_b;
        lp.args.tc_kq = ctx->tc_kq;
    }
    HIP_TRY(ctx, hipMemset(ctx->part, 0, n * sizeof(float)));
    launch_prop(ctx, lp, nullptr);
    HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, hipMemcpy(out_host, ctx->part, n * sizeof(float), hipMemcpyDeviceToHost));
    return (int)n;
}

}  // extern "C"
