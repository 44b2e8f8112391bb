// Pointwise (1x1, stride 1) convolution of the encoder as ONE library GEMM with its epilogue inside:
//     y[p, co] = act( sum_ci x[p, ci] * w[co, ci]  + bias[co]  (+ residual[p, co]) )
// over a channels-last tensor viewed as (pixels, channels).  These convolutions (conv1 / conv3 / downsample[0] of every
// bottleneck and adjust_dim, reference src/model/backbone/resnet.py:66-95, src/model/vos_net.py:27-52) are HBM-bound at 480p
// (51-205 FLOP/B, ridge ~310): running them as hipBLASLt GEMMs with the bias / residual / ReLU in the epilogue writes each
// output once instead of write + read + write (convolution, then vosprop_bias_act).  hipBLASLt is column-major, so the
// row-major product is issued transposed:  D^T (cout x pixels) = W (cout x cin) * X^T (cin x pixels), i.e. op(A) = T on the
// weight, op(B) = N on the activations, bias broadcast along the rows of D^T (= output channels), C = residual with beta = 1.
#pragma once
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace vosprop {

struct PwPlan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t lw = nullptr, lx = nullptr, ly = nullptr;
    hipblasLtMatmulAlgo_t algo;
    size_t ws = 0;
    bool tuned = false;      // algo was timed on this exact problem
    bool ok = false;
};

struct PwDevice {
    hipblasLtHandle_t handle = nullptr;
    void* workspace = nullptr;
    size_t ws_bytes = 0;
    std::string ident;     // device name + library version, for the rank cache key
    // (pixels, cin, cout, dtype, epilogue, residual) -> plan
    std::map<std::tuple<long long, int, int, int, int, int>, PwPlan> plans;
};

inline std::mutex& pw_mutex() { static std::mutex m; return m; }
inline std::map<int, PwDevice>& pw_devices() { static std::map<int, PwDevice> d; return d; }

// Which of the library's ranked candidates won the timing, per exact problem, remembered across processes: the next process asks
// the library for the same ranked list (milliseconds) and takes the remembered rank instead of timing 48 candidates again (~0.5 s,
// a fifth of a DAVIS-sized job).  The algorithm itself is always a fresh answer of the library for the exact problem; the key holds
// the device name and the library version, so another stack simply searches again.  One text line per entry, appended with a single
// write (safe with one process per GPU).  $VOSPROP_CACHE_DIR or ~/.cache/vosprop; VOSPROP_PW_CACHE=0 turns it off.
struct PwRankCache {
    bool loaded = false, enabled = true;
    std::string file;
    std::map<std::string, int> rank;
};
inline PwRankCache& pw_rank_cache() { static PwRankCache c; return c; }

inline void pw_rank_cache_load(PwRankCache& c) {
    c.loaded = true;
    const char* off = getenv("VOSPROP_PW_CACHE");
    if (off && off[0] == '0') { c.enabled = false; return; }
    std::string dir;
    if (const char* d = getenv("VOSPROP_CACHE_DIR")) dir = d;
    else if (const char* h = getenv("HOME")) {
        dir = std::string(h) + "/.cache";
        (void)mkdir(dir.c_str(), 0755);
        dir += "/vosprop";
    } else { c.enabled = false; return; }
    (void)mkdir(dir.c_str(), 0755);
    c.file = dir + "/pointwise_ranks_v1.txt";
    if (FILE* f = fopen(c.file.c_str(), "r")) {
        char key[512];
        int r;
        while (fscanf(f, "%511s %d", key, &r) == 2) c.rank[key] = r;
        fclose(f);
    }
}

inline void pw_rank_cache_store(PwRankCache& c, const std::string& key, int r) {
    if (!c.enabled || c.file.empty()) return;
    c.rank[key] = r;
    const std::string line = key + " " + std::to_string(r) + "\n";
    const int fd = open(c.file.c_str(), O_WRONLY | O_CREAT | O_APPEND, 0644);
    if (fd < 0) return;
    (void)!write(fd, line.data(), line.size());
    close(fd);
}

inline bool pw_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st != hipStreamCaptureStatusNone;
}

constexpr size_t kPwWorkspace = 32u << 20;
constexpr int kPwCandidatesDefault = 48;
inline int pw_candidates() {      // VOSPROP_PW_CANDIDATES: how many of the library's ranked algorithms the first call of a layer kind times
    static const int n = [] {
        const char* e = getenv("VOSPROP_PW_CANDIDATES");
        const int v = e ? atoi(e) : kPwCandidatesDefault;
        return v < 1 ? 1 : (v > 1024 ? 1024 : v);
    }();
    return n;
}

// returns 0 on success, 1 = invalid argument, 2 = library / HIP failure, 3 = no algorithm for this shape
inline int pointwise_conv(const void* x, const void* w, const void* bias, const void* residual, void* y, long long pixels,
                          int cin, int cout, int relu, hipDataType dt, int dtype_key, hipStream_t s) {
    if (!x || !w || !y || pixels < 0 || cin <= 0 || cout <= 0) return 1;
    if (pixels == 0) return 0;
    hipPointerAttribute_t at;
    int dev = 0;
    if (hipPointerGetAttributes(&at, y) == hipSuccess) dev = at.device;
    else { (void)hipGetLastError(); if (hipGetDevice(&dev) != hipSuccess) return 2; }
    int cur = dev;
    (void)hipGetDevice(&cur);
    if (cur != dev && hipSetDevice(dev) != hipSuccess) return 2;
    struct Restore { int cur, dev; ~Restore() { if (cur != dev) (void)hipSetDevice(cur); } } restore{cur, dev};

    std::lock_guard<std::mutex> lock(pw_mutex());
    PwDevice& D = pw_devices()[dev];
    const bool capturing = pw_capturing(s);
    if (!D.handle) {
        if (hipblasLtCreate(&D.handle) != HIPBLAS_STATUS_SUCCESS) return 2;
        hipDeviceProp_t prop;
        int ver = 0;
        (void)hipblasLtGetVersion(D.handle, &ver);
        D.ident = "unknown";
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) D.ident = std::string(prop.name) + ":" + prop.gcnArchName;
        else (void)hipGetLastError();
        for (char& ch : D.ident)
            if (ch == ' ' || ch == '\t' || ch == '\n') ch = '_';
        D.ident += ":lt" + std::to_string(ver);
    }
    if (!D.workspace && !capturing) {
        if (hipMalloc(&D.workspace, kPwWorkspace) == hipSuccess) D.ws_bytes = kPwWorkspace;
        else { (void)hipGetLastError(); D.workspace = nullptr; }
    }
    const int ep_key = (bias ? 1 : 0) | (relu ? 2 : 0);
    PwPlan& P = D.plans[std::make_tuple(pixels, cin, cout, dtype_key, ep_key, residual ? 1 : 0)];
    if (!P.desc) {
        if (hipblasLtMatmulDescCreate(&P.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS) return 2;
        const hipblasOperation_t opT = HIPBLAS_OP_T, opN = HIPBLAS_OP_N;
        hipblasLtMatmulDescSetAttribute(P.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opT, sizeof(opT));
        hipblasLtMatmulDescSetAttribute(P.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opN, sizeof(opN));
        const hipblasLtEpilogue_t ep = bias ? (relu ? HIPBLASLT_EPILOGUE_RELU_BIAS : HIPBLASLT_EPILOGUE_BIAS)
                                            : (relu ? HIPBLASLT_EPILOGUE_RELU : HIPBLASLT_EPILOGUE_DEFAULT);
        hipblasLtMatmulDescSetAttribute(P.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep, sizeof(ep));
        if (bias) {
            const int32_t bt = (int32_t)dt;
            hipblasLtMatmulDescSetAttribute(P.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt));
        }
        // weight (cout, cin) row-major = column-major (cin x cout), ld cin, transposed; activations (pixels, cin) row-major =
        // column-major (cin x pixels), ld cin; output / residual (pixels, cout) row-major = column-major (cout x pixels), ld cout
        if (hipblasLtMatrixLayoutCreate(&P.lw, dt, (uint64_t)cin, (uint64_t)cout, cin) != HIPBLAS_STATUS_SUCCESS ||
            hipblasLtMatrixLayoutCreate(&P.lx, dt, (uint64_t)cin, (uint64_t)pixels, cin) != HIPBLAS_STATUS_SUCCESS ||
            hipblasLtMatrixLayoutCreate(&P.ly, dt, (uint64_t)cout, (uint64_t)pixels, cout) != HIPBLAS_STATUS_SUCCESS)
            return 2;
    }
    if (bias) hipblasLtMatmulDescSetAttribute(P.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias));
    const float alpha = 1.f, beta = residual ? 1.f : 0.f;
    const void* C = residual ? residual : y;

    auto run = [&](const hipblasLtMatmulAlgo_t& algo, size_t ws) {
        return hipblasLtMatmul(D.handle, P.desc, &alpha, w, P.lw, x, P.lx, &beta, C, P.ly, y, P.ly, &algo,
                               ws ? D.workspace : nullptr, ws, s);
    };

    if (!P.ok || (!P.tuned && !capturing && residual != y)) {
        hipblasLtMatmulPreference_t pref = nullptr;
        if (hipblasLtMatmulPreferenceCreate(&pref) != HIPBLAS_STATUS_SUCCESS) return 2;
        const uint64_t max_ws = D.ws_bytes;
        hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &max_ws, sizeof(max_ws));
        const int want = pw_candidates();
        std::vector<hipblasLtMatmulHeuristicResult_t> res(want);
        int got = 0;
        const hipblasStatus_t hs = hipblasLtMatmulAlgoGetHeuristic(D.handle, P.desc, P.lw, P.lx, P.ly, P.ly, pref, want,
                                                                   res.data(), &got);
        hipblasLtMatmulPreferenceDestroy(pref);
        if (hs != HIPBLAS_STATUS_SUCCESS || got <= 0) return 3;
        int best = -1;
        PwRankCache& RC = pw_rank_cache();
        if (!RC.loaded) pw_rank_cache_load(RC);
        const std::string ckey = D.ident + "|" + std::to_string(pixels) + "|" + std::to_string(cin) + "|" + std::to_string(cout) + "|" +
                                 std::to_string(dtype_key) + "|" + std::to_string(ep_key) + "|" + (residual ? "r" : "-") + "|" +
                                 std::to_string(want);
        int cached_rank = -1;
        if (RC.enabled) {
            auto it = RC.rank.find(ckey);
            if (it != RC.rank.end()) cached_rank = it->second;
        }
        if (capturing || residual == y) {
            // no timing inside a capture (or when a timing run would accumulate into its own input): first candidate that fits
            for (int i = 0; i < got && best < 0; ++i)
                if (res[i].state == HIPBLAS_STATUS_SUCCESS && res[i].workspaceSize <= D.ws_bytes) best = i;
        } else if (cached_rank >= 0 && cached_rank < got && res[cached_rank].state == HIPBLAS_STATUS_SUCCESS &&
                   res[cached_rank].workspaceSize <= D.ws_bytes) {
            best = cached_rank;      // timed by an earlier process on this stack
            P.tuned = true;
            if (getenv("VOSPROP_PW_VERBOSE"))
                fprintf(stderr, "[vosprop] pointwise %lld x %d -> %d: rank #%d from %s\n", pixels, cin, cout, best, RC.file.c_str());
        } else {
            // the library's ranking is a model (the winners measured on MI355X sit at ranks 2-43): time its candidates once on
            // the real operands - the output is simply rewritten
            hipEvent_t e0, e1;
            if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 2;
            float best_ms = 1e30f;
            // one timed launch per candidate after a warm one (which also loads its code object); only candidates within 15 %
            // of the best so far get a 3-launch measurement - the search costs ~2 launches per candidate instead of 4
            auto timed = [&](int i, int reps, float* ms) {
                (void)hipEventRecord(e0, s);
                bool fine = true;
                for (int r = 0; r < reps && fine; ++r) fine = run(res[i].algo, res[i].workspaceSize) == HIPBLAS_STATUS_SUCCESS;
                (void)hipEventRecord(e1, s);
                if (hipEventSynchronize(e1) != hipSuccess) { (void)hipGetLastError(); fine = false; }
                if (!fine || hipEventElapsedTime(ms, e0, e1) != hipSuccess) return false;
                *ms /= (float)reps;
                return true;
            };
            for (int i = 0; i < got; ++i) {
                if (res[i].state != HIPBLAS_STATUS_SUCCESS || res[i].workspaceSize > D.ws_bytes) continue;
                if (run(res[i].algo, res[i].workspaceSize) != HIPBLAS_STATUS_SUCCESS) continue;   // warm
                float ms = 0.f;
                if (!timed(i, 1, &ms) || ms > 1.15f * best_ms) continue;
                if (!timed(i, 3, &ms)) continue;
                if (ms < best_ms) { best_ms = ms; best = i; }
            }
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            P.tuned = best >= 0;
            if (best >= 0) pw_rank_cache_store(RC, ckey, best);
            if (getenv("VOSPROP_PW_VERBOSE"))
                fprintf(stderr, "[vosprop] pointwise %lld x %d -> %d (bias %d relu %d residual %d): %d candidates, #%d wins, %.1f us\n",
                        pixels, cin, cout, bias ? 1 : 0, relu ? 1 : 0, residual ? 1 : 0, got, best, best_ms * 1e3f);
        }
        if (best < 0) return 3;
        P.algo = res[best].algo;
        P.ws = res[best].workspaceSize;
        P.ok = true;
    }
    return run(P.algo, P.ws) == HIPBLAS_STATUS_SUCCESS ? 0 : 2;
}

}  // namespace vosprop
