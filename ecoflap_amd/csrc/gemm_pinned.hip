// gemm_pinned.hip — the forward's 16-bit Linears through hipBLASLt with the SOLUTION PINNED per
// weight shape (plumbing of the shape modules, include/ecoflap_shape_ops.h; not the pruner ABI).
//
// Why: torch hands `F.linear` to hipBLASLt's heuristic, which (a) picks Stream-K kernels for the
// loop's larger GEMMs — in their default mode not reproducible call to call (~1 call in 30 000,
// ecoflap_amd/blas_guard.py) — and (b) picks DIFFERENT kernels for M = 257 and M = 16 * 257, so
// a batch of 16 evaluations is not the 16 single evaluations bit for bit and the loop cannot
// share their suffix at batch size 1 (profiles/r03_sparsegpt.json: 49 stages not batch
// invariant, 5.2 layers/s against 14.7 at batch size 8).  Both are properties of a choice the
// library makes per call.  Here the choice is made ONCE per (N, K, dtype, bias) and kept:
//   * candidates = the library's own heuristic list for the LARGE problem (16 x the probe M);
//   * a candidate is dropped by NAME when it is a Stream-K kernel (`_SK<n>`, n > 0) or splits K
//     over workgroups (`_GSU<n>`, n > 1): their summation order depends on the grid;
//   * the rest is MEASURED: the large problem twice (same bits?) and the probe problem alone
//     against its first and last slot of the large one (same bits at every row offset?);
//   * the survivor with the lowest solution index is pinned (no timing enters the choice: two
//     ranks of one job must make the same one): every later call of that weight shape, whatever
//     M, runs that solution index (re-bound to the problem size through getAlgosFromIndex).
// With one macro tile and one K order for every M the result of a row no longer depends on how
// many rows travel with it: batch invariance by construction, no environment variable involved.
//
// Bias.  On gfx950 every bias-epilogue solution of this library for fp16 / bf16 is a Stream-K
// kernel (`TensileLibrary_*_HA_Bias_SAV_UA_*_gfx950.dat`: 455 of 455 names carry `_SK3`), so a
// Linear with a bias runs the pinned NO-bias solution with beta = 1 on an output that was first
// filled with the bias rows (one small launch): y = fp16(acc + fp32(bias)), one rounding — the
// arithmetic of the bias epilogue — for one extra write + read of y.
#include "common.h"
#include "../../include/ecoflap_shape_ops.h"

#include <hipblaslt/hipblaslt.h>
#include <hipblaslt/hipblaslt-ext.hpp>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#define ECOFLAP_ELIBRARY 999     /* hipErrorUnknown: a hipBLASLt / runtime call failed */

namespace {

struct Bound {                       // the pinned solution bound to one problem size
    hipblasLtMatmulAlgo_t algo;
    size_t workspace = 0;
};
struct Problem;
struct Plan {
    int index = -1;
    std::string name;
    int tried = 0, passed = 0;
    float best_us = 0.f;
    float default_us = 0.f;          // the library's own first choice for the large problem, same data
    std::string default_name;
    std::map<int64_t, std::pair<Problem*, Bound>> by_m;     // descriptors + binding per row count
};
using Key = std::tuple<int64_t, int64_t, int, int, int>;      // N, K, dtype, has_bias, bias_dtype

std::mutex g_mu;
std::map<Key, Plan> g_plans;
hipblasLtHandle_t g_handle = nullptr;

inline hipDataType hip_type(int dt) {
    return dt == ECOFLAP_F16 ? HIP_R_16F : (dt == ECOFLAP_BF16 ? HIP_R_16BF : HIP_R_32F);
}

struct Problem {                     // descriptors of y[M,N] = x[M,K] W[N,K]^T (+ b): column-major
    hipblasLtMatmulDesc_t desc = nullptr;          // view y^T[N,M] = W^T(op T on [K,N]) x^T[K,M]
    hipblasLtMatrixLayout_t a = nullptr, b = nullptr, c = nullptr;
    ~Problem() {
        if (a) hipblasLtMatrixLayoutDestroy(a);
        if (b) hipblasLtMatrixLayoutDestroy(b);
        if (c) hipblasLtMatrixLayoutDestroy(c);
        if (desc) hipblasLtMatmulDescDestroy(desc);
    }
    bool make(int64_t M, int64_t N, int64_t K, int dt, bool has_bias, int bias_dt, const void* bias) {
        const hipDataType t = hip_type(dt);
        if (hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS) return false;
        const int32_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
        hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta));
        hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb));
        if (has_bias) {
            // bias epilogue (the tuning sweep only — ecoflap_linear_tune: the product adds the bias
            // around the GEMM): one value per row of y^T = per output feature, added to the fp32
            // accumulator before the rounding
            const hipblasLtEpilogue_t ep = HIPBLASLT_EPILOGUE_BIAS;
            const hipDataType bt = hip_type(bias_dt);
            hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep, sizeof(ep));
            hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt));
            hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias));
        }
        return hipblasLtMatrixLayoutCreate(&a, t, (uint64_t)K, (uint64_t)N, K) == HIPBLAS_STATUS_SUCCESS &&
               hipblasLtMatrixLayoutCreate(&b, t, (uint64_t)K, (uint64_t)M, K) == HIPBLAS_STATUS_SUCCESS &&
               hipblasLtMatrixLayoutCreate(&c, t, (uint64_t)N, (uint64_t)M, N) == HIPBLAS_STATUS_SUCCESS;
    }
};

// `_<tag><number>` anywhere in a Tensile solution name -> the number (-1: tag absent)
int tagged_number(const std::string& name, const char* tag) {
    const std::string t = std::string("_") + tag;
    size_t p = 0;
    while ((p = name.find(t, p)) != std::string::npos) {
        p += t.size();
        if (p < name.size() && name[p] >= '0' && name[p] <= '9') return atoi(name.c_str() + p);
    }
    return -1;
}
bool debug_on() {
    const char* e = getenv("ECOFLAP_GEMM_DEBUG");
    return e && e[0] == '1';
}
bool grid_dependent_sum(const std::string& name) {
    return tagged_number(name, "SK") > 0 || tagged_number(name, "GSU") > 1 ||
           name.find("StreamK") != std::string::npos;
}

__global__ void fill_pattern_kernel(void* p_, int64_t n, int64_t period, uint32_t salt, int dt) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint32_t h = (uint32_t)(i % period) * 2654435761u + salt;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const float v = ((float)(h & 0xffff) / 65536.0f - 0.5f) * 0.25f;     // [-1/8, 1/8)
        if (dt == ECOFLAP_F32) {
            ((float*)p_)[i] = v;
        } else if (dt == ECOFLAP_F16) {
            const _Float16 hv = (_Float16)v;
            ((uint16_t*)p_)[i] = __builtin_bit_cast(uint16_t, hv);
        } else {
            ((uint16_t*)p_)[i] = (uint16_t)(__float_as_uint(v) >> 16);     // bf16, truncated
        }
    }
}

bool bind(int index, const Problem& pr, Bound& out) {
    std::vector<int> idx{index};
    std::vector<hipblasLtMatmulHeuristicResult_t> res;
    if (hipblaslt_ext::getAlgosFromIndex(g_handle, idx, res) != HIPBLAS_STATUS_SUCCESS || res.empty()) return false;
    const float one = 1.f, zero = 0.f;
    size_t ws = 0;
    if (hipblaslt_ext::matmulIsAlgoSupported(g_handle, pr.desc, &one, pr.a, pr.b, &zero, pr.c, pr.c, res[0].algo, ws) !=
        HIPBLAS_STATUS_SUCCESS)
        return false;
    out.algo = res[0].algo;
    out.workspace = ws;
    return true;
}

hipblasStatus_t run(const Problem& pr, const Bound& b, const void* x, const void* w, void* y, void* ws,
                    size_t ws_bytes, hipStream_t s, float beta = 0.f) {
    const float one = 1.f;
    if (b.workspace > ws_bytes) return HIPBLAS_STATUS_ALLOC_FAILED;
    return hipblasLtMatmul(g_handle, pr.desc, &one, w, pr.a, x, pr.b, &beta, y, pr.c, y, pr.c, &b.algo, ws, ws_bytes, s);
}

// y[m, :] = bias[:] for every row (16-byte vectors; N % 8 == 0)
__global__ __launch_bounds__(256) void bias_fill_kernel(u32x4* __restrict__ y, const u32x4* __restrict__ bias,
                                                        int64_t nvec_total, int64_t nvec_row) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec_total; v += stride)
        __builtin_nontemporal_store(bias[v % nvec_row], y + v);
}

}  // namespace

extern "C" int ecoflap_linear_pinned_plan(int64_t m_probe, int64_t N, int64_t K, int dtype, int has_bias,
                                          int bias_dtype, int* solution_index, int* tried, int* passed,
                                          float* best_us, float* default_us, char* name_out, int name_len) {
    if (!dtype_ok(dtype) || m_probe <= 0 || N <= 0 || K <= 0) return ECOFLAP_EDTYPE;
    std::lock_guard<std::mutex> lock(g_mu);
    (void)has_bias; (void)bias_dtype;
    const Key key{N, K, dtype, 0, 0};          // one plan per weight shape: the bias is not the GEMM's business
    auto found = g_plans.find(key);
    if (found == g_plans.end()) {
        if (!g_handle && hipblasLtCreate(&g_handle) != HIPBLAS_STATUS_SUCCESS) return ECOFLAP_ELIBRARY;
        constexpr int SLOTS = 16;
        const int64_t Mb = SLOTS * m_probe;
        const size_t es = dtype == ECOFLAP_F32 ? 4 : 2, ws_bytes = (size_t)64 << 20;
        char *x = nullptr, *w = nullptr, *bias = nullptr, *yb = nullptr, *yb2 = nullptr, *ya = nullptr;
        void* ws = nullptr;
        bool ok = hipMalloc(&x, Mb * K * es) == hipSuccess && hipMalloc(&w, N * K * es) == hipSuccess &&
                  hipMalloc(&bias, N * 4) == hipSuccess && hipMalloc(&yb, Mb * N * es) == hipSuccess &&
                  hipMalloc(&yb2, Mb * N * es) == hipSuccess && hipMalloc(&ya, m_probe * N * es) == hipSuccess &&
                  hipMalloc(&ws, ws_bytes) == hipSuccess;
        Plan plan;
        if (ok) {
            hipStream_t s = nullptr;
            // every slot of the large input is a copy of the probe input (period = its size)
            hipLaunchKernelGGL(fill_pattern_kernel, dim3(1024), dim3(256), 0, s, x, Mb * K, m_probe * K, 17u, dtype);
            hipLaunchKernelGGL(fill_pattern_kernel, dim3(1024), dim3(256), 0, s, w, N * K, N * K, 91u, dtype);
            hipLaunchKernelGGL(fill_pattern_kernel, dim3(64), dim3(256), 0, s, bias, N, N, 5u, dtype);
            Problem big, small;
            ok = big.make(Mb, N, K, dtype, false, 0, nullptr) && small.make(m_probe, N, K, dtype, false, 0, nullptr);
            // candidates: the library's heuristic list for the LARGE problem first (its own ranking),
            // then every solution it has for these types (the heuristic list is short and, for the
            // shapes that matter, mostly Stream-K)
            std::vector<hipblasLtMatmulHeuristicResult_t> cand(64);
            int n_cand = 0;
            if (ok) {
                hipblasLtMatmulPreference_t pref;
                hipblasLtMatmulPreferenceCreate(&pref);
                hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws_bytes, sizeof(ws_bytes));
                if (hipblasLtMatmulAlgoGetHeuristic(g_handle, big.desc, big.a, big.b, big.c, big.c, pref, (int)cand.size(),
                                                    cand.data(), &n_cand) != HIPBLAS_STATUS_SUCCESS)
                    n_cand = 0;
                hipblasLtMatmulPreferenceDestroy(pref);
                cand.resize((size_t)n_cand);
                std::vector<hipblasLtMatmulHeuristicResult_t> all;
                const hipDataType t = hip_type(dtype);
                if (hipblaslt_ext::getAllAlgos(g_handle, hipblaslt_ext::GemmType::HIPBLASLT_GEMM, HIPBLAS_OP_T, HIPBLAS_OP_N,
                                               t, t, t, t, HIPBLAS_COMPUTE_32F, all) == HIPBLAS_STATUS_SUCCESS)
                    cand.insert(cand.end(), all.begin(), all.end());
                if (debug_on())
                    fprintf(stderr, "[gemm_pinned] %ldx%ld dt %d bias %d: %d heuristic + %zu library candidates\n",
                            (long)N, (long)K, dtype, has_bias, n_cand, all.size());
            }
            std::vector<char> h_alone((size_t)m_probe * N * es), h_big((size_t)m_probe * N * es);
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            // the library's own first choice (what the framework's GEMM call would run), timed on
            // the same data: the caller decides whether the pinned solution is worth its price
            if (ok && n_cand > 0 && cand[0].state == HIPBLAS_STATUS_SUCCESS) {
                Bound d;
                d.algo = cand[0].algo;
                d.workspace = cand[0].workspaceSize;
                plan.default_name = hipblaslt_ext::getSolutionNameFromAlgo(g_handle, cand[0].algo);
                if (run(big, d, x, w, yb, ws, ws_bytes, s) == HIPBLAS_STATUS_SUCCESS) {
                    (void)hipEventRecord(e0, s);
                    for (int r = 0; r < 3; ++r) run(big, d, x, w, yb, ws, ws_bytes, s);
                    (void)hipEventRecord(e1, s);
                    if (hipEventSynchronize(e1) == hipSuccess) {
                        float ms = 0.f;
                        (void)hipEventElapsedTime(&ms, e0, e1);
                        plan.default_us = ms * 1e3f / 3.f;
                    }
                }
                if (debug_on())
                    fprintf(stderr, "[gemm_pinned]   library's first choice: %7.1f us  %.120s\n", plan.default_us,
                            plan.default_name.c_str());
            }
            // stage 1: by name and by support, timed once on the large problem
            struct Timed { int index; std::string name; float us; Bound big, small; };
            std::vector<Timed> timed;
            std::vector<int> seen;
            for (size_t c = 0; ok && c < cand.size() && timed.size() < 400; ++c) {
                if (cand[c].state != HIPBLAS_STATUS_SUCCESS) continue;
                const int index = hipblaslt_ext::getIndexFromAlgo(cand[c].algo);
                if (index < 0 || std::find(seen.begin(), seen.end(), index) != seen.end()) continue;
                seen.push_back(index);
                const std::string name = hipblaslt_ext::getSolutionNameFromAlgo(g_handle, cand[c].algo);
                ++plan.tried;
                if (grid_dependent_sum(name)) {
                    if (debug_on() && (int)c < n_cand) fprintf(stderr, "[gemm_pinned]   skip (grid-dependent sum) %d %s\n", index, name.c_str());
                    continue;
                }
                Timed t;
                t.index = index;
                t.name = name;
                if (!bind(index, big, t.big) || !bind(index, small, t.small)) continue;
                if (run(big, t.big, x, w, yb, ws, ws_bytes, s) != HIPBLAS_STATUS_SUCCESS) continue;     // warm
                (void)hipEventRecord(e0, s);
                if (run(big, t.big, x, w, yb, ws, ws_bytes, s) != HIPBLAS_STATUS_SUCCESS) continue;
                (void)hipEventRecord(e1, s);
                if (hipEventSynchronize(e1) != hipSuccess) { ok = false; break; }
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                t.us = ms * 1e3f;
                timed.push_back(t);
            }
            // NO timing enters the choice (two processes of one job — data-parallel ranks, a run
            // resumed from its stage-1 checkpoint — must pin the same solution, and a single-shot
            // timing ranks differently from process to process): the survivors are verified in
            // ASCENDING SOLUTION INDEX and the first one that measures batch invariant and
            // repeatable is pinned.  The timings above go into the report only.
            std::sort(timed.begin(), timed.end(), [](const Timed& a, const Timed& b) { return a.index < b.index; });
            if (debug_on())
                fprintf(stderr, "[gemm_pinned]   %zu of %d distinct solutions are name-clean and support both sizes\n",
                        timed.size(), plan.tried);
            // stage 2: MEASURED — batch invariant (the probe problem alone == its first and its last
            // slot of the 16-slot problem) and repeatable (two runs of the large problem)?
            const Timed* pick = nullptr;
            for (size_t c = 0; ok && c < timed.size() && c < 64 && !pick; ++c) {
                const Timed& t = timed[c];
                if (run(big, t.big, x, w, yb, ws, ws_bytes, s) != HIPBLAS_STATUS_SUCCESS) continue;
                if (run(big, t.big, x, w, yb2, ws, ws_bytes, s) != HIPBLAS_STATUS_SUCCESS) continue;
                if (run(small, t.small, x, w, ya, ws, ws_bytes, s) != HIPBLAS_STATUS_SUCCESS) continue;
                if (hipStreamSynchronize(s) != hipSuccess) { ok = false; break; }
                bool same = true;
                (void)hipMemcpy(h_alone.data(), ya, h_alone.size(), hipMemcpyDeviceToHost);
                for (int slot : {0, SLOTS - 1}) {
                    (void)hipMemcpy(h_big.data(), yb + (size_t)slot * m_probe * N * es, h_big.size(), hipMemcpyDeviceToHost);
                    same = same && memcmp(h_big.data(), h_alone.data(), h_big.size()) == 0;
                    (void)hipMemcpy(h_big.data(), yb2 + (size_t)slot * m_probe * N * es, h_big.size(), hipMemcpyDeviceToHost);
                    same = same && memcmp(h_big.data(), h_alone.data(), h_big.size()) == 0;
                }
                if (debug_on())
                    fprintf(stderr, "[gemm_pinned]   %s %7.1f us  %d %s\n", same ? "PASS" : "fail", t.us, t.index, t.name.c_str());
                if (same) {
                    ++plan.passed;
                    pick = &t;
                }
            }
            if (pick) {
                plan.index = pick->index;
                plan.name = pick->name;
                plan.best_us = pick->us;
                if (debug_on()) fprintf(stderr, "[gemm_pinned]   pinned: %d\n", plan.index);
            }
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
        }
        for (void* p : {(void*)x, (void*)w, (void*)bias, (void*)yb, (void*)yb2, (void*)ya, ws})
            if (p) (void)hipFree(p);
        if (!ok) return ECOFLAP_ELIBRARY;
        found = g_plans.emplace(key, plan).first;
    }
    const Plan& p = found->second;
    if (solution_index) *solution_index = p.index;
    if (tried) *tried = p.tried;
    if (passed) *passed = p.passed;
    if (best_us) *best_us = p.best_us;
    if (default_us) *default_us = p.default_us;
    if (name_out && name_len > 0) {
        strncpy(name_out, p.name.c_str(), (size_t)name_len - 1);
        name_out[name_len - 1] = 0;
    }
    return p.index >= 0 ? 0 : ECOFLAP_ESIZE;      // ESIZE: no candidate survived — the caller keeps torch's GEMM
}

extern "C" int ecoflap_linear_pinned(const void* x, const void* w, const void* bias, void* y, int64_t M,
                                     int64_t N, int64_t K, int dtype, int bias_dtype, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    if (M <= 0) return 0;
    if (!x || !w || !y) return ECOFLAP_ENULL;
    const int per_vec = dtype == ECOFLAP_F32 ? 4 : 8;
    if (bias && (bias_dtype != dtype || N % per_vec != 0)) return ECOFLAP_EDTYPE;   // the bias rows are y's own dtype
    std::lock_guard<std::mutex> lock(g_mu);
    const Key key{N, K, dtype, 0, 0};
    auto it = g_plans.find(key);
    if (it == g_plans.end() || it->second.index < 0) return ECOFLAP_EMODE;    // plan first (outside capture)
    auto bm = it->second.by_m.find(M);
    if (bm == it->second.by_m.end()) {
        Problem* pr = new Problem();        // binding needs no device work: safe under stream capture
        Bound nb;
        if (!pr->make(M, N, K, dtype, false, 0, nullptr) || !bind(it->second.index, *pr, nb)) {
            delete pr;
            return ECOFLAP_ESIZE;
        }
        bm = it->second.by_m.emplace(M, std::make_pair(pr, nb)).first;
    }
    hipStream_t s = (hipStream_t)stream;
    if (bias) {
        const int64_t nvec_row = N / per_vec, nvec = M * nvec_row;
        int64_t blocks = (nvec + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(bias_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (u32x4*)y, (const u32x4*)bias, nvec,
                           nvec_row);
    }
    const hipblasStatus_t st = run(*bm->second.first, bm->second.second, x, w, y, workspace, workspace_bytes, s,
                                   bias ? 1.f : 0.f);
    if (st == HIPBLAS_STATUS_ALLOC_FAILED) return ECOFLAP_EWORKSPACE;
    return st == HIPBLAS_STATUS_SUCCESS ? 0 : ECOFLAP_ELIBRARY;
}


// ---- tuning (tools/tune_gemm.py): every solution of the library for this problem, timed ----------
// For one weight shape: all candidates (heuristic list + getAllAlgos) that support both the
// 16-slot problem and the probe problem, each timed on both (3 runs after a warm-up), then the
// `top` fastest by t(16 m) + t(m) checked for repeatability and batch invariance bit for bit
// (with the bias epilogue when has_bias).  No name filter: under TENSILE_STREAMK_DATA_PARALLEL=1
// (ecoflap_amd/blas_guard.py) the Stream-K kernels hand every workgroup whole tiles, and whether
// that holds for a candidate is exactly what is measured.  Fills, fastest first, up to `top` rows of
// (index, us at 16 m, us at m, flags: 1 = repeatable, 2 = batch invariant), names packed at
// name_len bytes each.  -> number of rows in *n_out.
extern "C" int ecoflap_linear_tune(int64_t m_probe, int64_t N, int64_t K, int dtype, int has_bias, int bias_dtype,
                                   int top, int* index_out, float* us_big_out, float* us_small_out, int* flags_out,
                                   char* names_out, int name_len, int* n_out, int* n_candidates) {
    if (!dtype_ok(dtype) || m_probe <= 0 || N <= 0 || K <= 0 || top <= 0) return ECOFLAP_EDTYPE;
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_handle && hipblasLtCreate(&g_handle) != HIPBLAS_STATUS_SUCCESS) return ECOFLAP_ELIBRARY;
    constexpr int SLOTS = 16;
    const int64_t Mb = SLOTS * m_probe;
    const size_t es = dtype == ECOFLAP_F32 ? 4 : 2, ws_bytes = (size_t)64 << 20;
    char *x = nullptr, *w = nullptr, *bias = nullptr, *yb = nullptr, *yb2 = nullptr, *ya = nullptr;
    void* ws = nullptr;
    bool ok = hipMalloc(&x, Mb * K * es) == hipSuccess && hipMalloc(&w, N * K * es) == hipSuccess &&
              hipMalloc(&bias, N * 4) == hipSuccess && hipMalloc(&yb, Mb * N * es) == hipSuccess &&
              hipMalloc(&yb2, Mb * N * es) == hipSuccess && hipMalloc(&ya, m_probe * N * es) == hipSuccess &&
              hipMalloc(&ws, ws_bytes) == hipSuccess;
    int rows = 0;
    if (ok) {
        hipStream_t s = nullptr;
        hipLaunchKernelGGL(fill_pattern_kernel, dim3(1024), dim3(256), 0, s, x, Mb * K, m_probe * K, 17u, dtype);
        hipLaunchKernelGGL(fill_pattern_kernel, dim3(1024), dim3(256), 0, s, w, N * K, N * K, 91u, dtype);
        hipLaunchKernelGGL(fill_pattern_kernel, dim3(64), dim3(256), 0, s, bias, N, N, 5u, has_bias ? bias_dtype : dtype);
        Problem big, small;
        ok = big.make(Mb, N, K, dtype, has_bias != 0, bias_dtype, bias) &&
             small.make(m_probe, N, K, dtype, has_bias != 0, bias_dtype, bias);
        std::vector<hipblasLtMatmulHeuristicResult_t> cand(64);
        if (ok) {
            int n_cand = 0;
            hipblasLtMatmulPreference_t pref;
            hipblasLtMatmulPreferenceCreate(&pref);
            hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws_bytes, sizeof(ws_bytes));
            if (hipblasLtMatmulAlgoGetHeuristic(g_handle, big.desc, big.a, big.b, big.c, big.c, pref, (int)cand.size(),
                                                cand.data(), &n_cand) != HIPBLAS_STATUS_SUCCESS)
                n_cand = 0;
            hipblasLtMatmulPreferenceDestroy(pref);
            cand.resize((size_t)n_cand);
            std::vector<hipblasLtMatmulHeuristicResult_t> all;
            const hipDataType t = hip_type(dtype);
            if (hipblaslt_ext::getAllAlgos(g_handle, hipblaslt_ext::GemmType::HIPBLASLT_GEMM, HIPBLAS_OP_T, HIPBLAS_OP_N,
                                           t, t, t, t, HIPBLAS_COMPUTE_32F, all) == HIPBLAS_STATUS_SUCCESS)
                cand.insert(cand.end(), all.begin(), all.end());
        }
        struct Timed { int index; std::string name; float us_big, us_small; Bound big, small; int flags; };
        std::vector<Timed> timed;
        std::vector<int> seen;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        auto time3 = [&](const Problem& pr, const Bound& b, void* y, float& us) {
            if (run(pr, b, x, w, y, ws, ws_bytes, s) != HIPBLAS_STATUS_SUCCESS) return false;
            (void)hipEventRecord(e0, s);
            for (int r = 0; r < 3; ++r)
                if (run(pr, b, x, w, y, ws, ws_bytes, s) != HIPBLAS_STATUS_SUCCESS) return false;
            (void)hipEventRecord(e1, s);
            if (hipEventSynchronize(e1) != hipSuccess) return false;
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            us = ms * 1e3f / 3.f;
            return true;
        };
        int first_index = -1;
        for (size_t c = 0; ok && c < cand.size(); ++c) {
            if (cand[c].state != HIPBLAS_STATUS_SUCCESS) continue;
            const int index = hipblaslt_ext::getIndexFromAlgo(cand[c].algo);
            if (index < 0 || std::find(seen.begin(), seen.end(), index) != seen.end()) continue;
            seen.push_back(index);
            if (first_index < 0) first_index = index;
            Timed t;
            t.index = index;
            t.flags = index == first_index ? 4 : 0;                   // 4: the heuristic's own first choice
            if (!bind(index, big, t.big) || !bind(index, small, t.small)) continue;
            if (!time3(big, t.big, yb, t.us_big) || !time3(small, t.small, ya, t.us_small)) {
                (void)hipGetLastError();
                continue;
            }
            t.name = hipblaslt_ext::getSolutionNameFromAlgo(g_handle, cand[c].algo);
            timed.push_back(t);
        }
        if (n_candidates) *n_candidates = (int)timed.size();
        std::sort(timed.begin(), timed.end(), [](const Timed& a, const Timed& b) {
            return a.us_big + a.us_small < b.us_big + b.us_small;
        });
        // the heuristic's first choice always makes the list (as its last row if it is not among the fastest)
        std::vector<Timed> pick;
        for (size_t c = 0; c < timed.size() && (int)pick.size() < top - 1; ++c) pick.push_back(timed[c]);
        bool has_first = false;
        for (const Timed& t : pick) has_first = has_first || (t.flags & 4);
        for (size_t c = pick.size(); c < timed.size() && (int)pick.size() < top; ++c)
            if (has_first || (timed[c].flags & 4)) pick.push_back(timed[c]);
        std::vector<char> h_alone((size_t)m_probe * N * es), h_big((size_t)m_probe * N * es);
        for (Timed& t : pick) {
            if (!ok) break;
            if (run(big, t.big, x, w, yb, ws, ws_bytes, s) != HIPBLAS_STATUS_SUCCESS) continue;
            if (run(big, t.big, x, w, yb2, ws, ws_bytes, s) != HIPBLAS_STATUS_SUCCESS) continue;
            if (run(small, t.small, x, w, ya, ws, ws_bytes, s) != HIPBLAS_STATUS_SUCCESS) continue;
            if (hipStreamSynchronize(s) != hipSuccess) { ok = false; break; }
            bool invariant = true, repeatable = true;
            (void)hipMemcpy(h_alone.data(), ya, h_alone.size(), hipMemcpyDeviceToHost);
            for (int slot : {0, 7, SLOTS - 1}) {
                (void)hipMemcpy(h_big.data(), yb + (size_t)slot * m_probe * N * es, h_big.size(), hipMemcpyDeviceToHost);
                invariant = invariant && memcmp(h_big.data(), h_alone.data(), h_big.size()) == 0;
                std::vector<char> again(h_big.size());
                (void)hipMemcpy(again.data(), yb2 + (size_t)slot * m_probe * N * es, again.size(), hipMemcpyDeviceToHost);
                repeatable = repeatable && memcmp(h_big.data(), again.data(), again.size()) == 0;
            }
            t.flags |= (repeatable ? 1 : 0) | (invariant ? 2 : 0);
            index_out[rows] = t.index;
            us_big_out[rows] = t.us_big;
            us_small_out[rows] = t.us_small;
            flags_out[rows] = t.flags;
            if (names_out && name_len > 0) {
                strncpy(names_out + (size_t)rows * name_len, t.name.c_str(), (size_t)name_len - 1);
                names_out[(size_t)rows * name_len + name_len - 1] = 0;
            }
            ++rows;
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    for (void* p : {(void*)x, (void*)w, (void*)bias, (void*)yb, (void*)yb2, (void*)ya, ws})
        if (p) (void)hipFree(p);
    if (n_out) *n_out = rows;
    return ok ? 0 : ECOFLAP_ELIBRARY;
}


// "<hipblasLtGetVersion>-<git revision>" of the hipBLASLt that serves this process: solution
// indices mean something only together with it.
extern "C" int ecoflap_linear_library_version(char* out, int len) {
    if (!out || len <= 0) return ECOFLAP_ENULL;
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_handle && hipblasLtCreate(&g_handle) != HIPBLAS_STATUS_SUCCESS) return ECOFLAP_ELIBRARY;
    int v = 0;
    char rev[128] = {0};
    (void)hipblasLtGetVersion(g_handle, &v);
    (void)hipblasLtGetGitRevision(g_handle, rev);
    snprintf(out, (size_t)len, "%d-%.100s", v, rev);
    return 0;
}
