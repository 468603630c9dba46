// reduce.hip — K3+K4: fused |grad| (x) |W| per-layer reduction for gfx950.
//
// Replaces the accumulate / product / .sum() chain of
//   LAVIS/lavis/compression/pruners/layer_single_base_pruner.py:446-471 (first order),
//   :551-559 (MEZO-GradMag*) and the per-layer `.sum()` at :370.
// The reference moves every gradient to the CPU and keeps a full fp32 accumulator
// per matrix; the score is linear in the per-batch term, so this build reduces each
// (W, g) pair to one double per layer on the device and never materialises it.
//
// HBM-bound: paired coalesced 16-byte loads of W and g, fp32 element terms,
// 64-wide shuffle reduction -> LDS across the 4 waves -> one double per block,
// then a fixed-order second stage (deterministic, no atomics).
// Algorithmic bytes: (s_W + s_g) per element (modes 0/1), s per element otherwise.
#include "common.h"

#define RED_BLOCKS_PER_LAYER 64

template <int MODE>
static __device__ __forceinline__ float term(float w, float g) {
    if (MODE == ECOFLAP_RED_ABSW_ABSG) return __builtin_fabsf(w) * __builtin_fabsf(g);
    if (MODE == ECOFLAP_RED_SQW_SQG) return (w * w) * (g * g);
    if (MODE == ECOFLAP_RED_ABSG) return __builtin_fabsf(g);
    if (MODE == ECOFLAP_RED_ABSW) return __builtin_fabsf(w);
    return w * w;
}

// sum over elements [0, n) of one (w, g) pair, this block taking vectors
// blk, blk+nblk, ... ; returns the block total in thread 0.
template <int DTW, int DTG, int MODE>
static __device__ __forceinline__ double pair_block_sum(const void* w, const void* g, int64_t n,
                                                        int blk, int nblk, double* lds4) {
    constexpr bool USE_W = (MODE != ECOFLAP_RED_ABSG);
    constexpr bool USE_G = (MODE == ECOFLAP_RED_ABSW_ABSG || MODE == ECOFLAP_RED_SQW_SQG ||
                            MODE == ECOFLAP_RED_ABSG);
    // vector granule: 8 elements when either side is 16-bit (two 16-byte loads on an f32 side)
    constexpr int NW = Vec<DTW>::N, NG = Vec<DTG>::N;
    constexpr int N = USE_W ? (USE_G ? (NW > NG ? NW : NG) : NW) : NG;
    const int64_t nvec = n / N;
    float acc = 0.f;   // per-lane fp32 partial (<= a few thousand terms per lane)
    double dacc = 0.0; // flushed every 64 vectors to bound fp32 error
    int since = 0;
    for (int64_t v = (int64_t)blk * 256 + threadIdx.x; v < nvec; v += (int64_t)nblk * 256) {
        float wf[N], gf[N];
        if (USE_W) {
#pragma unroll
            for (int j = 0; j < N / NW; ++j) Vec<DTW>::unpack(ld16(w, v * (N / NW) + j), wf + j * NW);
        }
        if (USE_G) {
#pragma unroll
            for (int j = 0; j < N / NG; ++j) Vec<DTG>::unpack(ld16(g, v * (N / NG) + j), gf + j * NG);
        }
#pragma unroll
        for (int i = 0; i < N; ++i) acc += term<MODE>(USE_W ? wf[i] : 0.f, USE_G ? gf[i] : 0.f);
        if (++since == 64) { dacc += (double)acc; acc = 0.f; since = 0; }
    }
    dacc += (double)acc;
    if (blk == 0) {  // ragged tail
        const int64_t e = nvec * N + threadIdx.x;
        if (e < n)
            dacc += (double)term<MODE>(USE_W ? Vec<DTW>::load1(w, e) : 0.f,
                                       USE_G ? Vec<DTG>::load1(g, e) : 0.f);
    }
    return block_sum_256(dacc, lds4);
}

template <int DTW, int DTG, int MODE>
__global__ __launch_bounds__(256) void absprod_partial_kernel(const void* __restrict__ w,
                                                              const void* __restrict__ g,
                                                              int64_t n,
                                                              double* __restrict__ partials) {
    __shared__ double lds4[4];
    const double r = pair_block_sum<DTW, DTG, MODE>(w, g, n, blockIdx.x, gridDim.x, lds4);
    if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

// multi-tensor: blockIdx.y = layer, blockIdx.x = slice of that layer
template <int DTW, int DTG, int MODE>
__global__ __launch_bounds__(256) void absprod_partial_multi_kernel(
    const int64_t* __restrict__ table, double* __restrict__ partials) {
    __shared__ double lds4[4];
    const int layer = blockIdx.y;
    const void* w = (const void*)table[3 * layer + 0];
    const void* g = (const void*)table[3 * layer + 1];
    const int64_t n = table[3 * layer + 2];
    const double r = pair_block_sum<DTW, DTG, MODE>(w, g, n, blockIdx.x, gridDim.x, lds4);
    if (threadIdx.x == 0) partials[(int64_t)layer * gridDim.x + blockIdx.x] = r;
}

// mixed dtypes in ONE launch (BLIP-2: fp16 ViT-g + bf16 FlanT5 + fp32 Q-Former): rows of four
// words {w, g, numel, dtype_w | dtype_g << 8}; the dtype switch is uniform per workgroup
template <int MODE>
__global__ __launch_bounds__(256) void absprod_partial_mixed_kernel(
    const int64_t* __restrict__ table, double* __restrict__ partials) {
    __shared__ double lds4[4];
    const int layer = blockIdx.y;
    const void* w = (const void*)table[4 * layer + 0];
    const void* g = (const void*)table[4 * layer + 1];
    const int64_t n = table[4 * layer + 2];
    const int dtw = (int)(table[4 * layer + 3] & 0xff), dtg = (int)((table[4 * layer + 3] >> 8) & 0xff);
    double r = 0.0;
#define MIX(DTW_, DTG_) \
    if (dtw == DTW_ && dtg == DTG_) r = pair_block_sum<DTW_, DTG_, MODE>(w, g, n, blockIdx.x, gridDim.x, lds4);
    MIX(ECOFLAP_F32, ECOFLAP_F32)
    else MIX(ECOFLAP_F16, ECOFLAP_F16)
    else MIX(ECOFLAP_BF16, ECOFLAP_BF16)
    else MIX(ECOFLAP_F16, ECOFLAP_F32)
    else MIX(ECOFLAP_BF16, ECOFLAP_F32)
#undef MIX
    if (threadIdx.x == 0) partials[(int64_t)layer * gridDim.x + blockIdx.x] = r;
}

// second stage: out[layer] += sum(partials[layer, 0:count]) in a fixed order
__global__ __launch_bounds__(64) void absprod_final_kernel(const double* __restrict__ partials,
                                                           int count, double* out) {
    const int layer = blockIdx.x;
    double v = 0.0;
    for (int i = threadIdx.x; i < count; i += 64) v += partials[(int64_t)layer * count + i];
    v = wave_sum(v);
    if (threadIdx.x == 0) out[layer] += v;
}

static inline unsigned red_grid(int64_t n) {
    int64_t b = (n / 8 + 256 * 4 - 1) / (256 * 4);  // ~4 vectors per lane minimum
    if (b < 1) b = 1;
    if (b > 1024) b = 1024;
    return (unsigned)b;
}

extern "C" size_t ecoflap_absprod_reduce_workspace_bytes(int64_t n) {
    return (size_t)red_grid(n) * sizeof(double);
}
extern "C" size_t ecoflap_absprod_reduce_multi_workspace_bytes(int n_layers) {
    return (size_t)n_layers * RED_BLOCKS_PER_LAYER * sizeof(double);
}

#define RED_CASE(DTW_, DTG_, MODE_, ...)                                   \
    if (dtype_w == DTW_ && dtype_g == DTG_ && mode == MODE_) {             \
        constexpr int DTW = DTW_, DTG = DTG_, MODE = MODE_;                \
        __VA_ARGS__;                                                       \
        launched = true;                                                   \
    }
#define RED_MODES(DTW_, DTG_, ...)                         \
    RED_CASE(DTW_, DTG_, ECOFLAP_RED_ABSW_ABSG, __VA_ARGS__) \
    RED_CASE(DTW_, DTG_, ECOFLAP_RED_SQW_SQG, __VA_ARGS__)   \
    RED_CASE(DTW_, DTG_, ECOFLAP_RED_ABSG, __VA_ARGS__)      \
    RED_CASE(DTW_, DTG_, ECOFLAP_RED_ABSW, __VA_ARGS__)      \
    RED_CASE(DTW_, DTG_, ECOFLAP_RED_SQW, __VA_ARGS__)
// gradients share the parameter's dtype (torch.autograd.grad), or are fp32
#define RED_DISPATCH(...)                                  \
    RED_MODES(ECOFLAP_F32, ECOFLAP_F32, __VA_ARGS__)       \
    RED_MODES(ECOFLAP_F16, ECOFLAP_F16, __VA_ARGS__)       \
    RED_MODES(ECOFLAP_BF16, ECOFLAP_BF16, __VA_ARGS__)     \
    RED_MODES(ECOFLAP_F16, ECOFLAP_F32, __VA_ARGS__)       \
    RED_MODES(ECOFLAP_BF16, ECOFLAP_F32, __VA_ARGS__)

static int check_mode(int mode, const void* w, const void* g) {
    if (mode < 0 || mode > 4) return ECOFLAP_EMODE;
    if (mode != ECOFLAP_RED_ABSG && !w) return ECOFLAP_ENULL;
    if (mode <= ECOFLAP_RED_ABSG && !g) return ECOFLAP_ENULL;
    return 0;
}

extern "C" int ecoflap_absprod_reduce(const void* w, const void* g, int64_t n, int dtype_w,
                                      int dtype_g, int mode, double* out_accum, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    if (!dtype_ok(dtype_w) || !dtype_ok(dtype_g)) return ECOFLAP_EDTYPE;
    if (n < 0) return ECOFLAP_ESIZE;
    int rc = check_mode(mode, w, g);
    if (rc) return rc;
    if (!out_accum || !workspace) return ECOFLAP_ENULL;
    if (n == 0) return 0;
    if ((w && !aligned16(w)) || (g && !aligned16(g))) return ECOFLAP_EALIGN;
    const unsigned grid = red_grid(n);
    if (workspace_bytes < grid * sizeof(double)) return ECOFLAP_EWORKSPACE;
    if (mode >= ECOFLAP_RED_ABSW) dtype_g = dtype_w;   // g unused: fold onto the diagonal case
    if (mode == ECOFLAP_RED_ABSG) dtype_w = dtype_g;
    hipStream_t s = (hipStream_t)stream;
    double* partials = (double*)workspace;
    bool launched = false;
    RED_DISPATCH(hipLaunchKernelGGL((absprod_partial_kernel<DTW, DTG, MODE>), dim3(grid), dim3(256),
                                    0, s, w, g, n, partials));
    if (!launched) return ECOFLAP_EDTYPE;
    ECO_CHECK_LAUNCH();
    hipLaunchKernelGGL(absprod_final_kernel, dim3(1), dim3(64), 0, s, partials, (int)grid, out_accum);
    ECO_CHECK_LAUNCH();
    return 0;
}

extern "C" int ecoflap_absprod_reduce_multi(const int64_t* table, int n_layers, int64_t max_numel,
                                            int dtype_w, int dtype_g, int mode, double* out_accum,
                                            void* workspace, size_t workspace_bytes, void* stream) {
    if (!dtype_ok(dtype_w) || !dtype_ok(dtype_g)) return ECOFLAP_EDTYPE;
    if (n_layers < 0 || max_numel < 0) return ECOFLAP_ESIZE;
    if (mode < 0 || mode > 4) return ECOFLAP_EMODE;
    if (n_layers == 0) return 0;
    if (!table || !out_accum || !workspace) return ECOFLAP_ENULL;
    if (workspace_bytes < ecoflap_absprod_reduce_multi_workspace_bytes(n_layers))
        return ECOFLAP_EWORKSPACE;
    if (mode >= ECOFLAP_RED_ABSW) dtype_g = dtype_w;
    if (mode == ECOFLAP_RED_ABSG) dtype_w = dtype_g;
    hipStream_t s = (hipStream_t)stream;
    double* partials = (double*)workspace;
    const dim3 grid(RED_BLOCKS_PER_LAYER, (unsigned)n_layers);
    bool launched = false;
    RED_DISPATCH(hipLaunchKernelGGL((absprod_partial_multi_kernel<DTW, DTG, MODE>), grid, dim3(256),
                                    0, s, table, partials));
    if (!launched) return ECOFLAP_EDTYPE;
    ECO_CHECK_LAUNCH();
    hipLaunchKernelGGL(absprod_final_kernel, dim3((unsigned)n_layers), dim3(64), 0, s, partials,
                       RED_BLOCKS_PER_LAYER, out_accum);
    ECO_CHECK_LAUNCH();
    return 0;
}

// table_host: the SAME rows as `table` (device), read here to validate the dtype words
extern "C" int ecoflap_absprod_reduce_mixed(const int64_t* table, const int64_t* table_host, int n_layers,
                                            int mode, double* out_accum, void* workspace,
                                            size_t workspace_bytes, void* stream) {
    if (n_layers < 0) return ECOFLAP_ESIZE;
    if (mode < 0 || mode > 4) return ECOFLAP_EMODE;
    if (n_layers == 0) return 0;
    if (!table || !table_host || !out_accum || !workspace) return ECOFLAP_ENULL;
    if (workspace_bytes < ecoflap_absprod_reduce_multi_workspace_bytes(n_layers))
        return ECOFLAP_EWORKSPACE;
    for (int l = 0; l < n_layers; ++l) {
        const int dtw = (int)(table_host[4 * l + 3] & 0xff), dtg = (int)((table_host[4 * l + 3] >> 8) & 0xff);
        if (table_host[4 * l + 2] < 0) return ECOFLAP_ESIZE;
        const bool ok = (dtw == dtg && dtype_ok(dtw)) ||
                        (dtg == ECOFLAP_F32 && (dtw == ECOFLAP_F16 || dtw == ECOFLAP_BF16));
        if (!ok) return ECOFLAP_EDTYPE;      // (g unused / w unused modes: the caller repeats the used dtype)
    }
    hipStream_t s = (hipStream_t)stream;
    double* partials = (double*)workspace;
    const dim3 grid(RED_BLOCKS_PER_LAYER, (unsigned)n_layers);
#define MIXED_GO(MODE_) hipLaunchKernelGGL((absprod_partial_mixed_kernel<MODE_>), grid, dim3(256), 0, s, table, partials)
    switch (mode) {
        case ECOFLAP_RED_ABSW_ABSG: MIXED_GO(ECOFLAP_RED_ABSW_ABSG); break;
        case ECOFLAP_RED_SQW_SQG: MIXED_GO(ECOFLAP_RED_SQW_SQG); break;
        case ECOFLAP_RED_ABSG: MIXED_GO(ECOFLAP_RED_ABSG); break;
        case ECOFLAP_RED_ABSW: MIXED_GO(ECOFLAP_RED_ABSW); break;
        default: MIXED_GO(ECOFLAP_RED_SQW); break;
    }
#undef MIXED_GO
    ECO_CHECK_LAUNCH();
    hipLaunchKernelGGL(absprod_final_kernel, dim3((unsigned)n_layers), dim3(64), 0, s, partials,
                       RED_BLOCKS_PER_LAYER, out_accum);
    ECO_CHECK_LAUNCH();
    return 0;
}
