// shape_ops.hip — fused element-wise / normalisation ops for the synthetic shape modules'
// forward (plumbing; see include/ecoflap_shape_ops.h).  HBM/L2-bound, 16-byte vectors,
// one 64-lane wave per row for the norm (shuffle reduction, no LDS).
#include "common.h"
#include "../../include/ecoflap_shape_ops.h"

// T5's RMS norm, optionally preceded by the residual add that produces its input:
//   s = dtype(x + r)            (written to sum_out)      [if r != nullptr]
//   y = w * dtype(float(s) * rsqrt(mean(float(s)^2) + eps))
// — the bits of the separate add and norm (the sum is rounded to the storage dtype before the
// statistics are taken).  One wave per row.
template <int DT>
__global__ __launch_bounds__(256) void t5_rmsnorm_kernel(const void* __restrict__ x,
                                                         const void* __restrict__ r,
                                                         const void* __restrict__ w,
                                                         void* __restrict__ sum_out,
                                                         void* __restrict__ y, int64_t rows,
                                                         int64_t d, float eps) {
    constexpr int N = Vec<DT>::N;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t nvec = d / N;
    const int64_t base = row * nvec;
    float ss = 0.f;
    for (int64_t v = lane; v < nvec; v += 64) {
        float f[N];
        Vec<DT>::unpack(ld16(x, base + v), f);
        if (r) {
            float g[N];
            Vec<DT>::unpack(ld16(r, base + v), g);
#pragma unroll
            for (int i = 0; i < N; ++i) f[i] = Vec<DT>::round(f[i] + g[i]);
            st16(sum_out, base + v, Vec<DT>::pack(f));
        }
#pragma unroll
        for (int i = 0; i < N; ++i) ss += f[i] * f[i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    const float inv = rsqrtf(ss / (float)d + eps);
    const void* src = r ? sum_out : x;       // (this lane's own stores of a moment ago)
    for (int64_t v = lane; v < nvec; v += 64) {
        float f[N], g[N];
        Vec<DT>::unpack(ld16(src, base + v), f);
        Vec<DT>::unpack(ld16(w, v), g);
#pragma unroll
        for (int i = 0; i < N; ++i) f[i] = g[i] * Vec<DT>::round(f[i] * inv);
        st16(y, base + v, Vec<DT>::pack(f));
    }
}

// LayerNorm of a 16-bit activation with fp32 statistics and fp32 affine parameters (what
// autocast makes of nn.LayerNorm: cast up, normalise in fp32, cast down at the next Linear),
// optionally preceded by the residual add that produces its input:
//   s = dtype(x + r)            (written to sum_out)      [if r != nullptr]
//   y = dtype((float(s) - mean) * rstd * w + b)
// one wave per row, two passes over registers' worth of row (mean, then centred squares).
// rb != nullptr: r is a Linear's output WITHOUT its bias (the pinned GEMM of csrc/gemm_pinned.hip
// has no bias epilogue on gfx950) and rb that bias: r' = dtype(r + rb) first, the value the
// Linear itself would have returned from a rounded accumulator.
template <int DT>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const void* __restrict__ x,
                                                            const void* __restrict__ r,
                                                            const void* __restrict__ rb,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ b,
                                                            void* __restrict__ sum_out,
                                                            void* __restrict__ y, int64_t rows,
                                                            int64_t d, float eps) {
    constexpr int N = Vec<DT>::N;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t nvec = d / N;
    const int64_t base = row * nvec;
    float acc = 0.f;
    for (int64_t v = lane; v < nvec; v += 64) {
        float f[N];
        Vec<DT>::unpack(ld16(x, base + v), f);
        if (r) {
            float g[N];
            Vec<DT>::unpack(ld16(r, base + v), g);
            if (rb) {
                float h[N];
                Vec<DT>::unpack(ld16(rb, v), h);
#pragma unroll
                for (int i = 0; i < N; ++i) g[i] = Vec<DT>::round(g[i] + h[i]);
            }
#pragma unroll
            for (int i = 0; i < N; ++i) f[i] = Vec<DT>::round(f[i] + g[i]);
            st16(sum_out, base + v, Vec<DT>::pack(f));
        }
#pragma unroll
        for (int i = 0; i < N; ++i) acc += f[i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    const float mean = acc / (float)d;
    const void* src = r ? (const void*)sum_out : x;     // own writes: same lane re-reads them
    float var = 0.f;
    for (int64_t v = lane; v < nvec; v += 64) {
        float f[N];
        Vec<DT>::unpack(ld16(src, base + v), f);
#pragma unroll
        for (int i = 0; i < N; ++i) var += (f[i] - mean) * (f[i] - mean);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) var += __shfl_xor(var, off, 64);
    const float rstd = rsqrtf(var / (float)d + eps);
    for (int64_t v = lane; v < nvec; v += 64) {
        float f[N];
        Vec<DT>::unpack(ld16(src, base + v), f);
#pragma unroll
        for (int q = 0; q < N / 4; ++q) {
            const u32x4 w4 = ld16(w, v * (N / 4) + q), b4 = ld16(b, v * (N / 4) + q);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                f[4 * q + i] = ((f[4 * q + i] - mean) * rstd) * __uint_as_float(w4[i]) + __uint_as_float(b4[i]);
        }
        st16(y, base + v, Vec<DT>::pack(f));
    }
}

// qkv += cat(q_bias, 0, v_bias).to(dtype)   (EVA attention, eva_vit.py:123-128), in place:
// one kernel instead of zeros_like / cat / cast / add.  qkv: [rows, 3*dim]; biases fp32 [dim].
template <int DT>
__global__ __launch_bounds__(256) void qkv_bias_add_kernel(void* __restrict__ qkv,
                                                           const float* __restrict__ qb,
                                                           const float* __restrict__ vb,
                                                           int64_t rows, int64_t dim) {
    constexpr int N = Vec<DT>::N;
    const int64_t vpr = 3 * dim / N, nvec = rows * vpr;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
        const int64_t c0 = (v % vpr) * N;
        if (c0 >= dim && c0 < 2 * dim) continue;          // k: + 0 changes nothing but -0 -> +0
        const float* b = c0 < dim ? qb + c0 : vb + (c0 - 2 * dim);
        float f[N];
        Vec<DT>::unpack(ld16(qkv, v), f);
#pragma unroll
        for (int i = 0; i < N; ++i) f[i] = f[i] + Vec<DT>::round(b[i]);
        st16(qkv, v, Vec<DT>::pack(f));
    }
}

template <int DT>
__global__ __launch_bounds__(256) void gelu_mul_kernel(const void* __restrict__ a,
                                                       const void* __restrict__ b,
                                                       void* __restrict__ y, int64_t nvec) {
    constexpr int N = Vec<DT>::N;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += stride) {
        float fa[N], fb[N];
        Vec<DT>::unpack(ld16(a, v), fa);
        Vec<DT>::unpack(ld16(b, v), fb);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float g = 0.5f * fa[i] * (1.0f + erff(fa[i] * 0.70710678118654752440f));
            fa[i] = Vec<DT>::round(g) * fb[i];
        }
        st16(y, v, Vec<DT>::pack(fa));
    }
}

static int t5_rmsnorm_impl(const void* x, const void* residual, const void* w, void* sum_out, void* y,
                           int64_t rows, int64_t d, float eps, int dtype, void* stream) {
    if (dtype != ECOFLAP_F16 && dtype != ECOFLAP_BF16) return ECOFLAP_EDTYPE;
    if (rows <= 0 || d <= 0 || d % 8 != 0) return ECOFLAP_ESIZE;
    if (!x || !w || !y || (residual && !sum_out)) return ECOFLAP_ENULL;
    if (!aligned16(x) || !aligned16(w) || !aligned16(y) || (residual && (!aligned16(residual) || !aligned16(sum_out))))
        return ECOFLAP_EALIGN;
    const dim3 grid((unsigned)((rows + 3) / 4));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == ECOFLAP_F16)
        hipLaunchKernelGGL((t5_rmsnorm_kernel<ECOFLAP_F16>), grid, dim3(256), 0, s, x, residual, w, sum_out, y, rows, d, eps);
    else
        hipLaunchKernelGGL((t5_rmsnorm_kernel<ECOFLAP_BF16>), grid, dim3(256), 0, s, x, residual, w, sum_out, y, rows, d, eps);
    ECO_CHECK_LAUNCH();
    return 0;
}

extern "C" int ecoflap_t5_rmsnorm(const void* x, const void* w, void* y, int64_t rows, int64_t d,
                                  float eps, int dtype, void* stream) {
    return t5_rmsnorm_impl(x, nullptr, w, nullptr, y, rows, d, eps, dtype, stream);
}

extern "C" int ecoflap_t5_add_rmsnorm(const void* x, const void* residual, const void* w, void* sum_out,
                                      void* y, int64_t rows, int64_t d, float eps, int dtype, void* stream) {
    if (!residual) return ECOFLAP_ENULL;
    return t5_rmsnorm_impl(x, residual, w, sum_out, y, rows, d, eps, dtype, stream);
}

extern "C" int ecoflap_gelu_mul(const void* a, const void* b, void* y, int64_t n, int dtype,
                                void* stream) {
    if (dtype != ECOFLAP_F16 && dtype != ECOFLAP_BF16) return ECOFLAP_EDTYPE;
    if (n <= 0 || n % 8 != 0) return ECOFLAP_ESIZE;
    if (!a || !b || !y) return ECOFLAP_ENULL;
    if (!aligned16(a) || !aligned16(b) || !aligned16(y)) return ECOFLAP_EALIGN;
    const int64_t nvec = n / 8;
    int64_t blocks = (nvec + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == ECOFLAP_F16)
        hipLaunchKernelGGL((gelu_mul_kernel<ECOFLAP_F16>), dim3((unsigned)blocks), dim3(256), 0, s, a, b, y, nvec);
    else
        hipLaunchKernelGGL((gelu_mul_kernel<ECOFLAP_BF16>), dim3((unsigned)blocks), dim3(256), 0, s, a, b, y, nvec);
    ECO_CHECK_LAUNCH();
    return 0;
}

static int add_layernorm_impl(const void* x, const void* residual, const void* residual_bias,
                              const float* w, const float* b, void* sum_out, void* y, int64_t rows,
                              int64_t d, float eps, int dtype, void* stream) {
    if (dtype != ECOFLAP_F16 && dtype != ECOFLAP_BF16) return ECOFLAP_EDTYPE;
    if (rows < 0 || d <= 0 || (d % 8) != 0) return ECOFLAP_ESIZE;
    if (rows == 0) return 0;
    if (!x || !w || !b || !y || (residual && !sum_out) || (residual_bias && !residual)) return ECOFLAP_ENULL;
    const dim3 grid((unsigned)((rows + 3) / 4)), blk(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == ECOFLAP_F16)
        hipLaunchKernelGGL((add_layernorm_kernel<ECOFLAP_F16>), grid, blk, 0, s, x, residual, residual_bias, w, b,
                           sum_out, y, rows, d, eps);
    else
        hipLaunchKernelGGL((add_layernorm_kernel<ECOFLAP_BF16>), grid, blk, 0, s, x, residual, residual_bias, w, b,
                           sum_out, y, rows, d, eps);
    ECO_CHECK_LAUNCH();
    return 0;
}

extern "C" int ecoflap_add_layernorm(const void* x, const void* residual, const float* w,
                                     const float* b, void* sum_out, void* y, int64_t rows,
                                     int64_t d, float eps, int dtype, void* stream) {
    return add_layernorm_impl(x, residual, nullptr, w, b, sum_out, y, rows, d, eps, dtype, stream);
}

extern "C" int ecoflap_add_bias_layernorm(const void* x, const void* residual, const void* residual_bias,
                                          const float* w, const float* b, void* sum_out, void* y,
                                          int64_t rows, int64_t d, float eps, int dtype, void* stream) {
    return add_layernorm_impl(x, residual, residual_bias, w, b, sum_out, y, rows, d, eps, dtype, stream);
}

// The two other consumers of a Linear output that arrives WITHOUT its bias (see above):
//   bias_gelu:          y = dtype(gelu(dtype(a + bias)))          (EVA Mlp: act(fc1(x)), erf GELU)
//   bias_add_residual:  y = dtype(x + dtype(m + bias))            (EVA Block: x + fc2(...))
// a, m, x, y: [rows, d]; bias: [d]; all of `dtype`.  One pass each — the pass the activation /
// the residual add makes anyway.
template <int DT, bool GELU>
__global__ __launch_bounds__(256) void bias_consumer_kernel(const void* __restrict__ a,
                                                            const void* __restrict__ bias,
                                                            const void* __restrict__ x,
                                                            void* __restrict__ y, int64_t nvec,
                                                            int64_t vpr) {
    constexpr int N = Vec<DT>::N;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += stride) {
        float fa[N], fb[N];
        Vec<DT>::unpack(ld16(a, v), fa);
        Vec<DT>::unpack(ld16(bias, v % vpr), fb);
#pragma unroll
        for (int i = 0; i < N; ++i) fa[i] = Vec<DT>::round(fa[i] + fb[i]);
        if (GELU) {
#pragma unroll
            for (int i = 0; i < N; ++i)
                fa[i] = 0.5f * fa[i] * (1.0f + erff(fa[i] * 0.70710678118654752440f));
        } else {
            float fx[N];
            Vec<DT>::unpack(ld16(x, v), fx);
#pragma unroll
            for (int i = 0; i < N; ++i) fa[i] = fx[i] + fa[i];
        }
        st16(y, v, Vec<DT>::pack(fa));
    }
}

static int bias_consumer(const void* a, const void* bias, const void* x, void* y, int64_t rows, int64_t d,
                         int dtype, bool gelu, void* stream) {
    if (dtype != ECOFLAP_F16 && dtype != ECOFLAP_BF16) return ECOFLAP_EDTYPE;
    if (rows < 0 || d <= 0 || (d % 8) != 0) return ECOFLAP_ESIZE;
    if (rows == 0) return 0;
    if (!a || !bias || !y || (!gelu && !x)) return ECOFLAP_ENULL;
    if (!aligned16(a) || !aligned16(bias) || !aligned16(y) || (x && !aligned16(x))) return ECOFLAP_EALIGN;
    const int64_t vpr = d / 8, nvec = rows * vpr;
    int64_t blocks = (nvec + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    const dim3 grid((unsigned)blocks), blk(256);
    hipStream_t s = (hipStream_t)stream;
#define BC_GO(DT_, G_) hipLaunchKernelGGL((bias_consumer_kernel<DT_, G_>), grid, blk, 0, s, a, bias, x, y, nvec, vpr)
    if (dtype == ECOFLAP_F16) { if (gelu) BC_GO(ECOFLAP_F16, true); else BC_GO(ECOFLAP_F16, false); }
    else { if (gelu) BC_GO(ECOFLAP_BF16, true); else BC_GO(ECOFLAP_BF16, false); }
#undef BC_GO
    ECO_CHECK_LAUNCH();
    return 0;
}

extern "C" int ecoflap_bias_gelu(const void* a, const void* bias, void* y, int64_t rows, int64_t d,
                                 int dtype, void* stream) {
    return bias_consumer(a, bias, nullptr, y, rows, d, dtype, true, stream);
}

extern "C" int ecoflap_bias_add_residual(const void* x, const void* m, const void* bias, void* y,
                                         int64_t rows, int64_t d, int dtype, void* stream) {
    return bias_consumer(m, bias, x, y, rows, d, dtype, false, stream);
}

extern "C" int ecoflap_qkv_bias_add(void* qkv, const float* q_bias, const float* v_bias,
                                    int64_t rows, int64_t dim, int dtype, void* stream) {
    if (dtype != ECOFLAP_F16 && dtype != ECOFLAP_BF16) return ECOFLAP_EDTYPE;
    if (rows < 0 || dim <= 0 || (dim % 8) != 0) return ECOFLAP_ESIZE;
    if (rows == 0) return 0;
    if (!qkv || !q_bias || !v_bias) return ECOFLAP_ENULL;
    int64_t b = (rows * 3 * dim / 8 + 255) / 256;
    if (b > 4096) b = 4096;
    const dim3 grid((unsigned)b), blk(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == ECOFLAP_F16)
        hipLaunchKernelGGL((qkv_bias_add_kernel<ECOFLAP_F16>), grid, blk, 0, s, qkv, q_bias, v_bias, rows, dim);
    else
        hipLaunchKernelGGL((qkv_bias_add_kernel<ECOFLAP_BF16>), grid, blk, 0, s, qkv, q_bias, v_bias, rows, dim);
    ECO_CHECK_LAUNCH();
    return 0;
}

// Several small device-to-device copies in ONE launch (the loop moves states of 6-10 tensors
// between static graph buffers: one ~4 us copy kernel each was 10 % of a FlanT5 matrix's step).
// Workgroup -> (item, 16 KiB chunk); 16-byte lanes when both pointers allow, bytes otherwise.
struct CopyGroup {
    int n;
    int32_t start[ECOFLAP_COPY_MAX_ITEMS + 1];
    void* dst[ECOFLAP_COPY_MAX_ITEMS];
    const void* src[ECOFLAP_COPY_MAX_ITEMS];
    int64_t bytes[ECOFLAP_COPY_MAX_ITEMS];
};
#define COPY_CHUNK 16384
__global__ __launch_bounds__(256) void multi_copy_kernel(const CopyGroup g) {
    int it = 0;
    while (it + 1 < g.n && (int)blockIdx.x >= g.start[it + 1]) ++it;
    const int64_t off = (int64_t)(blockIdx.x - g.start[it]) * COPY_CHUNK;
    int64_t len = g.bytes[it] - off;
    if (len > COPY_CHUNK) len = COPY_CHUNK;
    char* d = (char*)g.dst[it] + off;
    const char* s = (const char*)g.src[it] + off;
    if (((((uintptr_t)d) | ((uintptr_t)s)) & 15u) == 0) {
        const int64_t nv = len >> 4;
        for (int64_t v = threadIdx.x; v < nv; v += 256) ((u32x4*)d)[v] = ((const u32x4*)s)[v];
        for (int64_t i = (nv << 4) + threadIdx.x; i < len; i += 256) d[i] = s[i];
    } else {
        for (int64_t i = threadIdx.x; i < len; i += 256) d[i] = s[i];
    }
}

extern "C" int ecoflap_multi_copy(const ecoflap_copy_item* items, int n, void* stream) {
    if (n < 0 || n > ECOFLAP_COPY_MAX_ITEMS) return ECOFLAP_ESIZE;
    if (n == 0) return 0;
    if (!items) return ECOFLAP_ENULL;
    CopyGroup g;
    g.n = 0;
    g.start[0] = 0;
    for (int i = 0; i < n; ++i) {
        if (items[i].bytes < 0) return ECOFLAP_ESIZE;
        if (items[i].bytes == 0) continue;
        if (!items[i].dst || !items[i].src) return ECOFLAP_ENULL;
        const int z = g.n++;
        g.dst[z] = items[i].dst; g.src[z] = items[i].src; g.bytes[z] = items[i].bytes;
        g.start[z + 1] = g.start[z] + (int32_t)((items[i].bytes + COPY_CHUNK - 1) / COPY_CHUNK);
    }
    if (g.n == 0) return 0;
    hipLaunchKernelGGL(multi_copy_kernel, dim3((unsigned)g.start[g.n]), dim3(256), 0,
                       (hipStream_t)stream, g);
    ECO_CHECK_LAUNCH();
    return 0;
}

// Bitwise comparison of several tensor pairs in ONE launch (the loop's exactness checks compare
// whole states: a compare + an and-reduce kernel per tensor before): any differing byte sets
// *flag (never cleared here).  Same grouping as multi_copy_kernel.
__global__ __launch_bounds__(256) void multi_compare_kernel(const CopyGroup g, int* __restrict__ flag) {
    int it = 0;
    while (it + 1 < g.n && (int)blockIdx.x >= g.start[it + 1]) ++it;
    const int64_t off = (int64_t)(blockIdx.x - g.start[it]) * COPY_CHUNK;
    int64_t len = g.bytes[it] - off;
    if (len > COPY_CHUNK) len = COPY_CHUNK;
    const char* a = (const char*)g.dst[it] + off;
    const char* b = (const char*)g.src[it] + off;
    bool diff = false;
    if (((((uintptr_t)a) | ((uintptr_t)b)) & 15u) == 0) {
        const int64_t nv = len >> 4;
        for (int64_t v = threadIdx.x; v < nv; v += 256) {
            const u32x4 x = ((const u32x4*)a)[v], y = ((const u32x4*)b)[v];
            diff |= (x[0] != y[0]) | (x[1] != y[1]) | (x[2] != y[2]) | (x[3] != y[3]);
        }
        for (int64_t i = (nv << 4) + threadIdx.x; i < len; i += 256) diff |= a[i] != b[i];
    } else {
        for (int64_t i = threadIdx.x; i < len; i += 256) diff |= a[i] != b[i];
    }
    if (__ballot(diff) != 0 && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

extern "C" int ecoflap_multi_compare(const ecoflap_copy_item* items, int n, int* mismatch_flag,
                                     void* stream) {
    if (n < 0 || n > ECOFLAP_COPY_MAX_ITEMS) return ECOFLAP_ESIZE;
    if (n == 0) return 0;
    if (!items || !mismatch_flag) return ECOFLAP_ENULL;
    CopyGroup g;
    g.n = 0;
    g.start[0] = 0;
    for (int i = 0; i < n; ++i) {
        if (items[i].bytes < 0) return ECOFLAP_ESIZE;
        if (items[i].bytes == 0) continue;
        if (!items[i].dst || !items[i].src) return ECOFLAP_ENULL;
        const int z = g.n++;
        g.dst[z] = items[i].dst; g.src[z] = items[i].src; g.bytes[z] = items[i].bytes;
        g.start[z + 1] = g.start[z] + (int32_t)((items[i].bytes + COPY_CHUNK - 1) / COPY_CHUNK);
    }
    if (g.n == 0) return 0;
    hipLaunchKernelGGL(multi_compare_kernel, dim3((unsigned)g.start[g.n]), dim3(256), 0,
                       (hipStream_t)stream, g, mismatch_flag);
    ECO_CHECK_LAUNCH();
    return 0;
}
