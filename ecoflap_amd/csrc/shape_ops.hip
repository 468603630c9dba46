// shape_ops.hip — fused element-wise / normalisation ops for the synthetic shape modules'
// forward (plumbing; see include/ecoflap_shape_ops.h).  HBM/L2-bound, 16-byte vectors,
// one 64-lane wave per row for the norm (shuffle reduction, no LDS).
#include "common.h"
#include "../../include/ecoflap_shape_ops.h"

template <int DT>
__global__ __launch_bounds__(256) void t5_rmsnorm_kernel(const void* __restrict__ x,
                                                         const void* __restrict__ w,
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
#pragma unroll
        for (int i = 0; i < N; ++i) ss += f[i] * f[i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    const float inv = rsqrtf(ss / (float)d + eps);
    for (int64_t v = lane; v < nvec; v += 64) {
        float f[N], g[N];
        Vec<DT>::unpack(ld16(x, base + v), f);
        Vec<DT>::unpack(ld16(w, v), g);
#pragma unroll
        for (int i = 0; i < N; ++i) f[i] = g[i] * Vec<DT>::round(f[i] * inv);
        st16(y, base + v, Vec<DT>::pack(f));
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

extern "C" int ecoflap_t5_rmsnorm(const void* x, const void* w, void* y, int64_t rows, int64_t d,
                                  float eps, int dtype, void* stream) {
    if (dtype != ECOFLAP_F16 && dtype != ECOFLAP_BF16) return ECOFLAP_EDTYPE;
    if (rows <= 0 || d <= 0 || d % 8 != 0) return ECOFLAP_ESIZE;
    if (!x || !w || !y) return ECOFLAP_ENULL;
    if (!aligned16(x) || !aligned16(w) || !aligned16(y)) return ECOFLAP_EALIGN;
    const dim3 grid((unsigned)((rows + 3) / 4));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == ECOFLAP_F16)
        hipLaunchKernelGGL((t5_rmsnorm_kernel<ECOFLAP_F16>), grid, dim3(256), 0, s, x, w, y, rows, d, eps);
    else
        hipLaunchKernelGGL((t5_rmsnorm_kernel<ECOFLAP_BF16>), grid, dim3(256), 0, s, x, w, y, rows, d, eps);
    ECO_CHECK_LAUNCH();
    return 0;
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
