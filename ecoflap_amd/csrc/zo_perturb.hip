// zo_perturb.hip — K1: zeroth-order weight perturbation for gfx950.
//
// Replaces LayerSparsity.zo_perturb_parameters
//   (LAVIS/lavis/compression/pruners/layer_single_base_pruner.py:473-486) and the
//   +1 / -2 / +1 call triple around the two loss evaluations (:530-539).
//
// HBM-bound elementwise read-modify-write: one 16-byte vector per lane per
// iteration (1 KiB per wave instruction), z produced in registers by
// Philox4x32-10 + Box-Muller on the hardware log2/sqrt/sin/cos units (v_sin_f32
// takes revolutions, so 2*pi*u needs no multiply), three roundings to the storage
// dtype per reference op, no fma contraction (-ffp-contract=off).
// Algorithmic bytes: single pass 2*s per element; fused triple 4*s per element.
#include "common.h"

// ---------------------------------------------------------------- Philox4x32-10
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

static __device__ __forceinline__ void philox4x32_10(uint64_t counter, uint32_t k0, uint32_t k1,
                                                     uint32_t out[4]) {
    uint32_t c0 = (uint32_t)counter, c1 = (uint32_t)(counter >> 32), c2 = 0u, c3 = 0u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one v_mad_u64_u32 gives hi and lo of each 32x32 product
        const uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        const uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += PHILOX_W0;
        k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// two N(0,1) draws from two 32-bit words
static __device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
    // u1 in (0, 1]: (a + 0.5) * 2^-32 ; u2 in [0, 1]: revolutions
    const float u1 = __builtin_fmaf((float)a, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
    const float u2 = (float)b * 2.3283064365386963e-10f;
    // -2 ln(u1) = -2 ln2 * log2(u1)
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    z0 = r * __builtin_amdgcn_cosf(u2);
    z1 = r * __builtin_amdgcn_sinf(u2);
}

// z for the N elements of vector `vec` (element e = vec*N + i uses Philox counter e/4, word e%4)
template <int DT>
static __device__ __forceinline__ void gen_z(int64_t vec, uint32_t k0, uint32_t k1, float* z) {
    constexpr int N = Vec<DT>::N;
#pragma unroll
    for (int q = 0; q < N / 4; ++q) {
        uint32_t r[4];
        philox4x32_10((uint64_t)(vec * (N / 4) + q), k0, k1, r);
        box_muller(r[0], r[1], z[4 * q + 0], z[4 * q + 1]);
        box_muller(r[2], r[3], z[4 * q + 2], z[4 * q + 3]);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) z[i] = Vec<DT>::round(z[i]);  // torch.normal(dtype=param.dtype)
}

// scalar tail element
template <int DT>
static __device__ __forceinline__ float gen_z1(int64_t e, uint32_t k0, uint32_t k1) {
    uint32_t r[4];
    philox4x32_10((uint64_t)(e >> 2), k0, k1, r);
    float z[4];
    box_muller(r[0], r[1], z[0], z[1]);
    box_muller(r[2], r[3], z[2], z[3]);
    return Vec<DT>::round(z[e & 3]);
}

// one reference K1 application: three roundings (P:486)
template <int DT>
static __device__ __forceinline__ float k1_step(float w, float z, float sf, float eps) {
    const float t = Vec<DT>::round(z * sf);
    const float u = Vec<DT>::round(t * eps);
    return Vec<DT>::round(w + u);
}

// ---------------------------------------------------------------- kernels
template <int DT, bool HAS_Z>
__global__ __launch_bounds__(256) void zo_perturb_kernel(void* __restrict__ w,
                                                         const void* __restrict__ zin,
                                                         int64_t n, float sf, float eps,
                                                         uint32_t k0, uint32_t k1) {
    constexpr int N = Vec<DT>::N;
    const int64_t nvec = n / N;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        float wf[N], z[N];
        const u32x4 wv = ld16(w, v);
        if (HAS_Z) {
            Vec<DT>::unpack(ld16(zin, v), z);
        } else {
            gen_z<DT>(v, k0, k1, z);
        }
        Vec<DT>::unpack(wv, wf);
#pragma unroll
        for (int i = 0; i < N; ++i) wf[i] = k1_step<DT>(wf[i], z[i], sf, eps);
        st16(w, v, Vec<DT>::pack(wf));
    }
    // ragged tail (< N elements), one lane each
    const int64_t tail0 = nvec * N;
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)(n - tail0)) {
        const int64_t e = tail0 + threadIdx.x;
        const float z = HAS_Z ? Vec<DT>::load1(zin, e) : gen_z1<DT>(e, k0, k1);
        Vec<DT>::store1(w, e, k1_step<DT>(Vec<DT>::load1(w, e), z, sf, eps));
    }
}

// WRITE_PM = false: drift-only form (theta+ / theta- are not stored), used by ranks that do
// not own the batch but must keep their replica's weights identical (SURVEY.md §8e iii).
template <int DT, bool HAS_Z, bool WRITE_PM>
__global__ __launch_bounds__(256) void zo_perturb_triple_kernel(
    const void* win, void* wplus, void* wminus, void* wrest,  // wplus may alias win
    const void* __restrict__ zin, int64_t n, float eps, uint32_t k0, uint32_t k1) {
    constexpr int N = Vec<DT>::N;
    const int64_t nvec = n / N;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        float a[N], b[N], c[N], z[N];
        const u32x4 wv = ld16(win, v);
        if (HAS_Z) {
            Vec<DT>::unpack(ld16(zin, v), z);
        } else {
            gen_z<DT>(v, k0, k1, z);
        }
        Vec<DT>::unpack(wv, a);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            a[i] = k1_step<DT>(a[i], z[i], 1.0f, eps);   // theta + eps z
            b[i] = k1_step<DT>(a[i], z[i], -2.0f, eps);  // theta - eps z
            c[i] = k1_step<DT>(b[i], z[i], 1.0f, eps);   // "restored" (with the reference's drift)
        }
        if (WRITE_PM) {
            st16(wplus, v, Vec<DT>::pack(a));
            st16(wminus, v, Vec<DT>::pack(b));
        }
        st16(wrest, v, Vec<DT>::pack(c));
    }
    const int64_t tail0 = nvec * N;
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)(n - tail0)) {
        const int64_t e = tail0 + threadIdx.x;
        const float z = HAS_Z ? Vec<DT>::load1(zin, e) : gen_z1<DT>(e, k0, k1);
        const float a = k1_step<DT>(Vec<DT>::load1(win, e), z, 1.0f, eps);
        const float b = k1_step<DT>(a, z, -2.0f, eps);
        const float c = k1_step<DT>(b, z, 1.0f, eps);
        if (WRITE_PM) {
            Vec<DT>::store1(wplus, e, a);
            Vec<DT>::store1(wminus, e, b);
        }
        Vec<DT>::store1(wrest, e, c);
    }
}

template <int DT>
__global__ __launch_bounds__(256) void zo_fill_normal_kernel(void* __restrict__ zout, int64_t n,
                                                             uint32_t k0, uint32_t k1) {
    constexpr int N = Vec<DT>::N;
    const int64_t nvec = n / N;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        float z[N];
        gen_z<DT>(v, k0, k1, z);
        st16(zout, v, Vec<DT>::pack(z));
    }
    const int64_t tail0 = nvec * N;
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)(n - tail0)) {
        const int64_t e = tail0 + threadIdx.x;
        Vec<DT>::store1(zout, e, gen_z1<DT>(e, k0, k1));
    }
}

__global__ __launch_bounds__(256) void philox_u32_kernel(uint32_t* __restrict__ out, int64_t n,
                                                         uint32_t k0, uint32_t k1) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q * 4 < n; q += stride) {
        uint32_t r[4];
        philox4x32_10((uint64_t)q, k0, k1, r);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (q * 4 + i < n) out[q * 4 + i] = r[i];
    }
}

// ---------------------------------------------------------------- launch helpers
// Memory-bound grid: enough 256-thread blocks to keep every CU's queues full
// (256 CUs x 8), grid-stride beyond that.
static inline unsigned grid_for(int64_t nvec) {
    int64_t b = (nvec + 255) / 256;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (unsigned)b;
}

#define DISPATCH_DT(dt, ...)                                                  \
    switch (dt) {                                                             \
        case ECOFLAP_F32: { constexpr int DT = ECOFLAP_F32; __VA_ARGS__; } break;   \
        case ECOFLAP_F16: { constexpr int DT = ECOFLAP_F16; __VA_ARGS__; } break;   \
        case ECOFLAP_BF16: { constexpr int DT = ECOFLAP_BF16; __VA_ARGS__; } break; \
        default: return ECOFLAP_EDTYPE;                                       \
    }

extern "C" int ecoflap_zo_perturb(void* w, int64_t n, int dtype, float scaling_factor,
                                  float zo_eps, uint64_t seed, const void* z, void* stream) {
    if (!dtype_ok(dtype)) return ECOFLAP_EDTYPE;
    if (n < 0) return ECOFLAP_ESIZE;
    if (n == 0) return 0;
    if (!w) return ECOFLAP_ENULL;
    if (!aligned16(w) || (z && !aligned16(z))) return ECOFLAP_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    DISPATCH_DT(dtype, {
        const unsigned g = grid_for(n / Vec<DT>::N);
        if (z)
            hipLaunchKernelGGL((zo_perturb_kernel<DT, true>), dim3(g), dim3(256), 0, s, w, z, n,
                               scaling_factor, zo_eps, k0, k1);
        else
            hipLaunchKernelGGL((zo_perturb_kernel<DT, false>), dim3(g), dim3(256), 0, s, w, z, n,
                               scaling_factor, zo_eps, k0, k1);
    });
    ECO_CHECK_LAUNCH();
    return 0;
}

#define TRIPLE(HZ, PM)                                                                          \
    hipLaunchKernelGGL((zo_perturb_triple_kernel<DT, HZ, PM>), dim3(g), dim3(256), 0, s, w_in, \
                       w_plus, w_minus, w_restored, z, n, zo_eps, k0, k1)

extern "C" int ecoflap_zo_perturb_triple(const void* w_in, void* w_plus, void* w_minus,
                                         void* w_restored, int64_t n, int dtype, float zo_eps,
                                         uint64_t seed, const void* z, void* stream) {
    if (!dtype_ok(dtype)) return ECOFLAP_EDTYPE;
    if (n < 0) return ECOFLAP_ESIZE;
    if (n == 0) return 0;
    if (!w_in || !w_restored) return ECOFLAP_ENULL;
    const bool write_pm = (w_plus != nullptr) || (w_minus != nullptr);
    if (write_pm) {
        if (!w_plus || !w_minus) return ECOFLAP_ENULL;  // both or neither
        if (w_plus == w_minus || w_plus == w_restored || w_minus == w_restored) return ECOFLAP_ENULL;
        if (!aligned16(w_plus) || !aligned16(w_minus)) return ECOFLAP_EALIGN;
    }
    if (!aligned16(w_in) || !aligned16(w_restored) || (z && !aligned16(z))) return ECOFLAP_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    DISPATCH_DT(dtype, {
        const unsigned g = grid_for(n / Vec<DT>::N);
        if (z && write_pm) TRIPLE(true, true);
        else if (z) TRIPLE(true, false);
        else if (write_pm) TRIPLE(false, true);
        else TRIPLE(false, false);
    });
    ECO_CHECK_LAUNCH();
    return 0;
}

extern "C" int ecoflap_zo_fill_normal(void* z_out, int64_t n, int dtype, uint64_t seed,
                                      void* stream) {
    if (!dtype_ok(dtype)) return ECOFLAP_EDTYPE;
    if (n < 0) return ECOFLAP_ESIZE;
    if (n == 0) return 0;
    if (!z_out) return ECOFLAP_ENULL;
    if (!aligned16(z_out)) return ECOFLAP_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    DISPATCH_DT(dtype, {
        hipLaunchKernelGGL((zo_fill_normal_kernel<DT>), dim3(grid_for(n / Vec<DT>::N)), dim3(256),
                           0, s, z_out, n, k0, k1);
    });
    ECO_CHECK_LAUNCH();
    return 0;
}

extern "C" int ecoflap_philox_u32(uint32_t* out, int64_t n, uint64_t seed, void* stream) {
    if (n < 0) return ECOFLAP_ESIZE;
    if (n == 0) return 0;
    if (!out) return ECOFLAP_ENULL;
    hipLaunchKernelGGL(philox_u32_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, out, n, (uint32_t)seed, (uint32_t)(seed >> 32));
    ECO_CHECK_LAUNCH();
    return 0;
}
