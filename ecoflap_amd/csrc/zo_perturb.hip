// zo_perturb.hip — K1: zeroth-order weight perturbation for gfx950.
//
// Replaces LayerSparsity.zo_perturb_parameters
//   (LAVIS/lavis/compression/pruners/layer_single_base_pruner.py:473-486) and the
//   +1 / -2 / +1 call triple around the two loss evaluations (:530-539).
//
// HBM-bound elementwise read-modify-write: one 16-byte vector per lane per
// iteration (1 KiB per wave instruction), z produced in registers by
// Philox4x32-R (R = ECOFLAP_PHILOX_ROUNDS) + Box-Muller on the hardware log2/sqrt/sin/cos units (v_sin_f32
// takes revolutions, so 2*pi*u needs no multiply), three roundings to the storage
// dtype per reference op, no fma contraction (-ffp-contract=off).
// Algorithmic bytes: single pass 2*s per element; fused triple 4*s per element.
#include "common.h"

// ---------------------------------------------------------------- Philox4x32
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

static __device__ __forceinline__ void philox4x32(uint64_t counter, uint32_t k0, uint32_t k1,
                                                     uint32_t out[4]) {
    // Philox4x32-R with R = ECOFLAP_PHILOX_ROUNDS (include/ecoflap_hip.h)
    uint32_t c0 = (uint32_t)counter, c1 = (uint32_t)(counter >> 32), c2 = 0u, c3 = 0u;
#pragma unroll
    for (int r = 0; r < ECOFLAP_PHILOX_ROUNDS; ++r) {
        // one v_mad_u64_u32 gives hi and lo of each 32x32 product
        const uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        const uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        // gfx950 v_bitop3_b32 (truth table 0x96 = a ^ b ^ c): one op per 3-input xor
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
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
        philox4x32((uint64_t)(vec * (N / 4) + q), k0, k1, r);
        box_muller(r[0], r[1], z[4 * q + 0], z[4 * q + 1]);
        box_muller(r[2], r[3], z[4 * q + 2], z[4 * q + 3]);
    }
    // torch.normal(dtype=param.dtype): the fp32 draw rounded to the storage dtype
    if constexpr (DT == ECOFLAP_F32) {
        return;
    } else {
#pragma unroll
        for (int i = 0; i < N / 2; ++i)
            Vec<DT>::round_pair(z[2 * i], z[2 * i + 1], z[2 * i], z[2 * i + 1]);
    }
}

// scalar tail element
template <int DT>
static __device__ __forceinline__ float gen_z1(int64_t e, uint32_t k0, uint32_t k1) {
    uint32_t r[4];
    philox4x32((uint64_t)(e >> 2), k0, k1, r);
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

// One +1 / -2 / +1 unit on a whole vector: st (packed current weights) -> packed theta+ and
// theta-, st <- packed restored theta.
//
// FAST (16-bit dtypes, z generated in registers): the same VALUES as three k1_step calls with
// fewer instructions.  z is already a dtype value with |z| < 7, so t = rd(z*sf) is exact for
// sf in {1,-2} and needs no rounding; u(+1) is shared by the first and third step.
//   bf16: every rounding is one v_cvt_pk_bf16_f32 whose packed result is stored as is, and
//         u(-2) = -2*u(+1) exactly (power-of-two scaling commutes with rounding: bf16 has the
//         fp32 exponent range and |z*eps| > 1e-15 never reaches its subnormals).
//   fp16: the three adds are v_pk_add_f16 on packed pairs — for two f16 operands one f16
//         rounding equals torch's fp32-then-f16 rounding (24 >= 2*11+2 bits); z*eps and
//         -2z*eps go through fp32 and an explicit convert each, because eps is not an f16
//         value and z*eps is often an f16 subnormal.
// With caller-supplied z (parity mode) the generic three-rounding path is kept: a supplied
// fp16 z may overflow under -2z.  Both paths are checked bit for bit against the oracle.
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

template <int DT, bool FAST>
static __device__ __forceinline__ void unit_update(u32x4& st, const float* z, float eps,
                                                   u32x4& plus, u32x4& minus) {
    constexpr int N = Vec<DT>::N;
    if constexpr (FAST && DT == ECOFLAP_BF16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a0 = __uint_as_float(st[i] << 16), a1 = __uint_as_float(st[i] & 0xffff0000u);
            float ua, ub, pa, pb, ma, mb, ra, rb;
            Vec<DT>::round_pair(z[2 * i] * eps, z[2 * i + 1] * eps, ua, ub);
            plus[i] = Vec<DT>::round_pair(a0 + ua, a1 + ub, pa, pb);
            minus[i] = Vec<DT>::round_pair(pa + ua * -2.0f, pb + ub * -2.0f, ma, mb);
            st[i] = Vec<DT>::round_pair(ma + ua, mb + ub, ra, rb);
        }
    } else if constexpr (FAST && DT == ECOFLAP_F16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t u1 = Vec<DT>::f2h_pk(z[2 * i] * eps, z[2 * i + 1] * eps);
            const uint32_t u2 = Vec<DT>::f2h_pk((z[2 * i] * -2.0f) * eps, (z[2 * i + 1] * -2.0f) * eps);
            const half2_t h1 = __builtin_bit_cast(half2_t, u1), h2 = __builtin_bit_cast(half2_t, u2);
            const uint32_t cur = st[i];   // (bit_cast of a vector-element lvalue reads lane 0)
            const half2_t p = __builtin_bit_cast(half2_t, cur) + h1;
            const half2_t m = p + h2;
            const half2_t r = m + h1;
            plus[i] = __builtin_bit_cast(uint32_t, p);
            minus[i] = __builtin_bit_cast(uint32_t, m);
            st[i] = __builtin_bit_cast(uint32_t, r);
        }
    } else {
        float a[N], pf[N], mf[N];
        Vec<DT>::unpack(st, a);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            pf[i] = k1_step<DT>(a[i], z[i], 1.0f, eps);    // theta + eps z
            mf[i] = k1_step<DT>(pf[i], z[i], -2.0f, eps);  // theta - eps z
            a[i] = k1_step<DT>(mf[i], z[i], 1.0f, eps);    // "restored" (with the reference's drift)
        }
        plus = Vec<DT>::pack(pf);
        minus = Vec<DT>::pack(mf);
        st = Vec<DT>::pack(a);
    }
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
        float z[N];
        u32x4 st = ld16(win, v);
        if (HAS_Z) {
            Vec<DT>::unpack(ld16(zin, v), z);
        } else {
            gen_z<DT>(v, k0, k1, z);
        }
        u32x4 pp, mm;
        unit_update<DT, !HAS_Z>(st, z, eps, pp, mm);
        if (WRITE_PM) {
            st16(wplus, v, pp);
            st16(wminus, v, mm);
        }
        st16(wrest, v, st);
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

// ---------------------------------------------------------------- layer-batched form
// All perturbation units of one layer (its calibration batches x noise draws) in ONE pass:
// the unit chain  theta_0 -> (theta+_1, theta-_1, theta_1) -> (theta+_2, ...)  is an
// elementwise function of theta_0 and the seeds, so W is read once, every unit's theta+ /
// theta- goes to its own buffer (non-temporal: each is read once, by one forward, after
// gigabytes of other traffic) and only the final drifted theta is written back in place.
// 288 GB of HBM make the 2*U scratch copies of a matrix cheap (U=16: 0.67 GB for the largest).
// Algorithmic bytes: (2*U_owned + 2) * s per element per launch.
struct UnitTable {
    uint64_t seed[ECOFLAP_MAX_UNITS];
    void* plus[ECOFLAP_MAX_UNITS];
    void* minus[ECOFLAP_MAX_UNITS];
    const void* z[ECOFLAP_MAX_UNITS];
};

static __device__ __forceinline__ void st16_nt(void* p, int64_t vec_index, const u32x4& v) {
    __builtin_nontemporal_store(v, ((u32x4*)p) + vec_index);
}

template <int DT, bool HAS_Z>
__global__ __launch_bounds__(256) void zo_perturb_units_kernel(void* __restrict__ w, int64_t n,
                                                               float eps, int n_units,
                                                               const UnitTable tab) {
    constexpr int N = Vec<DT>::N;
    const int64_t nvec = n / N;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        u32x4 st = ld16(w, v);
        for (int u = 0; u < n_units; ++u) {
            float z[N];
            if (HAS_Z) {
                Vec<DT>::unpack(ld16(tab.z[u], v), z);
            } else {
                gen_z<DT>(v, (uint32_t)tab.seed[u], (uint32_t)(tab.seed[u] >> 32), z);
            }
            u32x4 pp, mm;
            unit_update<DT, !HAS_Z>(st, z, eps, pp, mm);
            void* dst = tab.plus[u];
            if (dst) {   // wave-uniform: not-owned units only carry the drift
                st16_nt(dst, v, pp);
                st16_nt(tab.minus[u], v, mm);
            }
        }
        st16(w, v, st);
    }
    const int64_t tail0 = nvec * N;
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)(n - tail0)) {
        const int64_t e = tail0 + threadIdx.x;
        float a = Vec<DT>::load1(w, e);
        for (int u = 0; u < n_units; ++u) {
            const float z = HAS_Z ? Vec<DT>::load1(tab.z[u], e)
                                  : gen_z1<DT>(e, (uint32_t)tab.seed[u], (uint32_t)(tab.seed[u] >> 32));
            a = k1_step<DT>(a, z, 1.0f, eps);
            const float b = k1_step<DT>(a, z, -2.0f, eps);
            if (tab.plus[u]) {
                Vec<DT>::store1(tab.plus[u], e, a);
                Vec<DT>::store1(tab.minus[u], e, b);
            }
            a = k1_step<DT>(b, z, 1.0f, eps);
        }
        Vec<DT>::store1(w, e, a);
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
        philox4x32((uint64_t)q, k0, k1, r);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (q * 4 + i < n) out[q * 4 + i] = r[i];
    }
}

// ---------------------------------------------------------------- launch helpers
// one 16-byte vector per lane (measured fastest on MI355X for 8-85 MB of traffic), grid-stride
// only beyond 65536 workgroups
static inline unsigned grid_exact(int64_t nvec) {
    int64_t b = (nvec + 255) / 256;
    if (b < 1) b = 1;
    if (b > 65536) b = 65536;
    return (unsigned)b;
}


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
        const unsigned g = grid_exact(n / Vec<DT>::N);
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
        const unsigned g = grid_exact(n / Vec<DT>::N);
        if (z && write_pm) TRIPLE(true, true);
        else if (z) TRIPLE(true, false);
        else if (write_pm) TRIPLE(false, true);
        else TRIPLE(false, false);
    });
    ECO_CHECK_LAUNCH();
    return 0;
}

extern "C" int ecoflap_zo_perturb_units(void* w, int64_t n, int dtype, float zo_eps, int n_units,
                                        const uint64_t* seeds, void* const* w_plus,
                                        void* const* w_minus, const void* const* z,
                                        void* stream) {
    if (!dtype_ok(dtype)) return ECOFLAP_EDTYPE;
    if (n < 0 || n_units < 0 || n_units > ECOFLAP_MAX_UNITS) return ECOFLAP_ESIZE;
    if (n == 0 || n_units == 0) return 0;
    if (!w || !seeds || !w_plus || !w_minus) return ECOFLAP_ENULL;
    if (!aligned16(w)) return ECOFLAP_EALIGN;
    UnitTable tab;
    for (int u = 0; u < ECOFLAP_MAX_UNITS; ++u) {
        tab.seed[u] = 0; tab.plus[u] = nullptr; tab.minus[u] = nullptr; tab.z[u] = nullptr;
    }
    for (int u = 0; u < n_units; ++u) {
        tab.seed[u] = seeds[u];
        if ((w_plus[u] == nullptr) != (w_minus[u] == nullptr)) return ECOFLAP_ENULL;
        if (w_plus[u]) {
            if (w_plus[u] == w || w_minus[u] == w || w_plus[u] == w_minus[u]) return ECOFLAP_ENULL;
            if (!aligned16(w_plus[u]) || !aligned16(w_minus[u])) return ECOFLAP_EALIGN;
        }
        tab.plus[u] = w_plus[u];
        tab.minus[u] = w_minus[u];
        if (z) {
            if (!z[u]) return ECOFLAP_ENULL;
            if (!aligned16(z[u])) return ECOFLAP_EALIGN;
            tab.z[u] = z[u];
        }
    }
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_DT(dtype, {
        const unsigned g = grid_exact(n / Vec<DT>::N);
        if (z)
            hipLaunchKernelGGL((zo_perturb_units_kernel<DT, true>), dim3(g), dim3(256), 0, s, w, n,
                               zo_eps, n_units, tab);
        else
            hipLaunchKernelGGL((zo_perturb_units_kernel<DT, false>), dim3(g), dim3(256), 0, s, w, n,
                               zo_eps, n_units, tab);
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
