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
#include <hip/hip_ext.h>

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

// ---------------------------------------------------------------- in-register z streams
// Work unit of every K1 kernel: one lane owns TWO 16-byte vectors, v and v + 64 (the same lane
// of two consecutive 1-KiB wave rows), i.e. a wave owns a "super-row" of 128 vectors; every
// global access stays a fully coalesced 1 KiB per wave instruction.
//
// Stream N32 (fp32 storage): element e uses Philox call e/4 (counter = e/4, key = seed); words
//   (w0,w1) -> elements 4c+0 = r cos, 4c+1 = r sin; (w2,w3) -> 4c+2, 4c+3.  32-bit radius word
//   and 32-bit angle word per pair.
// Stream N16 (fp16 / bf16 storage): a bf16 / fp16 z cannot resolve an angle finer than 2^-16 of
//   a turn (its own rounding is 2^-9 / 2^-12 relative), so a pair takes a full 32-bit radius
//   word (tail to 6.7 sigma, as N32) and a 16-bit angle: 48 bits per pair, THREE Philox calls
//   per 16 elements instead of four.  Block g = 64*(v/128) + v%64 holds the 16 elements of
//   vectors (128*(v/128) + v%64) and (that + 64); calls 3g, 3g+1, 3g+2 give words W[0..11];
//   pair j (0..7): radius word W[j], angle = halfword j of W[8..11] (low half first);
//   position p = 8*((v/64)%2) + e%8 = 4q + t takes pair 2q + (t&1), cos for t < 2, sin for
//   t >= 2 — so that packed register i of a vector holds (r_a f(a), r_b f(b)) of two pairs and
//   every rounding below is one packed convert of values that sit in adjacent registers.
// oracle/ecoflap_oracle.c:oracle_normal_stream restates both (double-precision log2 / sqrt /
// sin / cos rounded once to fp32; the hardware units differ from that by a few fp32 ulps, the
// tolerance the generator test states).
#define ECO_ROW 64          // vectors per wave row; a lane's two vectors are ECO_ROW apart

static __device__ __forceinline__ float radius_of(uint32_t a) {
    // u1 in (0, 1]: (a + 0.5) * 2^-32 ;  r = sqrt(-2 ln u1) = sqrt(-2 ln2 * log2 u1)
    const float u1 = __builtin_fmaf((float)a, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
    return __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
}

// two N(0,1) draws from two 32-bit words (N32)
static __device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
    const float u2 = (float)b * 2.3283064365386963e-10f;       // revolutions in [0, 1]
    const float r = radius_of(a);
    z0 = r * __builtin_amdgcn_cosf(u2);
    z1 = r * __builtin_amdgcn_sinf(u2);
}

// N32: z for the 4 elements of fp32 vector `vec`
static __device__ __forceinline__ void gen_z4(int64_t vec, uint32_t k0, uint32_t k1, float* z) {
    uint32_t r[4];
    philox4x32((uint64_t)vec, k0, k1, r);
    box_muller(r[0], r[1], z[0], z[1]);
    box_muller(r[2], r[3], z[2], z[3]);
}

// N16: z (already rounded to the storage dtype) for the 16 elements of block g:
// z[0..7] -> vector (g/64)*128 + g%64, z[8..15] -> that + 64
template <int DT>
static __device__ __forceinline__ void gen_z16(int64_t g, uint32_t k0, uint32_t k1, float* z) {
    uint32_t w[12];
    philox4x32((uint64_t)(3 * g + 0), k0, k1, w + 0);
    philox4x32((uint64_t)(3 * g + 1), k0, k1, w + 4);
    philox4x32((uint64_t)(3 * g + 2), k0, k1, w + 8);
#pragma unroll
    for (int q = 0; q < 4; ++q) {          // pairs a = 2q, b = 2q + 1
        const uint32_t ang = w[8 + q];
        const float ua = (float)(ang & 0xffffu) * 1.52587890625e-05f;   // exact: h * 2^-16
        const float ub = (float)(ang >> 16) * 1.52587890625e-05f;
        const float ra = radius_of(w[2 * q]), rb = radius_of(w[2 * q + 1]);
        const float ca = __builtin_amdgcn_cosf(ua), cb = __builtin_amdgcn_cosf(ub);
        const float sa = __builtin_amdgcn_sinf(ua), sb = __builtin_amdgcn_sinf(ub);
        // torch.normal(dtype=param.dtype): the fp32 draw rounded to the storage dtype
        Vec<DT>::round_pair(ra * ca, rb * cb, z[4 * q + 0], z[4 * q + 1]);
        Vec<DT>::round_pair(ra * sa, rb * sb, z[4 * q + 2], z[4 * q + 3]);
    }
}

// z of a lane's two vectors (v0 = 128R + lane, v1 = v0 + 64): z[0..N-1], z[N..2N-1]
template <int DT>
static __device__ __forceinline__ void gen_z_lane(int64_t v0, uint32_t k0, uint32_t k1, float* z) {
    if constexpr (DT == ECOFLAP_F32) {
        gen_z4(v0, k0, k1, z);
        gen_z4(v0 + ECO_ROW, k0, k1, z + 4);
    } else {
        gen_z16<DT>(((v0 >> 7) << 6) + (v0 & 63), k0, k1, z);
    }
}

// scalar tail element (n % N != 0: the last < N elements belong to no full vector)
template <int DT>
static __device__ __forceinline__ float gen_z1(int64_t e, uint32_t k0, uint32_t k1) {
    if constexpr (DT == ECOFLAP_F32) {
        float z[4];
        gen_z4(e >> 2, k0, k1, z);
        return z[e & 3];
    } else {
        const int64_t v = e >> 3;
        float z[16];
        gen_z16<DT>(((v >> 7) << 6) + (v & 63), k0, k1, z);
        return z[(((v >> 6) & 1) << 3) + (e & 7)];
    }
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
//         fp32 exponent range and |z*eps| > 1e-15 never reaches its subnormals), so
//         theta- = rd(fma(u, -2, theta+)): the product is exact, one rounding, as torch's add.
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
            minus[i] = Vec<DT>::round_pair(__builtin_fmaf(ua, -2.0f, pa), __builtin_fmaf(ub, -2.0f, pb),
                                           ma, mb);
            st[i] = Vec<DT>::round_pair(ma + ua, mb + ub, ra, rb);
        }
    } else if constexpr (FAST && DT == ECOFLAP_F16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // z is an f16 value held as f32: fma(z, eps, -0) = z * eps bit for bit (also for a zero
            // product's sign) and lets the f16 -> f32 convert fold into the multiply
            // (v_fma_mix_f32); (z * -2) * eps = z * (-2 eps): both scalings by two are exact.
            const float m2eps = -2.0f * eps;
            const uint32_t u1 = Vec<DT>::f2h_pk(__builtin_fmaf(z[2 * i], eps, -0.0f),
                                                __builtin_fmaf(z[2 * i + 1], eps, -0.0f));
            const uint32_t u2 = Vec<DT>::f2h_pk(__builtin_fmaf(z[2 * i], m2eps, -0.0f),
                                                __builtin_fmaf(z[2 * i + 1], m2eps, -0.0f));
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
// Lane geometry shared by all of them: wave `R` (grid-stride over super-rows) owns vectors
// [128R, 128R + 128); lane l owns v0 = 128R + l and v1 = v0 + 64 (either may lie past the end).
struct LaneVecs {
    int64_t v0, v1;
    bool ok0, ok1;
};
static __device__ __forceinline__ LaneVecs lane_vecs(int64_t R, int64_t nvec) {
    LaneVecs L;
    L.v0 = (R << 7) + (threadIdx.x & 63);
    L.v1 = L.v0 + ECO_ROW;
    L.ok0 = L.v0 < nvec;
    L.ok1 = L.v1 < nvec;
    return L;
}
// One wave per workgroup, one super-row per wave.  Workgroups are dispatched round-robin over
// the 8 XCDs (workgroup b runs on XCD b % 8, each with its own L2), so the row index is remapped
// for every XCD to walk ONE contiguous eighth of the matrix: the 2*U+1 write streams of a pass
// then reach HBM as 8 sequential fronts per stream instead of a 2-KiB interleave of all XCDs
// (memory-only twin of the kernel, tools/micro/k1_stores.hip: 63-69 % -> 68-73 % of 8 TB/s).
#define ECO_XCDS 8
#define ECO_K1_THREADS 64
#define ECO_FOR_SUPER_ROWS(R, nvec)                                                          \
    for (int64_t nrows__ = ((nvec) + 127) >> 7, per__ = (int64_t)(gridDim.x / ECO_XCDS),      \
                 R = (int64_t)(blockIdx.x % ECO_XCDS) * per__ + blockIdx.x / ECO_XCDS,        \
                 once__ = 0;                                                                  \
         once__ == 0 && R < nrows__; ++once__)

static __device__ __forceinline__ u32x4 ld16_if(bool ok, const void* p, int64_t v) {
    u32x4 r = {0u, 0u, 0u, 0u};
    if (ok) r = ld16(p, v);
    return r;
}

// z of the lane's two vectors from memory (parity mode)
template <int DT>
static __device__ __forceinline__ void load_z_lane(const void* zin, const LaneVecs& L, float* z) {
    constexpr int N = Vec<DT>::N;
    Vec<DT>::unpack(ld16_if(L.ok0, zin, L.v0), z);
    Vec<DT>::unpack(ld16_if(L.ok1, zin, L.v1), z + N);
}

template <int DT, bool HAS_Z>
__global__ __launch_bounds__(ECO_K1_THREADS) void zo_perturb_kernel(void* __restrict__ w,
                                                         const void* __restrict__ zin,
                                                         int64_t n, float sf, float eps,
                                                         uint32_t k0, uint32_t k1) {
    constexpr int N = Vec<DT>::N;
    const int64_t nvec = n / N;
    ECO_FOR_SUPER_ROWS(R, nvec) {
        const LaneVecs L = lane_vecs(R, nvec);
        float wf[2 * N], z[2 * N];
        const u32x4 w0 = ld16_if(L.ok0, w, L.v0), w1 = ld16_if(L.ok1, w, L.v1);
        if (HAS_Z) {
            load_z_lane<DT>(zin, L, z);
        } else {
            gen_z_lane<DT>(L.v0, k0, k1, z);
        }
        Vec<DT>::unpack(w0, wf);
        Vec<DT>::unpack(w1, wf + N);
#pragma unroll
        for (int i = 0; i < 2 * N; ++i) wf[i] = k1_step<DT>(wf[i], z[i], sf, eps);
        if (L.ok0) st16(w, L.v0, Vec<DT>::pack(wf));
        if (L.ok1) st16(w, L.v1, Vec<DT>::pack(wf + N));
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
__global__ __launch_bounds__(ECO_K1_THREADS) void zo_perturb_triple_kernel(
    const void* win, void* wplus, void* wminus, void* wrest,  // wplus may alias win
    const void* __restrict__ zin, int64_t n, float eps, uint32_t k0, uint32_t k1) {
    constexpr int N = Vec<DT>::N;
    const int64_t nvec = n / N;
    ECO_FOR_SUPER_ROWS(R, nvec) {
        const LaneVecs L = lane_vecs(R, nvec);
        float z[2 * N];
        u32x4 s0 = ld16_if(L.ok0, win, L.v0), s1 = ld16_if(L.ok1, win, L.v1);
        if (HAS_Z) {
            load_z_lane<DT>(zin, L, z);
        } else {
            gen_z_lane<DT>(L.v0, k0, k1, z);
        }
        u32x4 p0, m0, p1, m1;
        unit_update<DT, !HAS_Z>(s0, z, eps, p0, m0);
        unit_update<DT, !HAS_Z>(s1, z + N, eps, p1, m1);
        if (L.ok0) {
            if (WRITE_PM) {
                st16(wplus, L.v0, p0);
                st16(wminus, L.v0, m0);
            }
            st16(wrest, L.v0, s0);
        }
        if (L.ok1) {
            if (WRITE_PM) {
                st16(wplus, L.v1, p1);
                st16(wminus, L.v1, m1);
            }
            st16(wrest, L.v1, s1);
        }
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
__global__ __launch_bounds__(ECO_K1_THREADS) void zo_perturb_units_kernel(void* __restrict__ w, int64_t n,
                                                               float eps, int n_units,
                                                               const UnitTable tab) {
    constexpr int N = Vec<DT>::N;
    const int64_t nvec = n / N;
    ECO_FOR_SUPER_ROWS(R, nvec) {
        const LaneVecs L = lane_vecs(R, nvec);
        u32x4 s0 = ld16_if(L.ok0, w, L.v0), s1 = ld16_if(L.ok1, w, L.v1);
        for (int u = 0; u < n_units; ++u) {
            float z[2 * N];
            if (HAS_Z) {
                load_z_lane<DT>(tab.z[u], L, z);
            } else {
                gen_z_lane<DT>(L.v0, (uint32_t)tab.seed[u], (uint32_t)(tab.seed[u] >> 32), z);
            }
            u32x4 p0, m0, p1, m1;
            unit_update<DT, !HAS_Z>(s0, z, eps, p0, m0);
            unit_update<DT, !HAS_Z>(s1, z + N, eps, p1, m1);
            void* dp = tab.plus[u];
            if (dp) {   // wave-uniform: not-owned units only carry the drift
                void* dm = tab.minus[u];
                if (L.ok0) {
                    st16_nt(dp, L.v0, p0);
                    st16_nt(dm, L.v0, m0);
                }
                if (L.ok1) {
                    st16_nt(dp, L.v1, p1);
                    st16_nt(dm, L.v1, m1);
                }
            }
        }
        if (L.ok0) st16(w, L.v0, s0);
        if (L.ok1) st16(w, L.v1, s1);
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

// ---------------------------------------------------------------- block-batched form
// Every layer of one transformer block (the matrices scored back to back, all re-entering the
// forward at the same stage) in ONE launch.  K1 of a layer is a function of its original weights
// and its seeds only, so it can run ahead of the layer's turn — provided the drifted weights are
// NOT written over the originals yet (the other layers of the block are still evaluated with
// them): they go to a buffer of their own and the caller copies them in when the layer is done.
// What it buys is the per-launch cost a 135-285 MB pass pays in the loop (cold instruction and
// translation caches, dirty L2 lines of the forward that ran before): once per block instead of
// once per matrix.  Device table, one row of ECO_LAYER_ROW int64 per layer:
//   [0] w_in  [1] w_final  [2] numel  [3] n_units  [4] first super-row of the layer in the launch
//   [5 .. 5+U) seeds   [5+U .. 5+2U) theta+ pointers   [5+2U .. 5+3U) theta- pointers   (U = MAX_UNITS)
// The parity form (HAS_Z: z drawn by the caller as the reference draws it, torch.manual_seed +
// torch.normal per unit, read from memory here) carries one more column,
//   [5+3U .. 5+4U) z pointers,
// and runs the generic three-rounding update (a supplied z is not known to be safe for the
// shortcuts of the in-register path).  Algorithmic bytes: (3*U_owned' + 2) * s per element with
// U_owned' = owned units, plus s per element for each NOT-owned unit's z (drift needs it too).
#define ECO_LAYER_ROW (5 + 3 * ECOFLAP_MAX_UNITS)
#define ECO_LAYER_ROW_Z (5 + 4 * ECOFLAP_MAX_UNITS)

template <int DT, bool HAS_Z>
__global__ __launch_bounds__(ECO_K1_THREADS) void zo_perturb_layers_kernel(
    const int64_t* __restrict__ table, int n_layers, int64_t total_rows, float eps) {
    constexpr int N = Vec<DT>::N;
    constexpr int ROW = HAS_Z ? ECO_LAYER_ROW_Z : ECO_LAYER_ROW;
    const int64_t per = (int64_t)(gridDim.x / ECO_XCDS);
    const int64_t Rg = (int64_t)(blockIdx.x % ECO_XCDS) * per + blockIdx.x / ECO_XCDS;
    if (Rg >= total_rows) return;
    int l = 0;                                     // wave-uniform scan: scalar loads
    while (l + 1 < n_layers && Rg >= table[(int64_t)(l + 1) * ROW + 4]) ++l;
    const int64_t* __restrict__ row = table + (int64_t)l * ROW;
    const void* win = (const void*)row[0];
    void* wout = (void*)row[1];
    const int64_t n = row[2];
    const int n_units = (int)row[3];
    const int64_t R = Rg - row[4];
    const int64_t nvec = n / N;
    const LaneVecs L = lane_vecs(R, nvec);
    u32x4 s0 = ld16_if(L.ok0, win, L.v0), s1 = ld16_if(L.ok1, win, L.v1);
    for (int u = 0; u < n_units; ++u) {
        float z[2 * N];
        if constexpr (HAS_Z) {
            load_z_lane<DT>((const void*)row[5 + 3 * ECOFLAP_MAX_UNITS + u], L, z);
        } else {
            const uint64_t seed = (uint64_t)row[5 + u];
            gen_z_lane<DT>(L.v0, (uint32_t)seed, (uint32_t)(seed >> 32), z);
        }
        u32x4 p0, m0, p1, m1;
        unit_update<DT, !HAS_Z>(s0, z, eps, p0, m0);
        unit_update<DT, !HAS_Z>(s1, z + N, eps, p1, m1);
        void* dp = (void*)row[5 + ECOFLAP_MAX_UNITS + u];
        if (dp) {   // wave-uniform: not-owned units only carry the drift
            void* dm = (void*)row[5 + 2 * ECOFLAP_MAX_UNITS + u];
            if (L.ok0) {
                st16_nt(dp, L.v0, p0);
                st16_nt(dm, L.v0, m0);
            }
            if (L.ok1) {
                st16_nt(dp, L.v1, p1);
                st16_nt(dm, L.v1, m1);
            }
        }
    }
    if (L.ok0) st16(wout, L.v0, s0);
    if (L.ok1) st16(wout, L.v1, s1);
    const int64_t tail0 = nvec * N;                // ragged tail of this layer: its first wave
    if (R == 0 && threadIdx.x < (unsigned)(n - tail0)) {
        const int64_t e = tail0 + threadIdx.x;
        float a = Vec<DT>::load1(win, e);
        for (int u = 0; u < n_units; ++u) {
            float z;
            if constexpr (HAS_Z) {
                z = Vec<DT>::load1((const void*)row[5 + 3 * ECOFLAP_MAX_UNITS + u], e);
            } else {
                const uint64_t seed = (uint64_t)row[5 + u];
                z = gen_z1<DT>(e, (uint32_t)seed, (uint32_t)(seed >> 32));
            }
            a = k1_step<DT>(a, z, 1.0f, eps);
            const float b = k1_step<DT>(a, z, -2.0f, eps);
            void* dp = (void*)row[5 + ECOFLAP_MAX_UNITS + u];
            if (dp) {
                Vec<DT>::store1(dp, e, a);
                Vec<DT>::store1((void*)row[5 + 2 * ECOFLAP_MAX_UNITS + u], e, b);
            }
            a = k1_step<DT>(b, z, 1.0f, eps);
        }
        Vec<DT>::store1(wout, e, a);
    }
}

template <int DT>
__global__ __launch_bounds__(ECO_K1_THREADS) void zo_fill_normal_kernel(void* __restrict__ zout, int64_t n,
                                                             uint32_t k0, uint32_t k1) {
    constexpr int N = Vec<DT>::N;
    const int64_t nvec = n / N;
    ECO_FOR_SUPER_ROWS(R, nvec) {
        const LaneVecs L = lane_vecs(R, nvec);
        float z[2 * N];
        gen_z_lane<DT>(L.v0, k0, k1, z);
        if (L.ok0) st16(zout, L.v0, Vec<DT>::pack(z));
        if (L.ok1) st16(zout, L.v1, Vec<DT>::pack(z + N));
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

// ---------------------------------------------------------------- torch's own device stream, in registers
// The reference draws z with torch.manual_seed(seed); torch.normal(0, 1, size, device, dtype)
// (P:482-485).  On this device that is ATen's distribution_elementwise_grid_stride_kernel
// <float, 4> (torch/include/ATen/native/cuda/DistributionTemplates.h:50-100, shipped with the
// wheel) over rocRAND's Philox4x32-10 state: T = 256 * grid threads, grid = min(SMs *
// (maxThreadsPerSM / 256), ceil(n / 256)); thread idx draws rocrand_normal4 once per round j
// from Philox(counter = {j, 0, idx, 0}, key = {seed_lo, seed_hi}) (curand_init(seed,
// subsequence = idx, offset = 0): manual_seed resets the generator's offset) and writes the four
// values to elements idx + T*(4j + ii), ii = 0..3.  Values: rocrand_device::detail::box_muller
// (rocrand_normal.h:53-68) as hipcc compiled it into libtorch_hip.so for gfx950 — restated
// below instruction for instruction from that kernel's disassembly (fma contractions, the
// v_log_f32-based logf with its two-constant ln2 product, the correctly-rounded sqrtf fix-up,
// __sincosf = v_sin_f32 / v_cos_f32 of v * (1/2pi)), then ATen's transform fma(std, x, mean)
// with std = 1, mean = 0 and one rounding to the storage dtype.  Nothing here is "close to"
// torch's stream: tests/test_gpu_parity.py holds it to torch.normal bit for bit, and
// pruners/layer_sparsity.py probes that equality at start-up before it trusts this path (a
// torch / rocRAND upgrade that changes the stream sends the run back to materialised draws).
//
// Lane geometry: the four values of one Philox call lie T elements apart, so a lane takes the
// N subsequences idx0 .. idx0+N-1 of one round (N = elements per 16-byte vector) and owns FOUR
// vectors, one in each of the round's four rows — every global access of the wave is still a
// contiguous 1 KiB.  Work item I of a layer = (round j, wave chunk c): I = j * wpr + c,
// wpr = ceil(T / N / 64).
static __device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                     uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        const uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
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

// rocRAND's radius s = sqrtf(-2 logf(u)) of a 32-bit word, restated instruction for instruction
// from torch's kernel (the reference sequence: ocml's logf = v_log_f32 times a two-constant ln 2,
// the correctly rounded sqrtf = v_sqrt_f32 and one step either way).  Kept as the yardstick of
// `torch_radius` below (ecoflap_zo_torch_radius_sweep); no product kernel calls it.
static __device__ __forceinline__ float torch_radius_reference(uint32_t x) {
    const float u = __builtin_fmaf((float)x, 2.3283064365386963e-10f, 2.3283064365386963e-10f);
    // logf(u) (ocml, u is never subnormal here: u >= 2^-32)
    const float r = __builtin_amdgcn_logf(u);                       // v_log_f32: log2
    // ocml's split of ln 2 (hi 0x3f317217 + lo 0x3377d1cf) with rocRAND's "-2 *" folded into both
    // constants: scaling by -2 is exact at every step (no value here is subnormal), so
    // x2 = -2 * (yl + fma(lo, r, fma(r, hi, -yl))) bit for bit
    const float m2ln2_hi = __uint_as_float(0xbfb17217u);
    const float yl = r * m2ln2_hi;
    float t = __builtin_fmaf(r, m2ln2_hi, -yl);
    t = __builtin_fmaf(__uint_as_float(0xb3f7d1cfu), r, t);
    const float x2 = yl + t;
    // sqrtf(x2), correctly rounded: v_sqrt_f32 and one step either way (x2 is 0 or >= 1e-7:
    // ocml's rescaling of tiny arguments never triggers)
    const float s0 = __builtin_amdgcn_sqrtf(x2);
    const float sm = __uint_as_float(__float_as_uint(s0) - 1u), sp = __uint_as_float(__float_as_uint(s0) + 1u);
    const float rm = __builtin_fmaf(-sm, s0, x2), rp = __builtin_fmaf(-sp, s0, x2);
    float s = (0.0f >= rm) ? sm : s0;
    s = (0.0f < rp) ? sp : s;
    return s;
}

// The same VALUE for every one of the 2^32 words in 6 fewer instructions per pair (round 6; K1 in
// this mode is VALU-bound, tools/micro/k1_torch_bound.hip):
//   * ln 2 product: x2 = fma(r, hi, rn(lo * r)) — the exact product r * hi plus the rounded low
//     term, rounded once, instead of ocml's four-instruction compensated form;
//   * sqrt: y = v_rsq_f32(x2), g = x2 * y, s = fma(fma(-g, g, x2), y / 2, g) — one Newton step on
//     an exact residual instead of v_sqrt_f32 + two trial residuals + two selects;
//   * the 128 words that round to u = 1 (x2 = -0 here, +0 in the reference; rsq gives an
//     infinity and g a NaN): v_max_f32 against +0 returns the +0 the reference has.
// Neither shortcut is correctly rounded IN GENERAL; on this function's whole domain — which is
// enumerable — they are: ecoflap_zo_torch_radius_sweep compares the two sequences over every
// 32-bit word on the device (tests/test_torch_stream.py runs all 2^32; 0 differ).
static __device__ __forceinline__ float torch_radius(uint32_t x) {
    const float u = __builtin_fmaf((float)x, 2.3283064365386963e-10f, 2.3283064365386963e-10f);
    const float r = __builtin_amdgcn_logf(u);
    const float x2 = __builtin_fmaf(r, __uint_as_float(0xbfb17217u), __uint_as_float(0xb3f7d1cfu) * r);
    const float y = __builtin_amdgcn_rsqf(x2);
    const float g = x2 * y, h = 0.5f * y;
    const float e = __builtin_fmaf(-g, g, x2);
    float s = __builtin_fmaf(e, h, g);
    asm("v_max_f32 %0, 0, %1" : "=v"(s) : "v"(s));   // NaN (u = 1) -> +0; s >= 3e-4 otherwise
    return s;
}

// rocrand box_muller(x, y) -> (sin(v) * s, cos(v) * s), as compiled into torch's kernel
static __device__ __forceinline__ void torch_box_muller(uint32_t x, uint32_t y, float& zs, float& zc) {
    const float s = torch_radius(x);
    const float v = __builtin_fmaf((float)y, __uint_as_float(0x30c90fdbu), __uint_as_float(0x30c90fdbu));
    const float a = v * __uint_as_float(0x3e22f983u);               // 1/(2 pi): revolutions for v_sin / v_cos
    // rocRAND's product, then ATen's transform fma(std = 1, x, mean = 0) = x + 0 (a -0 becomes
    // +0).  The product is an exact zero or a normal number (|sin|, |cos| >= 1e-9 unless 0,
    // s >= 3e-4 unless 0), so ONE fma with a +0 addend rounds to the same bits as the multiply
    // followed by the add.
    zs = __builtin_fmaf(__builtin_amdgcn_sinf(a), s, 0.0f);
    zc = __builtin_fmaf(__builtin_amdgcn_cosf(a), s, 0.0f);
}

// z (rounded to the storage dtype, held as f32) of a lane's four vectors in round j:
// z[ii * N + t] = element idx0 + t + T * (4j + ii)
template <int DT>
static __device__ __forceinline__ void torch_z_tile(uint32_t j, uint32_t idx0, uint32_t k0, uint32_t k1,
                                                    float* z) {
    constexpr int N = Vec<DT>::N;
#pragma unroll
    for (int t = 0; t < N; ++t) {
        uint32_t w[4];
        philox4x32_10(j, 0u, idx0 + t, 0u, k0, k1, w);
        torch_box_muller(w[0], w[1], z[0 * N + t], z[1 * N + t]);
        torch_box_muller(w[2], w[3], z[2 * N + t], z[3 * N + t]);
    }
    // ATen: static_cast<scalar_t>(...), one rounding to the storage dtype
    if constexpr (DT != ECOFLAP_F32) {
#pragma unroll
        for (int i = 0; i < 4 * N; i += 2) Vec<DT>::round_pair(z[i], z[i + 1], z[i], z[i + 1]);
    }
}

// one element (ragged tails, the scalar restatement the tile is tested against)
template <int DT>
static __device__ __forceinline__ float torch_z1(int64_t e, int64_t T, uint32_t k0, uint32_t k1) {
    const int64_t q = e / T;
    const uint32_t idx = (uint32_t)(e - q * T), j = (uint32_t)(q >> 2);
    uint32_t w[4];
    philox4x32_10(j, 0u, idx, 0u, k0, k1, w);
    float zs, zc;
    if (q & 2) torch_box_muller(w[2], w[3], zs, zc);
    else torch_box_muller(w[0], w[1], zs, zc);
    return Vec<DT>::round((q & 1) ? zc : zs);
}

struct TorchLane {
    int64_t v[4];
    bool ok[4];
    uint32_t j, idx0;
};
// work item I of a tensor of nvec full vectors drawn by T threads
template <int N>
static __device__ __forceinline__ TorchLane torch_lane(int64_t I, int64_t T, int64_t nvec) {
    TorchLane L;
    const int64_t vpr = T / N, wpr = (vpr + 63) >> 6;
    const int64_t j = I / wpr, c = I - j * wpr;
    const int64_t vr = (c << 6) + (threadIdx.x & 63);
    L.j = (uint32_t)j;
    L.idx0 = (uint32_t)(vr * N);
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
        L.v[ii] = (4 * j + ii) * vpr + vr;
        L.ok[ii] = vr < vpr && L.v[ii] < nvec;
    }
    return L;
}
static inline int64_t torch_items(int64_t n, int64_t T, int N) {
    // full rounds take wpr items each; of the LAST round only the wave chunks that hold a vector
    // of its first row (the items behind them are empty: torch's own kernel draws and discards
    // there).  At least one item: item 0 also carries the ragged tail.
    const int64_t rounds = (n + 4 * T - 1) / (4 * T), vpr = T / N, wpr = (vpr + 63) / 64, nvec = n / N;
    if (rounds == 0) return 0;
    int64_t row0 = nvec - 4 * (rounds - 1) * vpr;          // vectors in the last round's first row
    if (row0 > vpr) row0 = vpr;
    int64_t last = row0 > 0 ? (row0 + 63) / 64 : 0;
    const int64_t items = (rounds - 1) * wpr + last;
    return items > 0 ? items : 1;
}
static inline int torch_threads_ok(int64_t n, int64_t T) {
    // T = 256 * grid, grid <= ceil(n / 256); 32-bit subsequence indices.  n < 2^31: a larger tensor
    // torch draws in several launches (TensorIterator::with_32bit_indexing), each at its own Philox
    // offset — not the single stream regenerated here; the caller materialises such a draw
    return n < ((int64_t)1 << 31) && T >= 256 && (T % 256) == 0 && T <= ((n + 255) / 256) * 256 &&
           T < ((int64_t)1 << 31);
}

#define ECO_XCD_ITEM(I, total)                                                                \
    const int64_t per__ = (int64_t)(gridDim.x / ECO_XCDS);                                     \
    const int64_t I = (int64_t)(blockIdx.x % ECO_XCDS) * per__ + blockIdx.x / ECO_XCDS;        \
    if (I >= (total)) return

// z = torch.normal's draw for (seed, n, dtype), materialised (tests, the start-up probe)
template <int DT>
__global__ __launch_bounds__(ECO_K1_THREADS) void zo_torch_fill_kernel(void* __restrict__ zout, int64_t n,
                                                                     int64_t T, int64_t items,
                                                                     uint32_t k0, uint32_t k1) {
    constexpr int N = Vec<DT>::N;
    ECO_XCD_ITEM(I, items);
    const int64_t nvec = n / N;
    const TorchLane L = torch_lane<N>(I, T, nvec);
    float z[4 * N];
    torch_z_tile<DT>(L.j, L.idx0, k0, k1, z);
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
        if (L.ok[ii]) st16(zout, L.v[ii], Vec<DT>::pack(z + ii * N));
    const int64_t tail0 = nvec * N;
    if (I == 0 && threadIdx.x < (unsigned)(n - tail0)) {
        const int64_t e = tail0 + threadIdx.x;
        Vec<DT>::store1(zout, e, torch_z1<DT>(e, T, k0, k1));
    }
}

// one reference K1 call, in place (P:473-486), z = torch's draw
template <int DT>
__global__ __launch_bounds__(ECO_K1_THREADS) void zo_torch_perturb_kernel(void* __restrict__ w, int64_t n,
                                                                        int64_t T, int64_t items, float sf,
                                                                        float eps, uint32_t k0, uint32_t k1) {
    constexpr int N = Vec<DT>::N;
    ECO_XCD_ITEM(I, items);
    const int64_t nvec = n / N;
    const TorchLane L = torch_lane<N>(I, T, nvec);
    u32x4 s[4];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) s[ii] = ld16_if(L.ok[ii], w, L.v[ii]);
    float z[4 * N];
    torch_z_tile<DT>(L.j, L.idx0, k0, k1, z);
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
        float wf[N];
        Vec<DT>::unpack(s[ii], wf);
#pragma unroll
        for (int i = 0; i < N; ++i) wf[i] = k1_step<DT>(wf[i], z[ii * N + i], sf, eps);
        if (L.ok[ii]) st16(w, L.v[ii], Vec<DT>::pack(wf));
    }
    const int64_t tail0 = nvec * N;
    if (I == 0 && threadIdx.x < (unsigned)(n - tail0)) {
        const int64_t e = tail0 + threadIdx.x;
        Vec<DT>::store1(w, e, k1_step<DT>(Vec<DT>::load1(w, e), torch_z1<DT>(e, T, k0, k1), sf, eps));
    }
}

// Block-batched K1 (every unit of every layer of a transformer block in ONE launch) with torch's
// draw in registers.  Device table, one row of ECO_LAYER_ROW_T int64 per layer:
//   [0] w_in  [1] w_final (may equal w_in: each lane reads its elements before it writes them)
//   [2] numel  [3] n_units  [4] first work item of the layer in the launch  [5] T
//   [6 .. 6+U) seeds   [6+U .. 6+2U) theta+ pointers   [6+2U .. 6+3U) theta- pointers
// A unit's theta+ may alias w_in as well (the triple form).  Algorithmic bytes:
// (2*U_owned + 2) * s per element — the same as the build's own in-register stream: no z bytes.
#define ECO_LAYER_ROW_T (6 + 3 * ECOFLAP_MAX_UNITS)

// Philox4x32-10 of counter {j, 0, idx, 0} given the first round's per-lane product (hi1, lo1) =
// M1 * idx: that product does not depend on the key, so a lane computes it once for all the
// units of its item.  (j is wave-uniform: its round-1 product and the round-2 product of
// hi(M0 * j) ^ k1 are scalar work; 16 vector multiplies per call are left of 20.)
static __device__ __forceinline__ void philox4x32_10_r1(uint32_t j, uint32_t hi1, uint32_t lo1, uint32_t k0,
                                                        uint32_t k1, uint32_t out[4]) {
    const uint64_t pj = (uint64_t)PHILOX_M0 * j;
    uint32_t c0 = hi1 ^ k0, c1 = lo1, c2 = (uint32_t)(pj >> 32) ^ k1, c3 = (uint32_t)pj;
    k0 += PHILOX_W0;
    k1 += PHILOX_W1;
#pragma unroll
    for (int r = 1; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        const uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
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

// global_* memory instructions for the table's pointers (the table holds int64: as generic
// pointers they compile to flat_*, which also count against lgkmcnt — the counter the unit
// loop's scalar loads of seeds and pointers wait on)
typedef __attribute__((address_space(1))) u32x4 g_u32x4;
static __device__ __forceinline__ u32x4 ldg16_if(bool ok, const void* p, int64_t v) {
    u32x4 r = {0u, 0u, 0u, 0u};
    if (ok) r = ((const g_u32x4*)p)[v];
    return r;
}
static __device__ __forceinline__ void stg16(void* p, int64_t v, const u32x4& x) { ((g_u32x4*)p)[v] = x; }
static __device__ __forceinline__ void stg16_nt(void* p, int64_t v, const u32x4& x) {
    __builtin_nontemporal_store(x, ((g_u32x4*)p) + v);
}

#ifndef ECO_K1_TORCH_WAVES
#define ECO_K1_TORCH_WAVES 5
#endif
// Workgroup -> work item of the block launch.  Workgroup b runs on XCD b % 8; XCD x takes the
// 64-item blocks x, x + 8, x + 16, ... of the launch's item sequence (64 items = 64 KiB of every
// stream of a row: long sequential runs per XCD, as the contiguous eighths of the other K1 kernels
// give) — interleaved, not one contiguous eighth each, because items are NOT of equal cost here:
// the light and empty items of a tensor's last round sit together at the end of each layer, and
// with contiguous eighths whole XCDs ran out of work early (a ViT-g block: XCDs 4 and 7 held
// 35-45 % light items; the launch took as long as the full ones, +12 %).
#ifndef ECO_K1_XCD_BLOCK_LOG2
#define ECO_K1_XCD_BLOCK_LOG2 6
#endif
static inline unsigned grid_items_blocked(int64_t items) {
    const int64_t q = (int64_t)ECO_XCDS << ECO_K1_XCD_BLOCK_LOG2;
    return (unsigned)((items + q - 1) / q * q);
}
template <int DT>
__global__ __launch_bounds__(ECO_K1_THREADS) __attribute__((amdgpu_waves_per_eu(ECO_K1_TORCH_WAVES)))
void zo_torch_layers_kernel(
    const int64_t* __restrict__ table, int n_layers, int64_t total_items, float eps) {
    constexpr int N = Vec<DT>::N;
    const int64_t slot = blockIdx.x / ECO_XCDS;
    const int64_t Ig = ((((slot >> ECO_K1_XCD_BLOCK_LOG2) * ECO_XCDS) + blockIdx.x % ECO_XCDS) << ECO_K1_XCD_BLOCK_LOG2)
                       + (slot & ((1 << ECO_K1_XCD_BLOCK_LOG2) - 1));
    if (Ig >= total_items) return;
    int l = 0;                                     // wave-uniform scan: scalar loads
    while (l + 1 < n_layers && Ig >= table[(int64_t)(l + 1) * ECO_LAYER_ROW_T + 4]) ++l;
    const int64_t* __restrict__ row = table + (int64_t)l * ECO_LAYER_ROW_T;
    const void* win = (const void*)row[0];
    void* wout = (void*)row[1];
    const int64_t n = row[2];
    const int n_units = (int)row[3];
    const int64_t I = Ig - row[4];
    const int64_t T = row[5];
    const int64_t nvec = n / N;
    const TorchLane L = torch_lane<N>(I, T, nvec);
    // Rows of the item that hold any vector of the tensor: a prefix of the four (row ii + 1 lies
    // T elements behind row ii) known from lane 0 (the wave's lowest vector of each row).  The
    // LAST round of a tensor is rarely full — a 6144 x 1408 matrix is 16.5 rows of T = 524288:
    // four full rounds and a fifth with half a row — and torch's own kernel draws and discards
    // there; this one skips what no element needs: the item itself when it is empty, the second
    // Box-Muller pair (rows 2 and 3) and the updates of absent rows.  Philox calls are per
    // (round, subsequence) and cannot be cut.
    int nv = 0;
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) nv += __builtin_amdgcn_readfirstlane((int)L.ok[ii]);
    const int64_t tail0 = nvec * N;                // ragged tail of this layer: its first wave
    const bool has_tail = I == 0 && n != tail0;
    if (nv == 0 && !has_tail) return;
    u32x4 s[4];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) s[ii] = ldg16_if(L.ok[ii], win, L.v[ii]);
    uint32_t hi1[N], lo1[N];                       // M1 * idx: the key-independent half of Philox round 1
#pragma unroll
    for (int t = 0; t < N; ++t) {
        const uint64_t p1 = (uint64_t)PHILOX_M1 * (L.idx0 + t);
        hi1[t] = (uint32_t)(p1 >> 32);
        lo1[t] = (uint32_t)p1;
    }
    for (int u = 0; u < n_units && nv > 0; ++u) {
        const uint64_t seed = (uint64_t)row[6 + u];
        const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
        float z[4 * N];
#pragma unroll
        for (int t = 0; t < N; ++t) {
            uint32_t w[4];
            philox4x32_10_r1(L.j, hi1[t], lo1[t], k0, k1, w);
            torch_box_muller(w[0], w[1], z[0 * N + t], z[1 * N + t]);
            if (nv > 2) torch_box_muller(w[2], w[3], z[2 * N + t], z[3 * N + t]);
        }
        void* dp = (void*)row[6 + ECOFLAP_MAX_UNITS + u];
        void* dm = (void*)row[6 + 2 * ECOFLAP_MAX_UNITS + u];
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            if (ii >= nv) break;                   // wave-uniform
            // ATen: static_cast<scalar_t>(...), one rounding to the storage dtype
            if constexpr (DT != ECOFLAP_F32) {
#pragma unroll
                for (int i = 0; i < N; i += 2)
                    Vec<DT>::round_pair(z[ii * N + i], z[ii * N + i + 1], z[ii * N + i], z[ii * N + i + 1]);
            }
            u32x4 p, m;
            unit_update<DT, true>(s[ii], z + ii * N, eps, p, m);
            if (dp && L.ok[ii]) {   // dp wave-uniform: not-owned units only carry the drift
                stg16_nt(dp, L.v[ii], p);
                stg16_nt(dm, L.v[ii], m);
            }
        }
    }
    // the drifted weights: in place they hit the lines the loads brought in; PARKED in a buffer of
    // their own (the block form: read back once, by a copy, after the layer's forwards) they are
    // one more write-once stream and go out non-temporal like the unit buffers (-6 % per launch)
    if (wout == win) {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
            if (L.ok[ii]) stg16(wout, L.v[ii], s[ii]);
    } else {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
            if (L.ok[ii]) stg16_nt(wout, L.v[ii], s[ii]);
    }
    if (has_tail && threadIdx.x < (unsigned)(n - tail0)) {
        const int64_t e = tail0 + threadIdx.x;
        float a = Vec<DT>::load1(win, e);
        for (int u = 0; u < n_units; ++u) {
            const uint64_t seed = (uint64_t)row[6 + u];
            const float z = torch_z1<DT>(e, T, (uint32_t)seed, (uint32_t)(seed >> 32));
            a = k1_step<DT>(a, z, 1.0f, eps);
            const float b = k1_step<DT>(a, z, -2.0f, eps);
            void* dp = (void*)row[6 + ECOFLAP_MAX_UNITS + u];
            if (dp) {
                Vec<DT>::store1(dp, e, a);
                Vec<DT>::store1((void*)row[6 + 2 * ECOFLAP_MAX_UNITS + u], e, b);
            }
            a = k1_step<DT>(b, z, 1.0f, eps);
        }
        Vec<DT>::store1(wout, e, a);
    }
}

// ---------------------------------------------------------------- launch helpers
// one single-wave workgroup per super-row of 128 vectors (two 16-byte vectors per lane),
// rounded up to a multiple of the XCD count for the remap in ECO_FOR_SUPER_ROWS
static inline unsigned grid_exact(int64_t nvec) {
    int64_t b = (nvec + 127) / 128;
    if (b < 1) b = 1;
    b = (b + ECO_XCDS - 1) / ECO_XCDS * ECO_XCDS;
    return (unsigned)b;
}
#define ECO_K1_MAX_ELEMS ((int64_t)1 << 40)   /* 2^31 workgroups of 128 16-byte vectors */


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
    if (n < 0 || n > ECO_K1_MAX_ELEMS) return ECOFLAP_ESIZE;
    if (n == 0) return 0;
    if (!w) return ECOFLAP_ENULL;
    if (!aligned16(w) || (z && !aligned16(z))) return ECOFLAP_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    DISPATCH_DT(dtype, {
        const unsigned g = grid_exact(n / Vec<DT>::N);
        if (z)
            hipLaunchKernelGGL((zo_perturb_kernel<DT, true>), dim3(g), dim3(ECO_K1_THREADS), 0, s, w, z, n,
                               scaling_factor, zo_eps, k0, k1);
        else
            hipLaunchKernelGGL((zo_perturb_kernel<DT, false>), dim3(g), dim3(ECO_K1_THREADS), 0, s, w, z, n,
                               scaling_factor, zo_eps, k0, k1);
    });
    ECO_CHECK_LAUNCH();
    return 0;
}

#define TRIPLE(HZ, PM)                                                                          \
    hipLaunchKernelGGL((zo_perturb_triple_kernel<DT, HZ, PM>), dim3(g), dim3(ECO_K1_THREADS), 0, s, w_in, \
                       w_plus, w_minus, w_restored, z, n, zo_eps, k0, k1)

extern "C" int ecoflap_zo_perturb_triple(const void* w_in, void* w_plus, void* w_minus,
                                         void* w_restored, int64_t n, int dtype, float zo_eps,
                                         uint64_t seed, const void* z, void* stream) {
    if (!dtype_ok(dtype)) return ECOFLAP_EDTYPE;
    if (n < 0 || n > ECO_K1_MAX_ELEMS) return ECOFLAP_ESIZE;
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

static int zo_perturb_units_impl(void* w, int64_t n, int dtype, float zo_eps, int n_units,
                                 const uint64_t* seeds, void* const* w_plus,
                                 void* const* w_minus, const void* const* z, void* stream,
                                 hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (!dtype_ok(dtype)) return ECOFLAP_EDTYPE;
    if (n < 0 || n > ECO_K1_MAX_ELEMS || n_units < 0 || n_units > ECOFLAP_MAX_UNITS) return ECOFLAP_ESIZE;
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
        // hipExtLaunchKernelGGL with null events is a plain launch; with events they take the
        // kernel's own begin / end timestamps (the dispatch's completion signal — what rocprofv3
        // reads), not the time markers queued around it take to be processed
        if (z)
            hipExtLaunchKernelGGL((zo_perturb_units_kernel<DT, true>), dim3(g), dim3(ECO_K1_THREADS), 0, s,
                                  ev_start, ev_stop, 0, w, n, zo_eps, n_units, tab);
        else
            hipExtLaunchKernelGGL((zo_perturb_units_kernel<DT, false>), dim3(g), dim3(ECO_K1_THREADS), 0, s,
                                  ev_start, ev_stop, 0, w, n, zo_eps, n_units, tab);
    });
    ECO_CHECK_LAUNCH();
    return 0;
}

extern "C" int ecoflap_zo_perturb_units(void* w, int64_t n, int dtype, float zo_eps, int n_units,
                                        const uint64_t* seeds, void* const* w_plus,
                                        void* const* w_minus, const void* const* z,
                                        void* stream) {
    return zo_perturb_units_impl(w, n, dtype, zo_eps, n_units, seeds, w_plus, w_minus, z, stream,
                                 nullptr, nullptr);
}

extern "C" int ecoflap_zo_perturb_units_timed(void* w, int64_t n, int dtype, float zo_eps,
                                              int n_units, const uint64_t* seeds,
                                              void* const* w_plus, void* const* w_minus,
                                              const void* const* z, void* stream,
                                              void* start_event, void* stop_event) {
    if (!start_event || !stop_event) return ECOFLAP_ENULL;
    return zo_perturb_units_impl(w, n, dtype, zo_eps, n_units, seeds, w_plus, w_minus, z, stream,
                                 (hipEvent_t)start_event, (hipEvent_t)stop_event);
}

template <bool HAS_Z>
static int zo_perturb_layers_impl(const int64_t* table, int n_layers, int64_t total_rows, int dtype,
                                  float zo_eps, void* stream, void* start_event, void* stop_event) {
    if (!dtype_ok(dtype)) return ECOFLAP_EDTYPE;
    if (n_layers < 0 || total_rows < 0 || total_rows > 0x7ffffff0LL) return ECOFLAP_ESIZE;
    if (n_layers == 0 || total_rows == 0) return 0;
    if (!table) return ECOFLAP_ENULL;
    if ((start_event == nullptr) != (stop_event == nullptr)) return ECOFLAP_ENULL;
    hipStream_t s = (hipStream_t)stream;
    const unsigned g = (unsigned)((total_rows + ECO_XCDS - 1) / ECO_XCDS * ECO_XCDS);
    DISPATCH_DT(dtype, {
        hipExtLaunchKernelGGL((zo_perturb_layers_kernel<DT, HAS_Z>), dim3(g), dim3(ECO_K1_THREADS), 0, s,
                              (hipEvent_t)start_event, (hipEvent_t)stop_event, 0, table, n_layers,
                              total_rows, zo_eps);
    });
    ECO_CHECK_LAUNCH();
    return 0;
}

extern "C" int ecoflap_zo_perturb_layers(const int64_t* table, int n_layers, int64_t total_rows,
                                         int dtype, float zo_eps, void* stream,
                                         void* start_event, void* stop_event) {
    return zo_perturb_layers_impl<false>(table, n_layers, total_rows, dtype, zo_eps, stream,
                                         start_event, stop_event);
}

extern "C" int ecoflap_zo_perturb_layers_z(const int64_t* table, int n_layers, int64_t total_rows,
                                           int dtype, float zo_eps, void* stream,
                                           void* start_event, void* stop_event) {
    return zo_perturb_layers_impl<true>(table, n_layers, total_rows, dtype, zo_eps, stream,
                                        start_event, stop_event);
}

// An empty launch through the same instrumented path: what the event pair of
// ecoflap_zo_perturb_units_timed reads for a kernel that does nothing (bench.py reports it next
// to the K1 durations; rocprofv3's dispatch timestamps of the same launches are shorter by
// about this much).
__global__ void null_kernel() {}
extern "C" int ecoflap_null_launch_timed(void* stream, void* start_event, void* stop_event) {
    if (!start_event || !stop_event) return ECOFLAP_ENULL;
    hipExtLaunchKernelGGL(null_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream,
                          (hipEvent_t)start_event, (hipEvent_t)stop_event, 0);
    ECO_CHECK_LAUNCH();
    return 0;
}

extern "C" int ecoflap_zo_fill_normal(void* z_out, int64_t n, int dtype, uint64_t seed,
                                      void* stream) {
    if (!dtype_ok(dtype)) return ECOFLAP_EDTYPE;
    if (n < 0 || n > ECO_K1_MAX_ELEMS) return ECOFLAP_ESIZE;
    if (n == 0) return 0;
    if (!z_out) return ECOFLAP_ENULL;
    if (!aligned16(z_out)) return ECOFLAP_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    DISPATCH_DT(dtype, {
        hipLaunchKernelGGL((zo_fill_normal_kernel<DT>), dim3(grid_exact(n / Vec<DT>::N)), dim3(ECO_K1_THREADS),
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

// ---------------------------------------------------------------- torch's stream: entry points
extern "C" int64_t ecoflap_torch_normal_threads(int64_t n, int multiprocessors, int max_threads_per_mp) {
    // ATen calc_execution_policy (DistributionTemplates.h:50-62): block 256
    if (n <= 0 || multiprocessors <= 0 || max_threads_per_mp < 256) return 0;
    int64_t grid = (n + 255) / 256;
    const int64_t cap = (int64_t)multiprocessors * (max_threads_per_mp / 256);
    if (grid > cap) grid = cap;
    return grid * 256;
}

extern "C" int64_t ecoflap_torch_layer_items(int64_t n, int64_t threads, int dtype) {
    if (!dtype_ok(dtype) || n <= 0 || !torch_threads_ok(n, threads)) return 0;
    return torch_items(n, threads, dtype == ECOFLAP_F32 ? 4 : 8);
}

static inline unsigned grid_items(int64_t items) {
    return (unsigned)((items + ECO_XCDS - 1) / ECO_XCDS * ECO_XCDS);
}

extern "C" int ecoflap_zo_fill_normal_torch(void* z_out, int64_t n, int dtype, uint64_t seed,
                                            int64_t threads, void* stream) {
    if (!dtype_ok(dtype)) return ECOFLAP_EDTYPE;
    if (n < 0 || n > ECO_K1_MAX_ELEMS) return ECOFLAP_ESIZE;
    if (n == 0) return 0;
    if (!torch_threads_ok(n, threads)) return ECOFLAP_ESIZE;
    if (!z_out) return ECOFLAP_ENULL;
    if (!aligned16(z_out)) return ECOFLAP_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    DISPATCH_DT(dtype, {
        const int64_t items = torch_items(n, threads, Vec<DT>::N);
        hipLaunchKernelGGL((zo_torch_fill_kernel<DT>), dim3(grid_items(items)), dim3(ECO_K1_THREADS), 0, s,
                           z_out, n, threads, items, k0, k1);
    });
    ECO_CHECK_LAUNCH();
    return 0;
}

extern "C" int ecoflap_zo_perturb_torch(void* w, int64_t n, int dtype, float scaling_factor,
                                        float zo_eps, uint64_t seed, int64_t threads, void* stream) {
    if (!dtype_ok(dtype)) return ECOFLAP_EDTYPE;
    if (n < 0 || n > ECO_K1_MAX_ELEMS) return ECOFLAP_ESIZE;
    if (n == 0) return 0;
    if (!torch_threads_ok(n, threads)) return ECOFLAP_ESIZE;
    if (!w) return ECOFLAP_ENULL;
    if (!aligned16(w)) return ECOFLAP_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    DISPATCH_DT(dtype, {
        const int64_t items = torch_items(n, threads, Vec<DT>::N);
        hipLaunchKernelGGL((zo_torch_perturb_kernel<DT>), dim3(grid_items(items)), dim3(ECO_K1_THREADS), 0, s,
                           w, n, threads, items, scaling_factor, zo_eps, k0, k1);
    });
    ECO_CHECK_LAUNCH();
    return 0;
}

extern "C" int ecoflap_zo_perturb_layers_torch(const int64_t* table, int n_layers, int64_t total_items,
                                               int dtype, float zo_eps, void* stream,
                                               void* start_event, void* stop_event) {
    if (!dtype_ok(dtype)) return ECOFLAP_EDTYPE;
    if (n_layers < 0 || total_items < 0 || total_items > 0x7ffffff0LL) return ECOFLAP_ESIZE;
    if (n_layers == 0 || total_items == 0) return 0;
    if (!table) return ECOFLAP_ENULL;
    if ((start_event == nullptr) != (stop_event == nullptr)) return ECOFLAP_ENULL;
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_DT(dtype, {
        hipExtLaunchKernelGGL((zo_torch_layers_kernel<DT>), dim3(grid_items_blocked(total_items)), dim3(ECO_K1_THREADS),
                              0, s, (hipEvent_t)start_event, (hipEvent_t)stop_event, 0, table, n_layers,
                              total_items, zo_eps);
    });
    ECO_CHECK_LAUNCH();
    return 0;
}

// every word of [first, first + count): the shipped radius against rocRAND's restated one
__global__ __launch_bounds__(256) void torch_radius_sweep_kernel(uint64_t first, uint64_t count,
                                                                  unsigned long long* __restrict__ differ) {
    unsigned long long bad = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const uint32_t x = (uint32_t)(first + i);
        bad += __float_as_uint(torch_radius(x)) != __float_as_uint(torch_radius_reference(x));
    }
    bad = wave_sum(bad);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(differ, bad);
}

extern "C" int ecoflap_zo_torch_radius_sweep(uint64_t first_word, uint64_t n_words,
                                             unsigned long long* differ, void* stream) {
    if (first_word > 0xffffffffull || n_words > (1ull << 32) - first_word) return ECOFLAP_ESIZE;
    if (n_words == 0) return 0;
    if (!differ) return ECOFLAP_ENULL;
    uint64_t blocks = (n_words + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(torch_radius_sweep_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       first_word, n_words, differ);
    ECO_CHECK_LAUNCH();
    return 0;
}
