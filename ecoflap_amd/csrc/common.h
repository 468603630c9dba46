// common.h — shared device helpers for the gfx950 ECoFLaP kernels.
// Wave = 64 lanes; every global access in the hot kernels is a 16-byte vector
// (4 x f32 or 8 x f16/bf16 per lane, 1 KiB per wave instruction).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ecoflap_hip.h"

#define ECO_WAVE 64

#define ECO_CHECK_LAUNCH()                         \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// ---- storage dtype traits: 16-byte vector <-> fp32 lanes -------------------------------
// bf16: plain casts (hipcc emits v_cvt_pk_bf16_f32: round-to-nearest-even, NaN-preserving;
// gfx950 has no bf16 VALU arithmetic to narrow into).  f16: explicit v_cvt asm, see below.
template <int DT> struct Vec;

template <> struct Vec<ECOFLAP_F32> {
    static constexpr int N = 4;       // elements per 16-byte vector
    static constexpr int BYTES = 4;
    typedef float scalar_t;
    static __device__ __forceinline__ void unpack(const u32x4& v, float* f) {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(v[i]);
    }
    static __device__ __forceinline__ u32x4 pack(const float* f) {
        u32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = __float_as_uint(f[i]);
        return v;
    }
    static __device__ __forceinline__ float round(float x) { return x; }
    static __device__ __forceinline__ float load1(const void* p, int64_t i) {
        return ((const float*)p)[i];
    }
    static __device__ __forceinline__ void store1(void* p, int64_t i, float x) {
        ((float*)p)[i] = x;
    }
};

template <> struct Vec<ECOFLAP_F16> {
    static constexpr int N = 8;
    static constexpr int BYTES = 2;
    typedef _Float16 scalar_t;
    static __device__ __forceinline__ float h2f(uint32_t bits) {
        uint16_t b = (uint16_t)bits;
        return (float)__builtin_bit_cast(_Float16, b);
    }
    // f32 -> f16 through an explicit v_cvt instruction.  A plain cast lets hipcc fold
    // `(half)(a_f32 * b_f32)` into v_fma_mixlo_f16, which rounds the EXACT product once;
    // torch rounds the product to f32 first and then to f16 (the reference's arithmetic,
    // differs at ties ~1e-4 of elements).  The asm operand forces the f32 result to exist.
    static __device__ __forceinline__ uint32_t f2h(float x) {
        uint32_t h;
        asm("v_cvt_f16_f32_e32 %0, %1" : "=v"(h) : "v"(x));
        return h & 0xffffu;
    }
    static __device__ __forceinline__ uint32_t f2h_pk(float lo, float hi) {
        uint32_t h;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(lo), "v"(hi));
        return h;
    }
    static __device__ __forceinline__ void unpack(const u32x4& v, float* f) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = h2f(v[i] & 0xffffu);
            f[2 * i + 1] = h2f(v[i] >> 16);
        }
    }
    static __device__ __forceinline__ u32x4 pack(const float* f) {
        u32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = f2h_pk(f[2 * i], f[2 * i + 1]);
        return v;
    }
    static __device__ __forceinline__ float round(float x) { return h2f(f2h(x)); }
    // round two values with ONE packed convert; returns the packed pair (ready to store)
    static __device__ __forceinline__ uint32_t round_pair(float x0, float x1, float& r0, float& r1) {
        typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
        const uint32_t pk = f2h_pk(x0, x1);
        const h2_t h = __builtin_bit_cast(h2_t, pk);     // (element extracts: a consumer's fma can take the half directly)
        r0 = (float)h[0];
        r1 = (float)h[1];
        return pk;
    }
    static __device__ __forceinline__ float load1(const void* p, int64_t i) {
        return (float)((const _Float16*)p)[i];
    }
    static __device__ __forceinline__ void store1(void* p, int64_t i, float x) {
        ((uint16_t*)p)[i] = (uint16_t)f2h(x);
    }
};

template <> struct Vec<ECOFLAP_BF16> {
    static constexpr int N = 8;
    static constexpr int BYTES = 2;
    typedef __bf16 scalar_t;
    static __device__ __forceinline__ uint32_t f2b(float x) {
        return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)x);
    }
    static __device__ __forceinline__ void unpack(const u32x4& v, float* f) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(v[i] << 16);
            f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ u32x4 pack(const float* f) {
        u32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = f2b(f[2 * i]) | (f2b(f[2 * i + 1]) << 16);
        return v;
    }
    static __device__ __forceinline__ float round(float x) {
        return __uint_as_float(f2b(x) << 16);
    }
    // x0 -> low half, x1 -> high half, round-to-nearest-even, NaN-preserving.  Explicit asm: the
    // compiler only sometimes fuses two scalar casts into the packed convert, and its SLP
    // vectoriser then re-pairs the operands and repacks the halves with extra SDWA ops.
    static __device__ __forceinline__ uint32_t f2b_pk(float x0, float x1) {
        uint32_t pk;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(x0), "v"(x1));
        return pk;
    }
    static __device__ __forceinline__ uint32_t round_pair(float x0, float x1, float& r0, float& r1) {
        const uint32_t pk = f2b_pk(x0, x1);
        r0 = __uint_as_float(pk << 16);
        r1 = __uint_as_float(pk & 0xffff0000u);
        return pk;
    }
    static __device__ __forceinline__ float load1(const void* p, int64_t i) {
        return __uint_as_float((uint32_t)((const uint16_t*)p)[i] << 16);
    }
    static __device__ __forceinline__ void store1(void* p, int64_t i, float x) {
        ((uint16_t*)p)[i] = (uint16_t)f2b(x);
    }
};

static __device__ __forceinline__ u32x4 ld16(const void* p, int64_t vec_index) {
    return ((const u32x4*)p)[vec_index];
}
static __device__ __forceinline__ void st16(void* p, int64_t vec_index, const u32x4& v) {
    ((u32x4*)p)[vec_index] = v;
}

// ---- wave / block reductions (64-wide shuffles, then LDS across the block's waves) ------
template <typename T>
static __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// sum over a 256-thread block; result valid in thread 0
static __device__ __forceinline__ double block_sum_256(double v, double* lds4) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) lds4[wave] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) r = ((lds4[0] + lds4[1]) + (lds4[2] + lds4[3]));
    return r;
}

static inline int dtype_ok(int dt) {
    return dt == ECOFLAP_F32 || dt == ECOFLAP_F16 || dt == ECOFLAP_BF16;
}
static inline size_t dtype_bytes(int dt) { return dt == ECOFLAP_F32 ? 4 : 2; }
static inline int aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
