// cholesky.hip — blocked fp32 Cholesky for SparseGPT's damped factorisations on gfx950.
//
// Replaces the two `torch.linalg.cholesky` calls of `SparseGPT.fasterprune`
//   (LAVIS/lavis/compression/pruners/sparsegpt_pruner.py:113-123 lower, :146-155 upper) —
// rocSOLVER's potrf on this stack is bound by the latency of its panel steps, not by flops or
// bytes: 4.0 / 6.0 / 16.3 / 20.3 ms at n = 1408 / 2048 / 5120 / 6144 (0.2 - 3.8 TFLOP/s,
// profiles/r05_sparsegpt/cholesky_bench.log), half of SparseGPT's stage 2 on the BLIP-2 shape —
// and two of its calls in flight in one process corrupt each other (profiles/r05_sparsegpt/README.md).
// This one keeps no state outside its arguments: no handle, no workspace.
//
// Right-looking, 64 columns per step, in place on the LOWER triangle of a row-major matrix:
//   panel launch     one single-wave workgroup per 64 rows from the diagonal block down.  EVERY
//                    wave factors the 64 x 64 diagonal block for itself (a row per lane, the row
//                    in registers, 64 dependent column steps; nobody waits for anybody: no flag,
//                    no barrier of any kind), wave 0 writes it back, the others solve their 64
//                    rows against it (X L^T = B, a row per lane, L^T's rows broadcast from LDS);
//   trailing launch  C[i, j] -= sum_t L[i, k0 + t] L[j, k0 + t] on the tiles i >= j of the
//                    trailing matrix: 64 x 64 tiles, four waves of one 32 x 32
//                    v_mfma_f32_32x32x2_f32 tile each, the two 64 x 64 panel slabs through LDS.
// Two launches per step on one stream; the host never waits.  Every sum is taken in a fixed order
// (bit-repeatable; two factorisations may run side by side on two streams: they share nothing).
// A pivot that is not positive (or not a number) is reported through *info (LAPACK's convention:
// the 1-based index of the first such column) and replaced by 1 so that the launch sequence ends
// with finite garbage instead of a NaN storm; the caller must check info.
// `upper`: the factor is computed as above (the input is symmetric: its lower triangle is read)
// and transposed at the end; the other triangle of the result is zero in both forms, as torch's.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CH_NB 64
#define CH_LD 65                // pitch of the diagonal block's LDS images (odd: conflict-free columns)
#define CH_PK 66                // pitch of the panel slabs (8-byte operand reads on 32 distinct even banks)

// ---- panel: one WAVE per 64 rows, a row per lane, the row in registers --------------------------
// Every wave factors the 64 x 64 diagonal block for itself (lane i holds row i; column j of the
// factor is scaled by the pivot read out of lane j's register; the rank-1 update takes L[c][j]
// out of lane c with v_readlane_b32, as a scalar operand) and then — except the wave of the
// diagonal block itself, which stores the factor — solves its 64 rows against it, column by
// column (x_t = b_t / L_tt, b_j -= x_t L_jt for j > t: the 63 - t updates are independent, L_jt
// again out of lane j's registers).  No LDS, no barrier, no wait of any kind inside the steps.
static __device__ __forceinline__ float lane_value(float v, int l) {     // v of lane l, wave-uniform (SGPR)
    return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), l));
}

// a[C0 + k] += m * (src of lane C0 + k), k = 0 .. N-1: N v_readlane_b32 into scalar registers, then N
// v_fmac_f32 with the scalar as an operand — written out by hand: left to the compiler the same
// source became 26 000 instructions (one s_nop per readlane for the scalar-write hazard, which a
// batch of eight needs none of, and thousands of scalar spills through v_writelane_b32).
template <int C0>
static __device__ __forceinline__ void fmac8_lanes(float (&a)[CH_NB], float src, float m) {
    asm volatile(
        "v_readlane_b32 s20, %8, %10\n\tv_readlane_b32 s21, %8, %11\n\tv_readlane_b32 s22, %8, %12\n\t"
        "v_readlane_b32 s23, %8, %13\n\tv_readlane_b32 s24, %8, %14\n\tv_readlane_b32 s25, %8, %15\n\t"
        "v_readlane_b32 s26, %8, %16\n\tv_readlane_b32 s27, %8, %17\n\t"
        "v_fmac_f32 %0, s20, %9\n\tv_fmac_f32 %1, s21, %9\n\tv_fmac_f32 %2, s22, %9\n\tv_fmac_f32 %3, s23, %9\n\t"
        "v_fmac_f32 %4, s24, %9\n\tv_fmac_f32 %5, s25, %9\n\tv_fmac_f32 %6, s26, %9\n\tv_fmac_f32 %7, s27, %9"
        : "+v"(a[C0]), "+v"(a[C0 + 1]), "+v"(a[C0 + 2]), "+v"(a[C0 + 3]), "+v"(a[C0 + 4]), "+v"(a[C0 + 5]),
          "+v"(a[C0 + 6]), "+v"(a[C0 + 7])
        : "v"(src), "v"(m), "n"(C0), "n"(C0 + 1), "n"(C0 + 2), "n"(C0 + 3), "n"(C0 + 4), "n"(C0 + 5), "n"(C0 + 6),
          "n"(C0 + 7)
        : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
}
template <int C>
static __device__ __forceinline__ void fmac1_lane(float (&a)[CH_NB], float src, float m) {
    asm volatile("v_readlane_b32 s20, %1, %3\n\ts_nop 3\n\tv_fmac_f32 %0, s20, %2"
                 : "+v"(a[C]) : "v"(src), "v"(m), "n"(C) : "s20");
}
// a[c] += m * (src of lane c) for c = LO .. 63
template <int LO>
static __device__ __forceinline__ void fmac_lanes_from(float (&a)[CH_NB], float src, float m) {
    if constexpr (LO < CH_NB) {
        if constexpr (LO + 8 <= CH_NB) {
            fmac8_lanes<LO>(a, src, m);
            fmac_lanes_from<LO + 8>(a, src, m);
        } else {
            fmac1_lane<LO>(a, src, m);
            fmac_lanes_from<LO + 1>(a, src, m);
        }
    }
}

// The 64 column steps as template instances: every register-array index is a constant in the
// source.  No LDS and no waits: a value another lane holds comes through v_readlane_b32 into a
// scalar register and is a scalar operand of the v_fmac that uses it.
template <int J>
struct FactorStep {
    static __device__ __forceinline__ void run(float (&a)[CH_NB], int lane, int& bad) {
        float d = lane_value(a[J], J);                       // the pivot
        // (selects, not branches: one basic block for the whole factorisation)
        const bool ok = d > 0.0f;
        bad = (!ok && bad == 0) ? J + 1 : bad;
        d = ok ? d : 1.0f;
        const float s = sqrtf(d);
        float quo = a[J] / s;          // every lane divides: the empty asm keeps the compiler from turning
        asm volatile("" : "+v"(quo));  // the select below back into a branch around the division
        const float l = lane > J ? quo : (lane == J ? s : 0.0f);
        a[J] = l;                      // row `lane` of the factor grows in place: a[J] = L[lane][J]
        // a[c] -= L[lane][J] L[c][J] for c > J (entries right of the diagonal are never read)
        fmac_lanes_from<J + 1>(a, l, -l);
        FactorStep<J + 1>::run(a, lane, bad);
    }
};
template <>
struct FactorStep<CH_NB> {
    static __device__ __forceinline__ void run(float (&)[CH_NB], int, int&) {}
};

// X L^T = B by columns: x_t = b_t / L[t][t], b_j -= x_t L[j][t] for j > t; L[j][t] is a[t] of lane j
template <int T>
struct SolveStep {
    static __device__ __forceinline__ void run(float (&b)[CH_NB], const float (&a)[CH_NB]) {
        const float x = b[T] / lane_value(a[T], T);
        b[T] = x;
        fmac_lanes_from<T + 1>(b, a[T], -x);
        SolveStep<T + 1>::run(b, a);
    }
};
template <>
struct SolveStep<CH_NB> {
    static __device__ __forceinline__ void run(float (&)[CH_NB], const float (&)[CH_NB]) {}
};

__global__ __launch_bounds__(64) void chol_panel_kernel(float* __restrict__ A, int64_t n, int64_t lda,
                                                        int64_t k0, int* __restrict__ info) {
    // (host: k0 + 64 <= n — a ragged last block goes to chol_ragged_kernel)
    const int lane = threadIdx.x;
    const bool vec = (lda & 3) == 0 && (((uintptr_t)A) & 15) == 0;      // (k0 is a multiple of 64)
    const int64_t last = n - 1;
    float a[CH_NB];
    {
        const float* p = A + (k0 + lane) * lda + k0;
        if (vec) {
#pragma unroll
            for (int v = 0; v < CH_NB / 4; ++v) {
                const f32x4 q = *(const f32x4*)(p + 4 * v);
#pragma unroll
                for (int e = 0; e < 4; ++e) a[4 * v + e] = q[e];
            }
        } else {
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) a[c] = p[c];
        }
    }
    int bad = 0;
    FactorStep<0>::run(a, lane, bad);
    if (blockIdx.x == 0) {
        if (bad && lane == 0 && info[0] == 0) info[0] = (int)(k0 + bad);
        float* p = A + (k0 + lane) * lda + k0;
#pragma unroll
        for (int c = 0; c < CH_NB; ++c)
            if (c <= lane) p[c] = a[c];
        return;
    }
    // rows r0 .. r0 + 63 of the panel: X L^T = B (rows past n: a clamped row is solved and not stored)
    const int64_t gr = k0 + (int64_t)blockIdx.x * CH_NB + lane;
    float b[CH_NB];
    {
        const float* p = A + (gr < n ? gr : last) * lda + k0;
        if (vec) {
#pragma unroll
            for (int v = 0; v < CH_NB / 4; ++v) {
                const f32x4 q = *(const f32x4*)(p + 4 * v);
#pragma unroll
                for (int e = 0; e < 4; ++e) b[4 * v + e] = q[e];
            }
        } else {
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) b[c] = p[c];
        }
    }
    SolveStep<0>::run(b, a);
    if (gr < n) {
        float* p = A + gr * lda + k0;
        if (vec) {
#pragma unroll
            for (int v = 0; v < CH_NB / 4; ++v) *(f32x4*)(p + 4 * v) = f32x4{b[4 * v], b[4 * v + 1], b[4 * v + 2], b[4 * v + 3]};
        } else {
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) p[c] = b[c];
        }
    }
}

// ---- a ragged last block (n % 64 != 0): the same factorisation by 256 threads through LDS -----------
// Runs once per factorisation at most (nothing lies below the last block: no solve).
__global__ __launch_bounds__(256) void chol_ragged_kernel(float* __restrict__ A, int64_t n, int64_t lda,
                                                          int64_t k0, int* __restrict__ info) {
    __shared__ float D[CH_NB][CH_LD];
    __shared__ float L[CH_NB][CH_LD];
    const int tid = threadIdx.x;
    for (int e = tid; e < CH_NB * CH_NB; e += 256) {
        const int r = e >> 6, c = e & 63;
        const int64_t gr = k0 + r, gc = k0 + c;
        D[r][c] = (gr < n && gc < n) ? (c <= r ? A[gr * lda + gc] : 0.0f) : (r == c ? 1.0f : 0.0f);
    }
    int bad = 0;
    for (int j = 0; j < CH_NB; ++j) {
        __syncthreads();
        float d = D[j][j];
        if (!(d > 0.0f)) {
            if (!bad) bad = j + 1;
            d = 1.0f;
        }
        const float s = sqrtf(d), rd = 1.0f / d;
        // column j of the factor goes to its own image: the updates below read the unscaled column
        if (tid < CH_NB) L[tid][j] = tid > j ? D[tid][j] / s : (tid == j ? s : 0.0f);
        for (int e = tid; e < CH_NB * CH_NB; e += 256) {
            const int i = e >> 6, c = e & 63;
            if (i > j && c > j && c <= i) D[i][c] -= (D[i][j] * rd) * D[c][j];
        }
    }
    __syncthreads();
    if (bad && tid == 0 && info[0] == 0) info[0] = (int)(k0 + bad);
    for (int e = tid; e < CH_NB * CH_NB; e += 256) {
        const int r = e >> 6, c = e & 63;
        const int64_t gr = k0 + r, gc = k0 + c;
        if (gr < n && gc < n && c <= r) A[gr * lda + gc] = L[r][c];
    }
}

// ---- trailing update on the matrix cores --------------------------------------------------------
__global__ __launch_bounds__(256) void chol_trailing_kernel(float* __restrict__ A, int64_t n, int64_t lda,
                                                            int64_t k0) {
    const int I = blockIdx.y, J = blockIdx.x;
    if (J > I) return;
    __shared__ __attribute__((aligned(16))) float Pa[CH_NB * CH_PK];
    __shared__ __attribute__((aligned(16))) float Pb[CH_NB * CH_PK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t t0 = k0 + CH_NB;
    const int64_t r0 = t0 + (int64_t)I * CH_NB, c0 = t0 + (int64_t)J * CH_NB;
    // slabs L[r0 .. +63, k0 .. k0 + 63] and L[c0 .. +63, k0 .. k0 + 63]: rows past n read as zeros
    const bool vec = (lda & 3) == 0 && (((uintptr_t)A) & 15) == 0;     // (k0 is a multiple of 64)
    for (int e = tid; e < CH_NB * (CH_NB / 4); e += 256) {
        const int r = e >> 4, v = e & 15;
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        if (r0 + r < n) {
            const float* p = A + (r0 + r) * lda + k0 + 4 * v;
            if (vec) {
                a = *(const f32x4*)p;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) a[q] = p[q];
            }
        }
        if (c0 + r < n) {
            const float* p = A + (c0 + r) * lda + k0 + 4 * v;
            if (vec) {
                b = *(const f32x4*)p;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) b[q] = p[q];
            }
        }
        float* da = &Pa[r * CH_PK + 4 * v];
        float* db = &Pb[r * CH_PK + 4 * v];
        *(f32x2*)da = f32x2{a[0], a[1]};
        *(f32x2*)(da + 2) = f32x2{a[2], a[3]};
        *(f32x2*)db = f32x2{b[0], b[1]};
        *(f32x2*)(db + 2) = f32x2{b[2], b[3]};
    }
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int wy = wave >> 1, wx = wave & 1;
    const int r32 = lane & 31, kh = lane >> 5;
    const float* Ap = &Pa[(32 * wy + r32) * CH_PK + 2 * kh];
    const float* Bp = &Pb[(32 * wx + r32) * CH_PK + 2 * kh];
#pragma unroll
    for (int q = 0; q < CH_NB / 4; ++q) {
        const f32x2 av = *(const f32x2*)(Ap + 4 * q);
        const f32x2 bv = *(const f32x2*)(Bp + 4 * q);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[1], acc, 0, 0, 0);
    }
    // C/D map of the 32 x 32 shapes: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const int64_t col = c0 + 32 * wx + r32;
    if (col >= n) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = r0 + 32 * wy + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row < n && col <= row) A[row * lda + col] -= acc[r];
    }
}

// ---- the other triangle ---------------------------------------------------------------------------
// lower: zero the strict upper triangle.  upper: U = L^T, the strict lower triangle zeroed.
__global__ __launch_bounds__(256) void chol_finalize_kernel(float* __restrict__ A, int64_t n, int64_t lda,
                                                            int upper) {
    const int I = blockIdx.y, J = blockIdx.x;       // tile (I, J) of the LOWER triangle, J <= I
    if (J > I) return;
    __shared__ float T[CH_NB][CH_LD];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)I * CH_NB, c0 = (int64_t)J * CH_NB;
    if (!upper) {
        if (I == J) {
            for (int e = tid; e < CH_NB * CH_NB; e += 256) {
                const int r = e >> 6, c = e & 63;
                if (c > r && r0 + r < n && c0 + c < n) A[(r0 + r) * lda + c0 + c] = 0.0f;
            }
        } else {                                      // its mirror tile (J, I) lies above the diagonal
            for (int e = tid; e < CH_NB * CH_NB; e += 256) {
                const int r = e >> 6, c = e & 63;
                if (c0 + r < n && r0 + c < n) A[(c0 + r) * lda + r0 + c] = 0.0f;
            }
        }
        return;
    }
    for (int e = tid; e < CH_NB * CH_NB; e += 256) {
        const int r = e >> 6, c = e & 63;
        const bool in = r0 + r < n && c0 + c < n && (I != J || c <= r);
        T[r][c] = in ? A[(r0 + r) * lda + c0 + c] : 0.0f;
    }
    __syncthreads();
    for (int e = tid; e < CH_NB * CH_NB; e += 256) {
        const int r = e >> 6, c = e & 63;             // element (c0 + r, r0 + c) of the upper triangle
        if (c0 + r < n && r0 + c < n) A[(c0 + r) * lda + r0 + c] = (I != J || c >= r) ? T[c][r] : 0.0f;
        if (I != J && r0 + r < n && c0 + c < n) A[(r0 + r) * lda + c0 + c] = 0.0f;   // (r, c) walks the lower tile too
    }
}

extern "C" int ecoflap_cholesky_f32(float* a, int64_t n, int64_t lda, int upper, int* info, void* stream) {
    if (n < 0 || lda < n || n > (1 << 20)) return ECOFLAP_ESIZE;
    if (!info) return ECOFLAP_ENULL;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(info, 0, sizeof(int), s) != hipSuccess) return ECOFLAP_ENULL;
    if (n == 0) return 0;
    if (!a) return ECOFLAP_ENULL;
    for (int64_t k0 = 0; k0 < n; k0 += CH_NB) {
        const unsigned chunks = (unsigned)((n - k0 + CH_NB - 1) / CH_NB);
        if (k0 + CH_NB <= n)
            hipLaunchKernelGGL(chol_panel_kernel, dim3(chunks), dim3(64), 0, s, a, n, lda, k0, info);
        else
            hipLaunchKernelGGL(chol_ragged_kernel, dim3(1), dim3(256), 0, s, a, n, lda, k0, info);
        if (chunks > 1)
            hipLaunchKernelGGL(chol_trailing_kernel, dim3(chunks - 1, chunks - 1), dim3(256), 0, s, a, n, lda, k0);
    }
    const unsigned tiles = (unsigned)((n + CH_NB - 1) / CH_NB);
    hipLaunchKernelGGL(chol_finalize_kernel, dim3(tiles, tiles), dim3(256), 0, s, a, n, lda, upper ? 1 : 0);
    ECO_CHECK_LAUNCH();
    return 0;
}
