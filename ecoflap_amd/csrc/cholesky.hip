// cholesky.hip — blocked fp32 Cholesky for SparseGPT's damped factorisations on gfx950.
//
// Replaces the two `torch.linalg.cholesky` calls of `SparseGPT.fasterprune`
//   (LAVIS/lavis/compression/pruners/sparsegpt_pruner.py:113-123 lower, :146-155 upper) —
// rocSOLVER's potrf on this stack is bound by the latency of its panel steps, not by flops or
// bytes: 4.0 / 6.0 / 16.3 / 20.3 ms at n = 1408 / 2048 / 5120 / 6144 (0.2 - 3.8 TFLOP/s,
// profiles/r05_sparsegpt/cholesky_bench.log), half of SparseGPT's stage 2 on the BLIP-2 shape —
// and two of its calls in flight in one process corrupt each other (profiles/r05_sparsegpt/README.md).
// This one keeps no state outside its arguments: no handle; 16 KB of caller-owned scratch per call.
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
                                                        int64_t k0, int* __restrict__ info,
                                                        float* __restrict__ Dk) {
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
        // The factored block must NOT go into A while the other waves of this launch may still be
        // loading the unfactored one (they start when the chip has room for them: with several
        // factorisations in flight some start after wave 0 is done — measured: one factor in ~12
        // came out different side by side): it waits in the call's own 16 KB scratch and the
        // trailing launch — which starts when this one is over — puts it in place.  A launch of
        // one wave (the last block) writes in place.
        float* p = gridDim.x == 1 ? A + (k0 + lane) * lda + k0 : Dk + lane * CH_NB;
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
                                                            int64_t k0, const float* __restrict__ Dk) {
    const int I = blockIdx.y, J = blockIdx.x;
    if (J > I) return;
    if (I == 0 && J == 0) {       // the panel launch's factored diagonal block goes home (see there)
        for (int e = threadIdx.x; e < CH_NB * CH_NB; e += 256) {
            const int r = e >> 6, c = e & 63;
            if (c <= r) A[(k0 + r) * lda + k0 + c] = Dk[r * CH_NB + c];
        }
    }
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

// ================================================================================================
// The inverse from the factor: out = (L L^T)^-1 = X^T X with X = L^-1   (torch.cholesky_inverse,
// sparsegpt_pruner.py:134; rocSOLVER's potri: 7.2 / 13.0 ms at n = 5120 / 6144)
//   1. the 64 x 64 diagonal blocks of X: one wave each, rows in registers, as the panel solve;
//   2. the blocks below them by doubling: with X11 = inv(L11), X22 = inv(L22) known for two
//      neighbouring diagonal blocks of size s, X21 = -X22 (L21 X11): two launches of tile GEMMs
//      per level, log2(n / 64) levels, every pair of a level in one launch;
//   3. out = X^T X on the tiles i >= j (k from the later of the two block rows), mirrored.
// Every product on v_mfma_f32_32x32x2_f32, 64 x 64 tiles, K in chunks of 64 through LDS, every
// sum in a fixed order.
template <int T, int S0>
static __device__ __forceinline__ void fmac8_regs(float (&b)[CH_NB], const float (&a)[CH_NB], float m) {
    // b[S0 + k] += m * (a[S0 + k] of lane T), k = 0 .. 7
    asm volatile(
        "v_readlane_b32 s20, %8, %17\n\tv_readlane_b32 s21, %9, %17\n\tv_readlane_b32 s22, %10, %17\n\t"
        "v_readlane_b32 s23, %11, %17\n\tv_readlane_b32 s24, %12, %17\n\tv_readlane_b32 s25, %13, %17\n\t"
        "v_readlane_b32 s26, %14, %17\n\tv_readlane_b32 s27, %15, %17\n\t"
        "v_fmac_f32 %0, s20, %16\n\tv_fmac_f32 %1, s21, %16\n\tv_fmac_f32 %2, s22, %16\n\tv_fmac_f32 %3, s23, %16\n\t"
        "v_fmac_f32 %4, s24, %16\n\tv_fmac_f32 %5, s25, %16\n\tv_fmac_f32 %6, s26, %16\n\tv_fmac_f32 %7, s27, %16"
        : "+v"(b[S0]), "+v"(b[S0 + 1]), "+v"(b[S0 + 2]), "+v"(b[S0 + 3]), "+v"(b[S0 + 4]), "+v"(b[S0 + 5]),
          "+v"(b[S0 + 6]), "+v"(b[S0 + 7])
        : "v"(a[S0]), "v"(a[S0 + 1]), "v"(a[S0 + 2]), "v"(a[S0 + 3]), "v"(a[S0 + 4]), "v"(a[S0 + 5]), "v"(a[S0 + 6]),
          "v"(a[S0 + 7]), "v"(m), "n"(T)
        : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
}
template <int T, int S>
static __device__ __forceinline__ void fmac1_reg(float (&b)[CH_NB], const float (&a)[CH_NB], float m) {
    asm volatile("v_readlane_b32 s20, %1, %3\n\ts_nop 3\n\tv_fmac_f32 %0, s20, %2"
                 : "+v"(b[S]) : "v"(a[S]), "v"(m), "n"(T) : "s20");
}
// b[s] += m * (a[s] of lane T) for s = S0 .. T - 1
template <int T, int S0>
static __device__ __forceinline__ void fmac_regs_below(float (&b)[CH_NB], const float (&a)[CH_NB], float m) {
    if constexpr (S0 < T) {
        if constexpr (S0 + 8 <= T) {
            fmac8_regs<T, S0>(b, a, m);
            fmac_regs_below<T, S0 + 8>(b, a, m);
        } else {
            fmac1_reg<T, S0>(b, a, m);
            fmac_regs_below<T, S0 + 1>(b, a, m);
        }
    }
}
// row `lane` of inv(L): x L = e_lane, from the last column back; L[t][s] is a[s] of lane t
template <int T>
struct InverseStep {
    static __device__ __forceinline__ void run(float (&b)[CH_NB], const float (&a)[CH_NB]) {
        const float x = b[T] / lane_value(a[T], T);
        b[T] = x;
        fmac_regs_below<T, 0>(b, a, -x);
        if constexpr (T > 0) InverseStep<T - 1>::run(b, a);
    }
};

__global__ __launch_bounds__(64) void trinv_diag_kernel(const float* __restrict__ L, float* __restrict__ X,
                                                        int64_t n, int64_t ldl, int64_t ldx) {
    // full blocks only (host: the ragged last block goes to trinv_ragged_kernel)
    const int lane = threadIdx.x;
    const int64_t k0 = (int64_t)blockIdx.x * CH_NB;
    float a[CH_NB], b[CH_NB];
    const float* p = L + (k0 + lane) * ldl + k0;
#pragma unroll
    for (int c = 0; c < CH_NB; ++c) {
        a[c] = c <= lane ? p[c] : 0.0f;
        b[c] = c == lane ? 1.0f : 0.0f;
    }
    InverseStep<CH_NB - 1>::run(b, a);
    float* q = X + (k0 + lane) * ldx + k0;
#pragma unroll
    for (int c = 0; c < CH_NB; ++c) q[c] = c <= lane ? b[c] : 0.0f;
}

__global__ __launch_bounds__(64) void trinv_ragged_kernel(const float* __restrict__ L, float* __restrict__ X,
                                                          int64_t n, int64_t ldl, int64_t ldx, int64_t k0) {
    // the last, partial diagonal block: row i of inv(L) by thread i, L through LDS
    __shared__ float Ls[CH_NB][CH_LD];
    const int i = threadIdx.x, m = (int)(n - k0);
    for (int c = 0; c < CH_NB; ++c) Ls[i][c] = (i < m && c < m && c <= i) ? L[(k0 + i) * ldl + k0 + c] : (i == c ? 1.0f : 0.0f);
    __syncthreads();
    float b[CH_NB];
    for (int c = 0; c < CH_NB; ++c) b[c] = c == i ? 1.0f : 0.0f;
    for (int t = CH_NB - 1; t >= 0; --t) {
        const float x = b[t] / Ls[t][t];
        b[t] = x;
        for (int s_ = 0; s_ < t; ++s_) b[s_] -= x * Ls[t][s_];
    }
    if (i < m)
        for (int c = 0; c < m; ++c) X[(k0 + i) * ldx + k0 + c] = c <= i ? b[c] : 0.0f;
}

// One 64 x 64 tile of C = alpha * op(A) B over k in [k_lo, k_hi): A [i][k] (TA: stored [k][i]),
// B stored [k][j].  Rows / columns / k past the given extents read as zeros.
template <bool TA>
static __device__ __forceinline__ void tile_gemm(const float* __restrict__ Ap, int64_t lda, int64_t a_rows,
                                                 const float* __restrict__ Bp, int64_t ldb, int64_t b_cols,
                                                 int64_t k_lo, int64_t k_hi, f32x16& acc, float* As, float* Bs) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wy = wave >> 1, wx = wave & 1, r32 = lane & 31, kh = lane >> 5;
    for (int64_t kc = k_lo; kc < k_hi; kc += CH_NB) {
        __syncthreads();                                   // the previous chunk's operands are consumed
        for (int e = tid; e < CH_NB * CH_NB; e += 256) {
            if constexpr (TA) {
                const int k = e >> 6, i = e & 63;          // global [k][i]: consecutive threads along i
                As[i * CH_PK + k] = (kc + k < k_hi && i < a_rows) ? Ap[(kc + k) * lda + i] : 0.0f;
            } else {
                const int i = e >> 6, k = e & 63;          // global [i][k]: consecutive threads along k
                As[i * CH_PK + k] = (kc + k < k_hi && i < a_rows) ? Ap[i * lda + kc + k] : 0.0f;
            }
            const int k = e >> 6, j = e & 63;              // global [k][j]: consecutive threads along j
            Bs[j * CH_PK + k] = (kc + k < k_hi && j < b_cols) ? Bp[(kc + k) * ldb + j] : 0.0f;
        }
        __syncthreads();
        const float* A2 = &As[(32 * wy + r32) * CH_PK + 2 * kh];
        const float* B2 = &Bs[(32 * wx + r32) * CH_PK + 2 * kh];
#pragma unroll
        for (int q = 0; q < CH_NB / 4; ++q) {
            const f32x2 av = *(const f32x2*)(A2 + 4 * q);
            const f32x2 bv = *(const f32x2*)(B2 + 4 * q);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[1], acc, 0, 0, 0);
        }
    }
}

// level of the doubling: pair z owns the diagonal blocks A = [2 z s, 2 z s + s) and B = [2 z s + s,
// min(2 z s + 2 s, n)).  MODE 0: T = L[B, A] X[A, A] (X[A, A] lower: k from the tile's column on);
// MODE 1: X[B, A] = -X[B, B] T (X[B, B] lower: k up to the tile's row).  T: [pairs][s][s] floats.
template <int MODE>
__global__ __launch_bounds__(256) void trinv_level_kernel(const float* __restrict__ L, int64_t ldl,
                                                          float* __restrict__ X, int64_t ldx,
                                                          float* __restrict__ T, int64_t n, int64_t s) {
    __shared__ __attribute__((aligned(16))) float As[CH_NB * CH_PK];
    __shared__ __attribute__((aligned(16))) float Bs[CH_NB * CH_PK];
    const int64_t a0 = 2 * (int64_t)blockIdx.z * s, b0 = a0 + s;
    const int64_t mb = (b0 + s <= n ? s : n - b0);          // rows of B (<= 0: this pair has no second block)
    const int64_t ti = (int64_t)blockIdx.y * CH_NB, tj = (int64_t)blockIdx.x * CH_NB;   // tile origin in (B rows, A cols)
    if (mb <= 0 || ti >= mb) return;
    float* Tp = T + (int64_t)blockIdx.z * s * s;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int64_t rows = mb - ti < CH_NB ? mb - ti : CH_NB;
    if constexpr (MODE == 0) {
        // T[ti.., tj..] = sum_{k >= tj} L[b0 + ti.., a0 + k] X[a0 + k, a0 + tj..]
        tile_gemm<false>(L + (b0 + ti) * ldl + a0, ldl, rows, X + a0 * ldx + a0 + tj, ldx, CH_NB, tj, s, acc, As, Bs);
    } else {
        // X[b0 + ti.., a0 + tj..] = -sum_{k < ti + 64} X[b0 + ti.., b0 + k] T[k, tj..]
        const int64_t kh_ = ti + CH_NB < mb ? ti + CH_NB : mb;
        tile_gemm<false>(X + (b0 + ti) * ldx + b0, ldx, rows, Tp + tj, s, CH_NB, 0, kh_, acc, As, Bs);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wy = wave >> 1, wx = wave & 1, r32 = lane & 31, kh = lane >> 5;
    const int64_t col = tj + 32 * wx + r32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = ti + 32 * wy + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row < mb) {
            if constexpr (MODE == 0) Tp[row * s + col] = acc[r];
            else X[(b0 + row) * ldx + a0 + col] = -acc[r];
        }
    }
}

// out = X^T X, X lower triangular: out[i][j] = sum_{t >= max(i, j)} X[t][i] X[t][j]; tiles I >= J, mirrored
__global__ __launch_bounds__(256) void lauum_kernel(const float* __restrict__ X, int64_t ldx, float* __restrict__ out,
                                                    int64_t ldo, int64_t n) {
    const int I = blockIdx.y, J = blockIdx.x;
    if (J > I) return;
    __shared__ __attribute__((aligned(16))) float As[CH_NB * CH_PK];
    __shared__ __attribute__((aligned(16))) float Bs[CH_NB * CH_PK];
    const int64_t i0 = (int64_t)I * CH_NB, j0 = (int64_t)J * CH_NB;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int64_t rows = n - i0 < CH_NB ? n - i0 : CH_NB, cols = n - j0 < CH_NB ? n - j0 : CH_NB;
    // A^T[i][t] = X[t][i0 + i] (stored [t][i]); B[t][j] = X[t][j0 + j]; t from i0 (X[t][i] = 0 for t < i)
    tile_gemm<true>(X + i0 * ldx + i0, ldx, rows, X + i0 * ldx + j0, ldx, cols, 0, n - i0, acc, As, Bs);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wy = wave >> 1, wx = wave & 1, r32 = lane & 31, kh = lane >> 5;
    const int64_t col = j0 + 32 * wx + r32;
    if (col >= n) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = i0 + 32 * wy + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row < n) {
            if (I != J || col <= row) out[row * ldo + col] = acc[r];
            if (I != J || col < row) out[col * ldo + row] = acc[r];
        }
    }
}

extern "C" size_t ecoflap_cholesky_inverse_workspace_bytes(int64_t n) {
    if (n <= 0) return 0;
    const size_t nn = (size_t)n * (size_t)n;
    return (nn + nn / 2 + 4096) * sizeof(float);       // X, and the level products T
}

extern "C" int ecoflap_cholesky_inverse_f32(const float* l, int64_t n, int64_t ldl, float* out, int64_t ldo,
                                            void* workspace, size_t workspace_bytes, void* stream) {
    if (n < 0 || ldl < n || ldo < n || n > (1 << 20)) return ECOFLAP_ESIZE;
    if (n == 0) return 0;
    if (!l || !out || !workspace) return ECOFLAP_ENULL;
    if (workspace_bytes < ecoflap_cholesky_inverse_workspace_bytes(n)) return ECOFLAP_ESIZE;
    hipStream_t s = (hipStream_t)stream;
    float* X = (float*)workspace;
    float* T = X + (size_t)n * (size_t)n;
    const int64_t full = n / CH_NB, nb = (n + CH_NB - 1) / CH_NB;
    if (hipMemsetAsync(X, 0, (size_t)n * (size_t)n * sizeof(float), s) != hipSuccess) return ECOFLAP_ENULL;
    if (full > 0) hipLaunchKernelGGL(trinv_diag_kernel, dim3((unsigned)full), dim3(64), 0, s, l, X, n, ldl, n);
    if (nb > full) hipLaunchKernelGGL(trinv_ragged_kernel, dim3(1), dim3(64), 0, s, l, X, n, ldl, n, full * CH_NB);
    for (int64_t sz = CH_NB; sz < n; sz *= 2) {
        const unsigned pairs = (unsigned)((n + 2 * sz - 1) / (2 * sz)), t = (unsigned)(sz / CH_NB);
        hipLaunchKernelGGL(trinv_level_kernel<0>, dim3(t, t, pairs), dim3(256), 0, s, l, ldl, X, n, T, n, sz);
        hipLaunchKernelGGL(trinv_level_kernel<1>, dim3(t, t, pairs), dim3(256), 0, s, l, ldl, X, n, T, n, sz);
    }
    hipLaunchKernelGGL(lauum_kernel, dim3((unsigned)nb, (unsigned)nb), dim3(256), 0, s, X, n, out, ldo, n);
    ECO_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t ecoflap_cholesky_workspace_bytes(void) { return CH_NB * CH_NB * sizeof(float); }

extern "C" int ecoflap_cholesky_f32(float* a, int64_t n, int64_t lda, int upper, int* info, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    if (n < 0 || lda < n || n > (1 << 20)) return ECOFLAP_ESIZE;
    if (!info || !workspace) return ECOFLAP_ENULL;
    if (workspace_bytes < ecoflap_cholesky_workspace_bytes()) return ECOFLAP_ESIZE;
    float* Dk = (float*)workspace;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(info, 0, sizeof(int), s) != hipSuccess) return ECOFLAP_ENULL;
    if (n == 0) return 0;
    if (!a) return ECOFLAP_ENULL;
    for (int64_t k0 = 0; k0 < n; k0 += CH_NB) {
        const unsigned chunks = (unsigned)((n - k0 + CH_NB - 1) / CH_NB);
        if (k0 + CH_NB <= n)
            hipLaunchKernelGGL(chol_panel_kernel, dim3(chunks), dim3(64), 0, s, a, n, lda, k0, info, Dk);
        else
            hipLaunchKernelGGL(chol_ragged_kernel, dim3(1), dim3(256), 0, s, a, n, lda, k0, info);
        if (chunks > 1)
            hipLaunchKernelGGL(chol_trailing_kernel, dim3(chunks - 1, chunks - 1), dim3(256), 0, s, a, n, lda, k0, Dk);
    }
    const unsigned tiles = (unsigned)((n + CH_NB - 1) / CH_NB);
    hipLaunchKernelGGL(chol_finalize_kernel, dim3(tiles, tiles), dim3(256), 0, s, a, n, lda, upper ? 1 : 0);
    ECO_CHECK_LAUNCH();
    return 0;
}
