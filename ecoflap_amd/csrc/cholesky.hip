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
//   panel launch     one workgroup per 64 rows from the diagonal block down.  EVERY workgroup
//                    factors the 64 x 64 diagonal block for itself in LDS (64 dependent column
//                    steps, ~6 us; nobody waits for anybody: no flag, no grid barrier), workgroup
//                    0 writes it back, the others solve their 64 rows against it (X L^T = B,
//                    one row per thread, L broadcast from LDS);
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

// ---- the diagonal block, factored in LDS by all 256 threads -----------------------------------
// D: the block (lower triangle valid; rows / columns past n are identity).  L: the factor, lower
// triangle incl. the diagonal.  -> 0, or the 1-based local index of the first non-positive pivot
static __device__ __forceinline__ int factor_diag(float (*D)[CH_LD], float (*L)[CH_LD]) {
    const int tid = threadIdx.x;
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
#pragma unroll 4
        for (int e = tid; e < CH_NB * CH_NB; e += 256) {
            const int i = e >> 6, c = e & 63;
            if (i > j && c > j && c <= i) D[i][c] -= (D[i][j] * rd) * D[c][j];
        }
    }
    __syncthreads();
    return bad;
}

__global__ __launch_bounds__(256) void chol_panel_kernel(float* __restrict__ A, int64_t n, int64_t lda,
                                                         int64_t k0, int* __restrict__ info) {
    __shared__ float D[CH_NB][CH_LD];
    __shared__ float L[CH_NB][CH_LD];
    __shared__ float B[CH_NB][CH_LD];
    const int tid = threadIdx.x;
    for (int e = tid; e < CH_NB * CH_NB; e += 256) {
        const int r = e >> 6, c = e & 63;
        const int64_t gr = k0 + r, gc = k0 + c;
        D[r][c] = (gr < n && gc < n) ? (c <= r ? A[gr * lda + gc] : 0.0f) : (r == c ? 1.0f : 0.0f);
    }
    const int bad = factor_diag(D, L);
    if (blockIdx.x == 0) {
        if (bad && tid == 0 && info[0] == 0) info[0] = (int)(k0 + bad);
        for (int e = tid; e < CH_NB * CH_NB; e += 256) {
            const int r = e >> 6, c = e & 63;
            const int64_t gr = k0 + r, gc = k0 + c;
            if (gr < n && gc < n && c <= r) A[gr * lda + gc] = L[r][c];
        }
        return;
    }
    // rows r0 .. r0 + 63 of the panel: X L^T = B, forward substitution along the 64 columns
    const int64_t r0 = k0 + (int64_t)blockIdx.x * CH_NB;
    for (int e = tid; e < CH_NB * CH_NB; e += 256) {
        const int r = e >> 6, c = e & 63;
        B[r][c] = (r0 + r < n && k0 + c < n) ? A[(r0 + r) * lda + k0 + c] : 0.0f;
    }
    __syncthreads();
    if (tid < CH_NB) {
        float x[CH_NB];
#pragma unroll
        for (int j = 0; j < CH_NB; ++j) {
            float acc = B[tid][j];
#pragma unroll
            for (int t = 0; t < j; ++t) acc -= x[t] * L[j][t];
            x[j] = acc / L[j][j];
        }
#pragma unroll
        for (int j = 0; j < CH_NB; ++j) B[tid][j] = x[j];
    }
    __syncthreads();
    for (int e = tid; e < CH_NB * CH_NB; e += 256) {
        const int r = e >> 6, c = e & 63;
        if (r0 + r < n && k0 + c < n) A[(r0 + r) * lda + k0 + c] = B[r][c];
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
        hipLaunchKernelGGL(chol_panel_kernel, dim3(chunks), dim3(256), 0, s, a, n, lda, k0, info);
        if (chunks > 1)
            hipLaunchKernelGGL(chol_trailing_kernel, dim3(chunks - 1, chunks - 1), dim3(256), 0, s, a, n, lda, k0);
    }
    const unsigned tiles = (unsigned)((n + CH_NB - 1) / CH_NB);
    hipLaunchKernelGGL(chol_finalize_kernel, dim3(tiles, tiles), dim3(256), 0, s, a, n, lda, upper ? 1 : 0);
    ECO_CHECK_LAUNCH();
    return 0;
}
