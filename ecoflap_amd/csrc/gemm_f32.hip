// gemm_f32.hip — y[M,N] = x[M,K] W[N,K]^T (+ bias[N]) in fp32 on v_mfma_f32_32x32x2_f32, for the
// forward's fp32 Linears (BLIP-2's Q-Former bridge).  Plumbing of the shape modules
// (include/ecoflap_shape_ops.h), not the pruner ABI.
//
// Why a GEMM of our own here: every fp32 solution hipBLASLt ships for gfx950 is a Stream-K kernel
// (`TensileLibrary_SS_SS_HA_Bias_SAV_UA_*_gfx950.dat`: 466 of 466 names carry `_SK3`;
// csrc/gemm_pinned.hip finds 0 of 4 632 candidates name-clean), so there is nothing to pin, and
// the library's choice is not batch invariant for these shapes: the last slot of 16 concatenated
// evaluations differs from the same rows alone (the loop's padding slots exist for that), and at
// batch size 1 the bridge was the one stage left that could not be shared.
//
// Numerics: every output element is ONE k-ordered fp32 fma chain (the MFMA's own definition:
// D = fma(a_k, b_k, C) for ascending k, one rounding per product), K walked in ascending chunks,
// the bias added last.  The chain of a row does not depend on which rows travel with it:
// batch invariant and repeatable by construction.
//
// Tiling: 128 x 128 output tile per 256-thread workgroup, four waves of 2 x 2 MFMA tiles
// (32 x 32 each, 64 accumulator registers) — 64 x 64 with one tile per wave when the problem is
// small —, K in chunks of 32 through LDS stored [k][row] (operand reads: 32 consecutive floats per
// half-wave, conflict-free), the next chunk's global loads in flight in registers while the
// current one is multiplied.
#include "common.h"
#include "../../include/ecoflap_shape_ops.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GF_BK 32

// TM x TN MFMA tiles (32 x 32) per wave, 2 x 2 waves per workgroup: (2, 2) = a 128 x 128 output
// tile, (1, 1) = 64 x 64 for problems whose 128-wide tiling would leave most CUs idle (the
// Q-Former at batch size 1: 512 x 768 = 24 tiles of 128, 96 of 64).
template <int TM, int TN>
__global__ __launch_bounds__(256) void gemm_f32_nt_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ w,
                                                          const float* __restrict__ bias,
                                                          float* __restrict__ y, int64_t M, int64_t N,
                                                          int64_t K, int tiles_m) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int LDA = BM + 1, LDB = BN + 1;          // [k][row] images, one float of padding per k row
    __shared__ float As[GF_BK * LDA];
    __shared__ float Bs[GF_BK * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // consecutive workgroups walk the row tiles of one column tile: its W rows stay in L2
    const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
    // global -> registers: thread t takes 16-byte vector (t % 8) of rows t / 8 + 32 j
    const int lrow = tid >> 3, lvec = tid & 7;
    f32x4 pa[2 * TM], pb[2 * TN];
    auto load_chunk = [&](int64_t k0) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2 * TM; ++j) {
            const int64_t ra = m0 + lrow + 32 * j;
            pa[j] = ra < M ? *(const f32x4*)(x + ra * K + k0 + 4 * lvec) : zero;
        }
#pragma unroll
        for (int j = 0; j < 2 * TN; ++j)
            pb[j] = *(const f32x4*)(w + (n0 + lrow + 32 * j) * K + k0 + 4 * lvec);      // N % BN == 0
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 2 * TM; ++j) As[(4 * lvec + i) * LDA + lrow + 32 * j] = pa[j][i];
#pragma unroll
            for (int j = 0; j < 2 * TN; ++j) Bs[(4 * lvec + i) * LDB + lrow + 32 * j] = pb[j][i];
        }
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // wave (wy, wx) owns rows [32 TM wy, ...) x columns [32 TN wx, ...) of the tile
    const int wy = wave >> 1, wx = wave & 1;
    const int r32 = lane & 31, kh = lane >> 5;
    load_chunk(0);
    for (int64_t k0 = 0; k0 < K; k0 += GF_BK) {
        __syncthreads();                     // everyone is done reading the previous chunk
        store_chunk();
        __syncthreads();
        if (k0 + GF_BK < K) load_chunk(k0 + GF_BK);      // in flight during the products below
#pragma unroll
        for (int ks = 0; ks < GF_BK; ks += 2) {
            float av[TM], bv[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) av[a] = As[(ks + kh) * LDA + 32 * TM * wy + 32 * a + r32];
#pragma unroll
            for (int b = 0; b < TN; ++b) bv[b] = Bs[(ks + kh) * LDB + 32 * TN * wx + 32 * b + r32];
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
    }
    // C/D map of the 32 x 32 shapes: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int64_t col = n0 + 32 * TN * wx + 32 * b + r32;
            const float bias_v = bias ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + 32 * TM * wy + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row < M) y[row * N + col] = bias ? acc[a][b][r] + bias_v : acc[a][b][r];
            }
        }
}

extern "C" int ecoflap_linear_f32(const float* x, const float* w, const float* bias, float* y,
                                  int64_t M, int64_t N, int64_t K, void* stream) {
    if (M < 0 || N <= 0 || K <= 0 || N % 128 != 0 || K % GF_BK != 0) return ECOFLAP_ESIZE;
    if (M == 0) return 0;
    if (!x || !w || !y) return ECOFLAP_ENULL;
    if (!aligned16(x) || !aligned16(w)) return ECOFLAP_EALIGN;
    // (the tile shape changes how the work is cut, never the k order of an output element:
    // results are the same bits whichever is chosen)
    const int64_t big_m = (M + 127) / 128, big = big_m * (N / 128);
    hipStream_t s = (hipStream_t)stream;
    if (big >= 192) {
        if (big > 0x7fffffffLL) return ECOFLAP_ESIZE;
        hipLaunchKernelGGL((gemm_f32_nt_kernel<2, 2>), dim3((unsigned)big), dim3(256), 0, s, x, w, bias, y, M, N, K,
                           (int)big_m);
    } else {
        const int64_t small_m = (M + 63) / 64, small = small_m * (N / 64);
        hipLaunchKernelGGL((gemm_f32_nt_kernel<1, 1>), dim3((unsigned)small), dim3(256), 0, s, x, w, bias, y, M, N, K,
                           (int)small_m);
    }
    ECO_CHECK_LAUNCH();
    return 0;
}
