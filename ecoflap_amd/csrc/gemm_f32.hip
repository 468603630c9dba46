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
// Numerics: every output element is ONE fp32 accumulation chain over k in a FIXED order — K in
// ascending quads, inside quad q the MFMA steps take the pairs (4q, 4q + 2) then (4q + 1, 4q + 3),
// the two products of a step summed by the MFMA itself, one rounding per accumulate, the bias
// added last.  The chain of a row does not depend on which rows travel with it, nor on M: batch
// invariant and repeatable by construction.
//
// Tiling: 64 x 64 output tile per 256-thread workgroup, one 32 x 32 MFMA tile per wave (16
// accumulator registers); K in chunks of 32 through two LDS images per operand — one barrier per
// chunk: the next chunk is written, from the registers its global loads filled a chunk earlier,
// while this one multiplies; four workgroups per CU, so a SIMD's MFMA pipe has four waves to draw
// from.  What the variants measured on the Q-Former's shapes at 16 evaluations
// (profiles/r04_secondary/gemm_f32_variants.md): 128 x 128 and 128 x 64 tiles lose to 64 x 64
// everywhere (4096 x 768 x 768 is 192 tiles of 128 on 256 CUs; the big tiles hold one or two
// workgroups per CU); chunks of 64 lose 8 % (two workgroups per CU); a hand-fenced instruction order
// loses to the compiler's; v_mfma_f32_16x16x4_f32 and a second register stage change nothing.
// 113-124 TFLOP/s = 72-79 % of the 157 TFLOP/s fp32 MFMA peak on the four larger shapes (three
// boxes), 89-112 on 4096 x 768 x 768; the MFMA pipes busy 79 % of the cycles at 2.41 GHz (the
// library's Stream-K kernel on the same problem: 84 %, same clock).
#include <stdlib.h>

#include "common.h"
#include "../../include/ecoflap_shape_ops.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define GF_BK 32                // k per chunk
#define GF_BT 64                // output tile edge
// Row pitch of the LDS images in floats: the 8-byte operand reads of a half-wave (32 rows, one k
// pair) fall on 32 distinct even banks (34 r mod 64 = 2 (17 r mod 32)); 8-byte aligned stores.
#define GF_LDK (GF_BK + 2)

__global__ __launch_bounds__(256) void gemm_f32_nt_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ w,
                                                          const float* __restrict__ bias,
                                                          float* __restrict__ y, int64_t M, int64_t N,
                                                          int64_t K, int tiles_n) {
    __shared__ __attribute__((aligned(16))) float As[2][GF_BT * GF_LDK];
    __shared__ __attribute__((aligned(16))) float Bs[2][GF_BT * GF_LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // consecutive workgroups walk the column tiles of one row tile: its x rows are fetched once
    // and W (a few MB) stays in L2 / the Infinity Cache whole
    const int tn = blockIdx.x % tiles_n, tm = blockIdx.x / tiles_n;
    const int64_t m0 = (int64_t)tm * GF_BT, n0 = (int64_t)tn * GF_BT;
    // global -> registers: thread t takes 16-byte vector (t % 8) of rows t / 8 and t / 8 + 32.
    // Rows past M read row M - 1 (their products are never stored): no branch in the loop.
    const int lrow = tid >> 3, lvec = tid & 7;
    const float* xr[2];
    const float* wr[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int64_t ra = m0 + lrow + 32 * j;
        xr[j] = x + (ra < M ? ra : M - 1) * K + 4 * lvec;
        wr[j] = w + (n0 + lrow + 32 * j) * K + 4 * lvec;        // N % 64 == 0
    }
    f32x4 pa[2], pb[2];
    auto load_chunk = [&](int64_t k0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            pa[j] = *(const f32x4*)(xr[j] + k0);
            pb[j] = *(const f32x4*)(wr[j] + k0);
        }
    };
    const int st_off = lrow * GF_LDK + 4 * lvec;
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float* da = &As[buf][st_off + 32 * j * GF_LDK];
            float* db = &Bs[buf][st_off + 32 * j * GF_LDK];
            *(f32x2*)da = f32x2{pa[j][0], pa[j][1]};
            *(f32x2*)(da + 2) = f32x2{pa[j][2], pa[j][3]};
            *(f32x2*)db = f32x2{pb[j][0], pb[j][1]};
            *(f32x2*)(db + 2) = f32x2{pb[j][2], pb[j][3]};
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // wave (wy, wx) owns rows [32 wy, +32) x columns [32 wx, +32) of the tile.  MFMA operand lane
    // l = row (or column) l & 31, k slot l >> 5: it reads the k pair 4 q + 2 (l >> 5) + {0, 1} —
    // its operands of two MFMA steps — in one 8-byte read.
    const int wy = wave >> 1, wx = wave & 1;
    const int r32 = lane & 31, kh = lane >> 5;
    const int a_off = (32 * wy + r32) * GF_LDK + 2 * kh;
    const int b_off = (32 * wx + r32) * GF_LDK + 2 * kh;
    auto products = [&](int buf) {
        const float* A = &As[buf][a_off];
        const float* B = &Bs[buf][b_off];
#pragma unroll
        for (int q = 0; q < GF_BK / 4; ++q) {
            const f32x2 av = *(const f32x2*)(A + 4 * q);
            const f32x2 bv = *(const f32x2*)(B + 4 * q);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[1], acc, 0, 0, 0);
        }
    };
    const int64_t last = K - GF_BK;        // the load past the end re-reads the last chunk (never used)
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    load_chunk(GF_BK < K ? GF_BK : last);
    int64_t k0 = 0;
    int buf = 0;
    for (; k0 + GF_BK < K; k0 += GF_BK, buf ^= 1) {
        // chunk k0 out of image `buf`; meanwhile the next chunk goes registers -> the other image
        // (nobody reads that one before the barrier) and the chunk after it starts on its way
        products(buf);
        store_chunk(buf ^ 1);
        load_chunk(k0 + 2 * GF_BK < K ? k0 + 2 * GF_BK : last);
        __syncthreads();        // image `buf` is free to be rewritten, the other one is complete
    }
    products(buf);
    // C/D map of the 32 x 32 shapes: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const int64_t col = n0 + 32 * wx + r32;
    const float bias_v = bias ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + 32 * wy + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row < M) y[row * N + col] = bias ? acc[r] + bias_v : acc[r];
    }
}

extern "C" int ecoflap_linear_f32(const float* x, const float* w, const float* bias, float* y,
                                  int64_t M, int64_t N, int64_t K, void* stream) {
    if (M < 0 || N <= 0 || K <= 0 || N % 128 != 0 || K % GF_BK != 0) return ECOFLAP_ESIZE;
    if (M == 0) return 0;
    if (!x || !w || !y) return ECOFLAP_ENULL;
    if (!aligned16(x) || !aligned16(w)) return ECOFLAP_EALIGN;
    const int64_t tiles_n = N / GF_BT, tiles = ((M + GF_BT - 1) / GF_BT) * tiles_n;
    if (tiles > 0x7fffffffLL) return ECOFLAP_ESIZE;
    hipLaunchKernelGGL(gemm_f32_nt_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, x, w, bias, y,
                       M, N, K, (int)tiles_n);
    ECO_CHECK_LAUNCH();
    return 0;
}
