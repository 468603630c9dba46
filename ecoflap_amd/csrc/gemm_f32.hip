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
// (32 x 32 each, 64 accumulator registers) — 128 x 64 or 64 x 64 when that fills the CUs better —,
// K in chunks of 32 through two LDS images per operand, the operands of the next two MFMA steps
// read while the current ones multiply, the next chunk written and the one after it loaded in the
// middle of the current chunk's products (one barrier per chunk).
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "../../include/ecoflap_shape_ops.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define GF_BK 32
#define GF_LDK (GF_BK + 2)      // row pitch of the LDS images in floats (see the bank note below)

// TM x TN MFMA tiles (32 x 32) per wave, 2 x 2 waves per workgroup: (2, 2) = a 128 x 128 output
// tile, (1, 1) = 64 x 64 for problems whose 128-wide tiling would leave CUs idle or badly
// quantised (4096 x 768: 192 tiles of 128 on 256 CUs, 768 of 64 = three each).
//
// LDS images: [row][GF_LDK] with the k's of every quad stored (k, k+2, k+1, k+3): one ds_read_b64
// then hands lane l (row l & 31) the pair (k, k+2) in the low half-wave and (k+1, k+3) in the high
// one — the operands of TWO consecutive MFMA steps, k ascending across them — at an immediate
// offset from one base register per operand.  Pitch 34: the 32 rows of a half-wave fall on 32
// distinct even banks (34 r mod 64 = 2 (17 r mod 32)), the 8-byte reads and writes conflict-free.
// Two images per operand: chunk c+1 is written (from the registers its global loads filled one
// chunk earlier) in the middle of chunk c's products, one barrier per chunk.
template <int TM, int TN, int SCHED>
__global__ __launch_bounds__(256) void gemm_f32_nt_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ w,
                                                          const float* __restrict__ bias,
                                                          float* __restrict__ y, int64_t M, int64_t N,
                                                          int64_t K, int tiles_n) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    __shared__ float As[2][BM * GF_LDK];
    __shared__ float Bs[2][BN * GF_LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // consecutive workgroups walk the column tiles of one row tile: its x rows are fetched once
    // and W (a few MB) stays in L2 / the Infinity Cache whole
    const int tn = blockIdx.x % tiles_n, tm = blockIdx.x / tiles_n;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
    // global -> registers: thread t takes 16-byte vector (t % 8) of rows t / 8 + 32 j.  Rows past
    // M read row M - 1 (their products are never stored): no branch in the loop.
    const int lrow = tid >> 3, lvec = tid & 7;
    const float* xr[2 * TM];
    const float* wr[2 * TN];
#pragma unroll
    for (int j = 0; j < 2 * TM; ++j) {
        const int64_t ra = m0 + lrow + 32 * j;
        xr[j] = x + (ra < M ? ra : M - 1) * K + 4 * lvec;
    }
#pragma unroll
    for (int j = 0; j < 2 * TN; ++j) wr[j] = w + (n0 + lrow + 32 * j) * K + 4 * lvec;      // N % BN == 0
    f32x4 pa[2 * TM], pb[2 * TN];
    auto load_chunk = [&](int64_t k0) {
#pragma unroll
        for (int j = 0; j < 2 * TM; ++j) pa[j] = *(const f32x4*)(xr[j] + k0);
#pragma unroll
        for (int j = 0; j < 2 * TN; ++j) pb[j] = *(const f32x4*)(wr[j] + k0);
    };
    const int st_off = lrow * GF_LDK + 4 * lvec;
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2 * TM; ++j) {
            float* d = &As[buf][st_off + 32 * j * GF_LDK];
            *(f32x2*)d = f32x2{pa[j][0], pa[j][1]};
            *(f32x2*)(d + 2) = f32x2{pa[j][2], pa[j][3]};
        }
#pragma unroll
        for (int j = 0; j < 2 * TN; ++j) {
            float* d = &Bs[buf][st_off + 32 * j * GF_LDK];
            *(f32x2*)d = f32x2{pb[j][0], pb[j][1]};
            *(f32x2*)(d + 2) = f32x2{pb[j][2], pb[j][3]};
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
    const int a_off = (32 * TM * wy + r32) * GF_LDK + 2 * kh;
    const int b_off = (32 * TN * wx + r32) * GF_LDK + 2 * kh;
    // One chunk's products: four groups of 8 k (two quads: 4 MFMA steps x TM x TN tiles), the next
    // group's operands read while this one multiplies.  STORE: in the second group, the next
    // chunk goes registers -> the other image (nobody reads it before the barrier that follows)
    // and the chunk after it starts on its way from global memory.
    auto chunk = [&](int buf, auto store_tag, int64_t k_after) {
        constexpr bool STORE = decltype(store_tag)::value;
        const float* A = &As[buf][a_off];
        const float* B = &Bs[buf][b_off];
        f32x2 av[TM][2], bv[TN][2], an[TM][2], bn[TN][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int a = 0; a < TM; ++a) av[a][h] = *(const f32x2*)(A + 32 * a * GF_LDK + 4 * h);
#pragma unroll
            for (int b = 0; b < TN; ++b) bv[b][h] = *(const f32x2*)(B + 32 * b * GF_LDK + 4 * h);
        }
#pragma unroll
        for (int g = 0; g < GF_BK / 8; ++g) {
            if (g + 1 < GF_BK / 8) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int a = 0; a < TM; ++a)
                        an[a][h] = *(const f32x2*)(A + 32 * a * GF_LDK + 8 * (g + 1) + 4 * h);
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        bn[b][h] = *(const f32x2*)(B + 32 * b * GF_LDK + 8 * (g + 1) + 4 * h);
                }
            }
            if (SCHED) __builtin_amdgcn_sched_barrier(0);       // the reads above are issued before the products below
            if (STORE && g == 1) store_chunk(buf ^ 1);
            if (STORE && g == 2) load_chunk(k_after);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < TN; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a][h][i], bv[b][h][i], acc[a][b], 0, 0, 0);
            // nothing crosses a group's end: the stores stay in group 1 and the loads in group 2,
            // a chunk's worth of products (~4000 cycles) between a load and the store that needs it
            if (SCHED) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int a = 0; a < TM; ++a) av[a][h] = an[a][h];
#pragma unroll
                for (int b = 0; b < TN; ++b) bv[b][h] = bn[b][h];
            }
        }
    };
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    if (GF_BK < K) load_chunk(GF_BK);
    int buf = 0;
    int64_t k0 = 0;
    for (; k0 + GF_BK < K; k0 += GF_BK, buf ^= 1) {
        // (the last store-carrying chunk re-loads the final chunk: harmless, keeps the body branch-free)
        chunk(buf, std::true_type{}, k0 + 2 * GF_BK < K ? k0 + 2 * GF_BK : K - GF_BK);
        __syncthreads();        // this image is free to be rewritten, the other one is complete
    }
    chunk(buf, std::false_type{}, 0);
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

// Debug knobs for tools/gemm_f32_launches.py (read at every launch: the tool flips them):
// ECOFLAP_GEMM_F32_TILE=22|21|11 forces a tile shape, ECOFLAP_GEMM_F32_SCHED=0 leaves the
// instruction order to the compiler.
static int gemm_f32_knob(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

template <int TM, int TN>
static void gemm_f32_launch(const float* x, const float* w, const float* bias, float* y, int64_t M,
                            int64_t N, int64_t K, hipStream_t s) {
    const int64_t tiles_n = N / (64 * TN), tiles = ((M + 64 * TM - 1) / (64 * TM)) * tiles_n;
    if (gemm_f32_knob("ECOFLAP_GEMM_F32_SCHED", 1))
        hipLaunchKernelGGL((gemm_f32_nt_kernel<TM, TN, 1>), dim3((unsigned)tiles), dim3(256), 0, s, x, w, bias, y, M, N,
                           K, (int)tiles_n);
    else
        hipLaunchKernelGGL((gemm_f32_nt_kernel<TM, TN, 0>), dim3((unsigned)tiles), dim3(256), 0, s, x, w, bias, y, M, N,
                           K, (int)tiles_n);
}

extern "C" int ecoflap_linear_f32(const float* x, const float* w, const float* bias, float* y,
                                  int64_t M, int64_t N, int64_t K, void* stream) {
    if (M < 0 || N <= 0 || K <= 0 || N % 128 != 0 || K % GF_BK != 0) return ECOFLAP_ESIZE;
    if (M == 0) return 0;
    if (!x || !w || !y) return ECOFLAP_ENULL;
    if (!aligned16(x) || !aligned16(w)) return ECOFLAP_EALIGN;
    if (((M + 63) / 64) * (N / 64) > 0x7fffffffLL) return ECOFLAP_ESIZE;
    // The tile shape changes how the work is cut, never the k order of an output element: the
    // same bits whichever is chosen.  Cost model: a CU's four SIMDs each run one wave of a
    // workgroup, TM x TN MFMAs per k step, so a shape costs (workgroups per CU, rounded up) x
    // TM x TN; the smaller tiles re-read more of x and W through L2 and are taken only when the
    // rounding says they win clearly.
    const int64_t cus = 256;
    auto cost = [&](int tm, int tn) {
        const int64_t t = ((M + 64 * tm - 1) / (64 * tm)) * (N / (64 * tn));
        return ((t + cus - 1) / cus) * tm * tn;
    };
    int pick = gemm_f32_knob("ECOFLAP_GEMM_F32_TILE", 0);
    if (pick != 22 && pick != 21 && pick != 11) {
        const int64_t c22 = cost(2, 2), c21 = cost(2, 1), c11 = cost(1, 1);
        pick = 22;
        int64_t best = c22;
        if (c21 * 8 <= best * 7) { pick = 21; best = c21; }
        if (c11 * 8 <= best * 7) { pick = 11; best = c11; }
    }
    hipStream_t s = (hipStream_t)stream;
    if (pick == 22) gemm_f32_launch<2, 2>(x, w, bias, y, M, N, K, s);
    else if (pick == 21) gemm_f32_launch<2, 1>(x, w, bias, y, M, N, K, s);
    else gemm_f32_launch<1, 1>(x, w, bias, y, M, N, K, s);
    ECO_CHECK_LAUNCH();
    return 0;
}
