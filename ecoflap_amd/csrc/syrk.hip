// syrk.hip — SparseGPT Hessian accumulation on the matrix cores (gfx950 MFMA).
//
// Replaces SparseGPT.add_batch's update
//   LAVIS/lavis/compression/pruners/sparsegpt_pruner.py:71-82
//     self.H *= n / (n + b);  n += b
//     inp = sqrt(2 / n) * inp.float();  self.H += inp.matmul(inp.t())       (inp = x^T)
// i.e.  H <- beta * H + alpha * X^T X   with beta = n/(n+b), alpha = 2/(n+b), X = [tokens, cols]
// the Linear's input as the forward produced it (fp16 under the ViT's autocast, bf16 for T5).
//
// The one GEMM-shaped contraction of the path: fp16 / bf16 operands
// straight into v_mfma_f32_32x32x16_{f16,bf16} (products exact in fp32, fp32 accumulate), upper
// triangle of 128x128 tiles only (H is symmetric), each off-diagonal tile stored a second time
// transposed through LDS so both triangles stay filled for the factorisations that follow.
//   1. syrk_transpose_kernel: X[T, C] -> Xt[C, Kpad] (k contiguous, zero padded to a multiple of
//      64), so that BOTH operand fragments of Xt * Xt^T are 16-byte row reads.
//   2. syrk_kernel: one workgroup (4 waves, 2x2, 64x64 each = 2x2 MFMA tiles) per tile pair,
//      K in chunks of 64 staged through LDS (rows padded to 144 B: conflict-free ds_read_b128).
// MFMA roofline: 2 * T * C * (C + 128) / 2 flops per call against the dense fp16/bf16 peak.
#include "common.h"

#define SY_TILE 128
#define SY_KC 64
#define SY_ROWB (SY_KC * 2 + 16)      // LDS bytes per staged row (128 B of k + 16 B pad)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// X[T, C] (16-bit) -> Xt[C, Kpad], 64x64 tiles through LDS; columns T..Kpad-1 are zero
__global__ __launch_bounds__(256) void syrk_transpose_kernel(const uint16_t* __restrict__ x,
                                                             int64_t T, int64_t C, int64_t Kpad,
                                                             uint16_t* __restrict__ xt) {
    __shared__ uint16_t tile[64][66];
    const int64_t c0 = (int64_t)blockIdx.x * 64, t0 = (int64_t)blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;       // 64 x 4
    for (int r = ty; r < 64; r += 4) {
        const int64_t t = t0 + r, c = c0 + tx;
        tile[r][tx] = (t < T && c < C) ? x[t * C + c] : (uint16_t)0;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int64_t c = c0 + r, t = t0 + tx;
        if (c < C && t < Kpad) xt[c * Kpad + t] = tile[tx][r];
    }
}

template <int DT> struct Mfma;
template <> struct Mfma<ECOFLAP_F16> {
    static __device__ __forceinline__ f32x16 run(const u32x4& a, const u32x4& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a),
                                                      __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mfma<ECOFLAP_BF16> {
    static __device__ __forceinline__ f32x16 run(const u32x4& a, const u32x4& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                       __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};

// WT = MFMA tiles per wave and dimension: workgroup tile = 64*WT (128 for large C; 64 when the
// 128-tile grid would leave most of the 256 CUs idle, e.g. C = 1408 -> 66 tiles)
template <int DT, int WT>
__global__ __launch_bounds__(256) void syrk_kernel(const uint16_t* __restrict__ xt, int64_t C,
                                                   int64_t Kpad, float* __restrict__ H, float beta,
                                                   float alpha, int ntiles) {
    constexpr int TILE = 64 * WT;
    constexpr int LD_PER_THREAD = TILE * 8 / 256;       // 16-byte loads per thread and operand
    __shared__ __attribute__((aligned(16))) char lds[2 * SY_TILE * SY_ROWB];
    char* ldsA = lds;
    char* ldsB = lds + TILE * SY_ROWB;
    // workgroups are dispatched round-robin over the 8 XCDs: give every XCD a contiguous range of
    // the (column-major) upper-triangle order, so the B panel of a column and the A panels next
    // to each other are shared in ITS L2
    const int per = (ntiles + 7) / 8;
    const int t = (int)(blockIdx.x % 8) * per + (int)(blockIdx.x / 8);
    if (t >= ntiles) return;
    int bj = (int)((__builtin_sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while ((bj + 1) * (bj + 2) / 2 <= t) ++bj;
    while (bj * (bj + 1) / 2 > t) --bj;
    const int bi = t - bj * (bj + 1) / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int64_t rowA0 = (int64_t)bi * TILE, rowB0 = (int64_t)bj * TILE;

    f32x16 acc[WT][WT];
#pragma unroll
    for (int mi = 0; mi < WT; ++mi)
#pragma unroll
        for (int ni = 0; ni < WT; ++ni)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[mi][ni][q] = 0.f;

    // register double buffering: the next chunk's global loads are in flight during the MFMAs
    u32x4 ra[LD_PER_THREAD], rb[LD_PER_THREAD];
    auto fetch = [&](int64_t k0) {
#pragma unroll
        for (int i = 0; i < LD_PER_THREAD; ++i) {
            const int idx = tid + 256 * i, row = idx >> 3, seg = idx & 7;
            ra[i] = u32x4{0u, 0u, 0u, 0u};
            rb[i] = u32x4{0u, 0u, 0u, 0u};
            if (rowA0 + row < C) ra[i] = *(const u32x4*)(xt + (rowA0 + row) * Kpad + k0 + seg * 8);
            if (rowB0 + row < C) rb[i] = *(const u32x4*)(xt + (rowB0 + row) * Kpad + k0 + seg * 8);
        }
    };
    fetch(0);
    for (int64_t k0 = 0; k0 < Kpad; k0 += SY_KC) {
#pragma unroll
        for (int i = 0; i < LD_PER_THREAD; ++i) {
            const int idx = tid + 256 * i, row = idx >> 3, seg = idx & 7;
            *(u32x4*)(ldsA + row * SY_ROWB + seg * 16) = ra[i];
            *(u32x4*)(ldsB + row * SY_ROWB + seg * 16) = rb[i];
        }
        __syncthreads();
        if (k0 + SY_KC < Kpad) fetch(k0 + SY_KC);
#pragma unroll
        for (int kk = 0; kk < SY_KC / 16; ++kk) {
            u32x4 a[WT], b[WT];
#pragma unroll
            for (int mi = 0; mi < WT; ++mi)      // A[row r][k = 8h + j]
                a[mi] = *(const u32x4*)(ldsA + (wm * 32 * WT + mi * 32 + r) * SY_ROWB + (kk * 16 + 8 * h) * 2);
#pragma unroll
            for (int ni = 0; ni < WT; ++ni)      // B[k = 8h + j][col r] = Xt[col][k]
                b[ni] = *(const u32x4*)(ldsB + (wn * 32 * WT + ni * 32 + r) * SY_ROWB + (kk * 16 + 8 * h) * 2);
#pragma unroll
            for (int mi = 0; mi < WT; ++mi)
#pragma unroll
                for (int ni = 0; ni < WT; ++ni) acc[mi][ni] = Mfma<DT>::run(a[mi], b[ni], acc[mi][ni]);
        }
        __syncthreads();
    }

    // ---- epilogue: H = beta * H + alpha * acc on the tile (C/D map: col = lane & 31,
    //      row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)); the value stays in acc for the mirror
#pragma unroll
    for (int mi = 0; mi < WT; ++mi)
#pragma unroll
        for (int ni = 0; ni < WT; ++ni) {
            const int64_t gc = rowB0 + wn * 32 * WT + ni * 32 + r;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int64_t gr = rowA0 + wm * 32 * WT + mi * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                float v = alpha * acc[mi][ni][q];
                if (gr < C && gc < C) {
                    if (beta != 0.f) v += beta * H[gr * C + gc];
                    H[gr * C + gc] = v;
                }
                acc[mi][ni][q] = v;
            }
        }
    if (bi == bj) return;          // diagonal tiles were computed in full
    // ---- mirror: H[col][row] = H[row][col], transposed through LDS so the stores run along rows
    constexpr int TR = 32 * WT;                        // rows of the wave's sub-tile
    float* ldsT = (float*)lds + wave * (32 * (TR + 1));   // per wave [32 cols][TR rows + 1]
#pragma unroll
    for (int ni = 0; ni < WT; ++ni) {
        __syncthreads();                               // staging buffers / previous half are free
#pragma unroll
        for (int mi = 0; mi < WT; ++mi)
#pragma unroll
            for (int q = 0; q < 16; ++q)
                ldsT[r * (TR + 1) + mi * 32 + (q & 3) + 8 * (q >> 2) + 4 * h] = acc[mi][ni][q];
        __syncthreads();
        if (lane < TR) {
            const int64_t gr = rowA0 + wm * TR + lane;     // original row -> mirrored column
            for (int c = 0; c < 32; ++c) {
                const int64_t gc = rowB0 + wn * TR + ni * 32 + c;   // original column -> mirrored row
                if (gc < C && gr < C) H[gc * C + gr] = ldsT[c * (TR + 1) + lane];
            }
        }
    }
}

static inline int64_t syrk_kpad(int64_t tokens) { return (tokens + SY_KC - 1) / SY_KC * SY_KC; }

extern "C" size_t ecoflap_hessian_workspace_bytes(int64_t tokens, int64_t cols) {
    if (tokens <= 0 || cols <= 0) return 0;
    return (size_t)cols * (size_t)syrk_kpad(tokens) * 2;      // Xt, 16-bit
}

extern "C" int ecoflap_hessian_accum(float* H, const void* x, int64_t tokens, int64_t cols,
                                     int dtype, int64_t nsamples_before, int64_t batch,
                                     void* workspace, size_t workspace_bytes, void* stream) {
    if (dtype != ECOFLAP_F16 && dtype != ECOFLAP_BF16) return ECOFLAP_EDTYPE;   // fp32 X: library GEMM
    if (tokens <= 0 || cols <= 0 || nsamples_before < 0 || batch <= 0) return ECOFLAP_ESIZE;
    // 128-wide tiles when they give every CU work, 64-wide otherwise
    const int64_t nt128 = (cols + 127) / 128;
    const int wt = nt128 * (nt128 + 1) / 2 >= 512 ? 2 : 1;
    const int64_t nt = (cols + 64 * wt - 1) / (64 * wt);
    if (nt * (nt + 1) / 2 > 0x7ffffff0LL) return ECOFLAP_ESIZE;
    if (!H || !x || !workspace) return ECOFLAP_ENULL;
    if (!aligned16(workspace)) return ECOFLAP_EALIGN;
    if (workspace_bytes < ecoflap_hessian_workspace_bytes(tokens, cols)) return ECOFLAP_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int64_t kpad = syrk_kpad(tokens);
    uint16_t* xt = (uint16_t*)workspace;
    hipLaunchKernelGGL(syrk_transpose_kernel, dim3((unsigned)((cols + 63) / 64), (unsigned)(kpad / 64)),
                       dim3(256), 0, s, (const uint16_t*)x, tokens, cols, kpad, xt);
    ECO_CHECK_LAUNCH();
    // (:79-81) H *= n/(n+b); n += b; inp = sqrt(2/n) x  ->  alpha = 2/n_new on x^T x
    const float beta = (float)((double)nsamples_before / (double)(nsamples_before + batch));
    const float alpha = (float)(2.0 / (double)(nsamples_before + batch));
    const int ntiles = (int)(nt * (nt + 1) / 2);
    const dim3 grid((unsigned)((ntiles + 7) / 8 * 8));
#define SYRK_GO(DT_, WT_) hipLaunchKernelGGL((syrk_kernel<DT_, WT_>), grid, dim3(256), 0, s, xt, cols, kpad, H, beta, alpha, ntiles)
    if (dtype == ECOFLAP_F16) { if (wt == 2) SYRK_GO(ECOFLAP_F16, 2); else SYRK_GO(ECOFLAP_F16, 1); }
    else { if (wt == 2) SYRK_GO(ECOFLAP_BF16, 2); else SYRK_GO(ECOFLAP_BF16, 1); }
#undef SYRK_GO
    ECO_CHECK_LAUNCH();
    return 0;
}
