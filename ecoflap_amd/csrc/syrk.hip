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
//      (syrk_transpose8_kernel for C % 8 == 0: 16-byte loads and stores, two rows paired in
//      registers so the LDS traffic is dwords — 186 -> 85 us at [16448, 6144]).
//   2. syrk_kernel: one workgroup (4 waves, 2x2, 64x64 each = 2x2 MFMA tiles) per tile pair,
//      K in chunks of 64 staged through LDS (rows padded to 144 B: conflict-free ds_read_b128).
//      Few tiles, long K and C % 8 != 0: 4 K-slices (blockIdx.y) into partial buffers +
//      syrk_combine_kernel (slice order).
//   3. syrk256_kernel (C > 4096 and a long K): 256-wide tiles, one persistent workgroup per CU of
//      8 waves (two per SIMD, 128x64 each), reading X AS IT LIES — no transposed copy: token-major
//      stages in LDS, MFMA operands through ds_read_b64_tr_b16.  Tiles beyond a whole round are cut
//      into K-slices whose fp32 slabs the last slice to arrive (ticket) adds up in slice order.
//      Few tiles and a long K (C = 1408 at 8 samples per call: 21 tiles for 256 CUs): every tile
//      cut into K-slices that write partial Hessians, syrk_combine_kernel adds them up.
//      (ECOFLAP_SYRK_FORM=1: the same on the transposed copy — taken when C % 8 != 0 —, =0: 4 waves
//      of 128x128 on the transposed copy; A/B switches.)
// Every path sums in a fixed order: the same call gives the same bits, whoever finishes last.
// MFMA roofline: 2 * T * C * (C + 128) / 2 flops per call against the dense fp16/bf16 peak.
#include "common.h"
#include <cstdio>
#include <cstdlib>

#define SY_TILE 128
#define SY_KC 64
#define SY_ROWB (SY_KC * 2 + 16)      // LDS bytes per staged row (128 B of k + 16 B pad)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

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

// The same for C % 8 == 0 (every shape of the models): 16-byte loads along c and 16-byte stores
// along t — two rows (t, t + 1) are paired in registers, so the LDS traffic is whole dwords
// (tile[c][t / 2], 33 dwords per row: 2-way conflicts on the writes, none on the reads).
__global__ __launch_bounds__(256) void syrk_transpose8_kernel(const uint16_t* __restrict__ x,
                                                              int64_t T, int64_t C, int64_t Kpad,
                                                              uint16_t* __restrict__ xt) {
    __shared__ uint32_t tile[64 * 33];
    const int64_t c0 = (int64_t)blockIdx.x * 64, t0 = (int64_t)blockIdx.y * 64;
    const int tid = threadIdx.x;
    {
        const int p = tid >> 3, cs = tid & 7;                 // rows t0 + 2p, t0 + 2p + 1; columns c0 + 8 cs ..
        const int64_t t = t0 + 2 * p, c = c0 + 8 * cs;
        u32x4 lo = {0u, 0u, 0u, 0u}, hi = {0u, 0u, 0u, 0u};
        if (c < C) {
            if (t < T) lo = *(const u32x4*)(x + t * C + c);
            if (t + 1 < T) hi = *(const u32x4*)(x + (t + 1) * C + c);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {                          // columns 2j, 2j + 1 of the eight
            tile[(8 * cs + 2 * j) * 33 + p] = __builtin_amdgcn_perm(hi[j], lo[j], 0x05040100u);
            tile[(8 * cs + 2 * j + 1) * 33 + p] = __builtin_amdgcn_perm(hi[j], lo[j], 0x07060302u);
        }
    }
    __syncthreads();
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int cr = (tid >> 3) + 32 * half, ts = tid & 7;   // row c0 + cr of Xt, t0 + 8 ts ..
        const int64_t c = c0 + cr, t = t0 + 8 * ts;
        u32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = tile[cr * 33 + 4 * ts + i];
        if (c < C && t < Kpad) *(u32x4*)(xt + c * Kpad + t) = v;
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
// blockIdx.y = K-slice (few tiles, long K — C = 1408 with 8 samples per call: 66 tiles for 256 CUs):
// slice y covers k in [y * kslice, min(Kpad, (y + 1) * kslice)) and writes alpha * X_y^T X_y to
// H + y * hstride (the caller passes beta = 0 and a partial buffer per slice; syrk_combine_kernel adds
// them up in slice order).  One slice: kslice = Kpad, hstride = 0.
template <int DT, int WT>
__global__ __launch_bounds__(256) void syrk_kernel(const uint16_t* __restrict__ xt, int64_t C,
                                                   int64_t Kpad, float* __restrict__ H, float beta,
                                                   float alpha, int ntiles, int64_t kslice,
                                                   int64_t hstride) {
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
    const int64_t kbeg = (int64_t)blockIdx.y * kslice;
    const int64_t kend = kbeg + kslice < Kpad ? kbeg + kslice : Kpad;
    H += (int64_t)blockIdx.y * hstride;
    fetch(kbeg);
    for (int64_t k0 = kbeg; k0 < kend; k0 += SY_KC) {
#pragma unroll
        for (int i = 0; i < LD_PER_THREAD; ++i) {
            const int idx = tid + 256 * i, row = idx >> 3, seg = idx & 7;
            *(u32x4*)(ldsA + row * SY_ROWB + seg * 16) = ra[i];
            *(u32x4*)(ldsB + row * SY_ROWB + seg * 16) = rb[i];
        }
        __syncthreads();
        if (k0 + SY_KC < kend) fetch(k0 + SY_KC);
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

// ---- 256x256 tiles (wide matrices, long K): with 2x2 MFMA tiles per wave (the kernel above) every
// k-step of 4 MFMAs needs 4 ds_read_b128 — at 4 waves per CU the LDS is as busy as the matrix
// cores and the loop tops out near 560 TFLOP/s.  Here a wave owns 128x64 (8 waves per workgroup,
// two per SIMD: 6 reads per 8 MFMAs, and one wave's barrier / LDS wait is the other's MFMA time)
// or 128x128 (4 waves, one per SIMD: 8 reads per 16 MFMAs — kept for A/B: it pays every wait in
// full and ends up slower).  One persistent workgroup per CU.  K in stages of 32: global ->
// registers (two stages of loads in flight, the compiler's counted vmcnt) -> one of TWO LDS stage
// buffers; the MFMA operands of k-step j+1 are read from LDS while the MFMAs of step j run (two
// register sets), with the issue order of MFMAs / LDS reads / LDS writes / global loads fixed by
// sched_group_barrier.  (LDS-DMA into four stage buffers with counted vmcnt was tried first: the
// same speed, `profiles/r03_secondary/syrk_notes.md`.)
// Transposed-copy form (TRX = false): LDS rows are 64 B without padding, the 16-byte k-segment s
// of row rho sits in slot s ^ ((rho >> 2) & 3): each of ds_read_b128's 16-lane groups ({0-3,12-15,
// 20-27}, {4-11,16-19,28-31}, + 32) then covers all 64 banks once; the swizzle is applied on the
// GLOBAL side (the lane that fills slot p fetches segment p ^ swz).
#define S2_T 256
#define S2_KC 32
#define S2_OPB (S2_T * S2_KC * 2)      // bytes per operand and stage (16 KB)
#define S2_STG (2 * S2_OPB)
#define S2_NBUF 2

// Work split of one call: the chip runs `G` workgroups (one per CU), the upper triangle has
// `ntiles` tiles.  `full` = the largest multiple of G tiles are computed whole, one round each;
// the `left` tiles that would make a nearly empty extra round (C = 6144: 300 tiles on 256 CUs)
// are cut into `S` K-slices each, one slice per workgroup ahead of its whole tile: the last slice
// of a tile to finish (ticket) sums the S fp32 slabs IN SLICE ORDER and writes the tile — the
// result does not depend on who arrives when.
struct Syrk256Plan {
    int ntiles, full, left, S, G;
    float* slabs;            // [left * S][256 * 256] fp32, in the accumulators' (tile, register, thread) order
    unsigned* tickets;       // [left], zeroed ahead of the launch
    const void* zero16;      // 16 zero bytes (the zeroed words after the tickets)
    // "partials" form (few tiles, long K — C = 1408: 21 tiles for 256 CUs): part_S > 0, every tile is
    // cut into part_S K-slices, item t = tile * part_S + slice writes alpha * X_s^T X_s of its tile
    // (beta = 0, mirror included) to the slice's OWN Hessian at part + slice * part_stride — no
    // hand-off; syrk_combine_kernel adds the partial Hessians up in slice order afterwards
    int part_S;
    float* part;
    int64_t part_stride;
};

// NW = 4: waves 2 x 2, 128x128 each, one per SIMD.  NW = 8: waves 2 x 4, 128x64 each, TWO per SIMD
// (128 accumulator registers): one wave's barrier / LDS wait is the other's MFMA time, at 6 LDS
// reads per 8 MFMAs instead of 8 per 16.
// TRX: X is read as it lies, [tokens][C] — no transposed copy.  A stage is then 32 token rows of
// 256 columns per operand (full 512-byte runs from memory), staged token-major in LDS with rows
// of 576 B (4 consecutive rows cover all 64 banks once for the 8-byte transposed reads), and the
// MFMA operands (8 consecutive k of one column per lane) come from `ds_read_b64_tr_b16`: lane
// 4q + p of a 16-lane group addresses row q, columns 4p..4p+3 of a 4 x 16 block and receives
// column (lane & 15) of the 4 rows.  `xt` = X, `Kpad` = the token count (any), rows past it and
// 16-byte column runs past C are read from P.zero16.
#define S2_TRP 576                                    // LDS bytes per token row of an operand (512 + 64)
template <int DT, int NW, bool TRX>
__global__ __launch_bounds__(64 * NW) void syrk256_kernel(const uint16_t* __restrict__ xt, int64_t C,
                                                          int64_t Kpad, float* __restrict__ H,
                                                          float beta, float alpha, Syrk256Plan P) {
    constexpr int NI = NW == 4 ? 4 : 2;              // MFMA tiles per wave along the columns
    constexpr int NP = 16 / NW;                      // loads per thread, operand and stage
    constexpr int NT = 64 * NW;
    constexpr int MIRROR_BYTES = NW * 32 * 65 * 4;   // per wave [32 cols][64 rows + 1] floats
    constexpr int OPB = TRX ? S2_KC * S2_TRP : S2_OPB;        // bytes per operand and stage
    constexpr int STG = 2 * OPB;
    __shared__ __attribute__((aligned(16))) unsigned char
        lds[S2_NBUF * STG > MIRROR_BYTES ? S2_NBUF * STG : MIRROR_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = NW == 4 ? wave >> 1 : wave >> 2, wn = NW == 4 ? wave & 1 : wave & 3;
    const int r = lane & 31, h = lane >> 5;
    const int nst_all = TRX ? 2 * (int)((Kpad + 2 * S2_KC - 1) / (2 * S2_KC)) : (int)(Kpad / S2_KC);
    const int npairs = nst_all / 2;
    // this workgroup's items: its slice (if any), then its whole tiles (XCD x = blockIdx % 8 takes a
    // contiguous range of the column-major tile order: neighbours share panels in ITS L2).  (Tried:
    // tiles first and the slices drawn from a counter, 4 / 5 / 8 slices per left-over tile, with 4
    // and with 8 waves: 5-15 % slower every time — the slices and their reducers then sit at the
    // END of the launch; write-through slab stores instead of the release fence: 40 % slower.)
    const int b = (int)blockIdx.x;
    const int nslices = P.left * P.S;
    const int per = (P.G + 7) / 8;
    const int bperm = (b % 8) * per + b / 8;              // a permutation of 0..G-1 when G % 8 == 0
    for (int item = b < nslices ? -1 : 0;; ++item) {
        int t, st0 = 0, st1 = nst_all, slab = -1;
        if (item < 0) {
            t = P.full + b / P.S;
            const int sl = b % P.S;
            st0 = 2 * (int)((int64_t)sl * npairs / P.S);
            st1 = 2 * (int)((int64_t)(sl + 1) * npairs / P.S);
            slab = b;
        } else {
            t = item * P.G + (P.G % 8 == 0 ? bperm : b);
            if (t >= P.full) break;
        }
        float* Hout = H;
        float beta_l = beta;
        int tile = t;
        if (P.part_S) {
            tile = t / P.part_S;
            const int sl = t - tile * P.part_S;
            st0 = 2 * (int)((int64_t)sl * npairs / P.part_S);
            st1 = 2 * (int)((int64_t)(sl + 1) * npairs / P.part_S);
            Hout = P.part + (int64_t)sl * P.part_stride;
            beta_l = 0.f;
        }
        int bj = (int)((__builtin_sqrtf(8.0f * (float)tile + 1.0f) - 1.0f) * 0.5f);
        while ((bj + 1) * (bj + 2) / 2 <= tile) ++bj;
        while (bj * (bj + 1) / 2 > tile) --bj;
        const int bi = tile - bj * (bj + 1) / 2;
        const int64_t rowA0 = (int64_t)bi * S2_T, rowB0 = (int64_t)bj * S2_T;

        // global -> LDS image.  Transposed copy: piece p (16 rows x 64 B) of an operand goes to wave
        // p % NW; lane -> row 16 p + (lane >> 2), slot lane & 3, which holds segment
        // slot ^ ((row >> 2) & 3).  TRX: thread -> token row (tid >> 5) of a pass of NT / 32 rows,
        // 16-byte column run tid & 31 of the operand's 256 columns.
        const int q = lane >> 2;
        const int seg = (lane & 3) ^ ((lane >> 4) & 3);
        const uint16_t* pa[NP];
        const uint16_t* pb[NP];
        bool colA = true, colB = true;
        if (TRX) {
            colA = rowA0 + 8 * (tid & 31) + 8 <= C;
            colB = rowB0 + 8 * (tid & 31) + 8 <= C;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                pa[i] = xt + (int64_t)(i * (NT / 32) + (tid >> 5)) * C + rowA0 + 8 * (tid & 31);
                pb[i] = xt + (int64_t)(i * (NT / 32) + (tid >> 5)) * C + rowB0 + 8 * (tid & 31);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                int64_t ra = rowA0 + 16 * (wave + NW * i) + q, rb = rowB0 + 16 * (wave + NW * i) + q;
                ra = ra < C ? ra : C - 1;        // rows past C: any valid row (their products are dropped)
                rb = rb < C ? rb : C - 1;
                pa[i] = xt + ra * Kpad + seg * 8;
                pb[i] = xt + rb * Kpad + seg * 8;
            }
        }
        // (a diagonal tile loads its panel twice: the loop is the same for every tile — 24 of 300)

        f32x16 acc[4][NI];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

        const int swz = (r >> 2) & 3;
        // transposed copy: row-major operand rows, k-steps 0 / 1 of a stage are segments h, 2 + h;
        // TRX: token rows 8 h + q (+ 4) of k-step 0 / 16 + the same of k-step 1, columns
        // 16 (group & 1) + 4 p of the fragment's 32
        const int trl = (8 * h + ((lane & 15) >> 2)) * S2_TRP + 2 * (16 * ((lane >> 4) & 1) + 4 * (lane & 3));
        const int offA = TRX ? 2 * (wm * 128) + trl : (wm * 128 + r) * 64;
        const int offB = TRX ? 2 * (wn * (32 * NI)) + trl + OPB : (wn * (32 * NI) + r) * 64 + OPB;
        const int so0 = TRX ? 0 : (h ^ swz) << 4, so1 = TRX ? 16 * S2_TRP : ((2 + h) ^ swz) << 4;
        u32x4 fa0[4], fb0[NI], fa1[4], fb1[NI];
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
#define S2_TR(ADDR) __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ADDR)))
#define S2_READ1(F, ADDR_ROWMAJOR, ADDR_TR)                                  \
    if (TRX) {                                                               \
        const uint2 x0_ = S2_TR(ADDR_TR), x1_ = S2_TR((ADDR_TR) + 4 * S2_TRP); \
        F = u32x4{x0_.x, x0_.y, x1_.x, x1_.y};                               \
    } else {                                                                 \
        F = *(const u32x4*)(ADDR_ROWMAJOR);                                  \
    }
#define S2_READ(FA, FB, BASE, SO)                                            \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                         \
        S2_READ1(FA[i_], (BASE) + offA + i_ * 2048 + (SO), (BASE) + offA + i_ * 64 + (SO)) \
    _Pragma("unroll") for (int i_ = 0; i_ < NI; ++i_)                        \
        S2_READ1(FB[i_], (BASE) + offB + i_ * 2048 + (SO), (BASE) + offB + i_ * 64 + (SO))
#define S2_MFMA(FA, FB)                                                      \
    _Pragma("unroll") for (int mi_ = 0; mi_ < 4; ++mi_)                      \
        _Pragma("unroll") for (int ni_ = 0; ni_ < NI; ++ni_)                 \
            acc[mi_][ni_] = Mfma<DT>::run(FA[mi_], FB[ni_], acc[mi_][ni_]);
        // global -> registers -> LDS, two register stages in flight (the loads of stage i + 2 and
        // i + 3 are outstanding while stage i computes; the compiler's vmcnt is counted, in order)
        u32x4 ra0[NP], rb0[NP], ra1[NP], rb1[NP];
        // lane-linear image of the transposed copy (row 16 p + (lane >> 2), slot lane & 3) / TRX:
        // token row (tid >> 5) of the pass, 16-byte run tid & 31
        const int wofs = TRX ? (tid >> 5) * S2_TRP + 16 * (tid & 31) : 1024 * wave + 16 * lane;
        constexpr int WSTRIDE = TRX ? (NT / 32) * S2_TRP : 1024 * NW;        // LDS bytes between a thread's loads
        const int64_t klast = (int64_t)(st1 - 1) * S2_KC;
        const uint16_t* zero16 = (const uint16_t*)P.zero16;
#define S2_GLOAD(RA, RB, ST)                                                 \
    {                                                                        \
        const int64_t k0_ = (ST) < st1 ? (int64_t)(ST) * S2_KC : klast;      \
        _Pragma("unroll") for (int i_ = 0; i_ < NP; ++i_) {                  \
            if (TRX) {                                                       \
                const bool tok_ = k0_ + i_ * (NT / 32) + (tid >> 5) < Kpad;  \
                RA[i_] = *(const u32x4*)(tok_ && colA ? pa[i_] + k0_ * C : zero16); \
                RB[i_] = *(const u32x4*)(tok_ && colB ? pb[i_] + k0_ * C : zero16); \
            } else {                                                         \
                RA[i_] = *(const u32x4*)(pa[i_] + k0_);                      \
                RB[i_] = *(const u32x4*)(pb[i_] + k0_);                      \
            }                                                                \
        }                                                                    \
    }
#define S2_LWRITE(RA, RB, BUF)                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < NP; ++i_) {                      \
        *(u32x4*)((BUF) + wofs + WSTRIDE * i_) = RA[i_];                     \
        *(u32x4*)((BUF) + OPB + wofs + WSTRIDE * i_) = RB[i_];               \
    }
    // all of this wave's LDS traffic is done (its reads of the buffer about to be overwritten next
    // and its writes), then everybody's; global loads stay in flight (no vmcnt here)
#define S2_SYNC()                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       \
    __builtin_amdgcn_s_barrier();                                            \
    asm volatile("" ::: "memory")
        // ONE uniform loop over stage PAIRS (no tail code: the 256 accumulators stay put only if
        // they live through a single loop; stage numbers past the end are clamped, the duplicates
        // land in a buffer nobody reads again).
        unsigned char* buf0 = lds;
        unsigned char* buf1 = lds + STG;
        __syncthreads();                          // the previous item's epilogue is out of the LDS
        S2_GLOAD(ra0, rb0, st0);
        S2_GLOAD(ra1, rb1, st0 + 1);
        S2_LWRITE(ra0, rb0, buf0);
        S2_SYNC();
        S2_READ(fa0, fb0, buf0, so0);
        S2_GLOAD(ra0, rb0, st0 + 2);
    // Issue order inside a half stage (4 NI MFMAs; whatever is not placed in an MFMA's shadow is
    // paid in full, also with two waves per SIMD: left to the compiler the 8-wave form runs 990 us
    // instead of 867): MFMA, LDS read, {LDS write | global load} while there are any, then bare
    // MFMAs under which the LDS traffic drains before the barrier / the next half needs it.
#define S2_PIPE(SECOND_MASK)                                                 \
    _Pragma("unroll") for (int g_ = 0; g_ < 4 + NI; ++g_) {                  \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                   \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                   \
        if (g_ < 2 * NP) __builtin_amdgcn_sched_group_barrier(SECOND_MASK, 1, 0); \
    }                                                                        \
    __builtin_amdgcn_sched_group_barrier(0x008, 4 * NI - (4 + NI), 0);       \
    __builtin_amdgcn_sched_barrier(0)
#define S2_STAGE(CUR, NXT, RA, RB, ST)                                       \
    S2_READ(fa1, fb1, CUR, so1);                                             \
    S2_LWRITE(RA, RB, NXT);                                                  \
    S2_MFMA(fa0, fb0);                                                       \
    S2_PIPE(0x200);                                                          \
    S2_SYNC();                                                               \
    S2_READ(fa0, fb0, NXT, so0);                                             \
    S2_GLOAD(RA, RB, (ST) + 3);                                              \
    S2_MFMA(fa1, fb1);                                                       \
    S2_PIPE(0x020);
        for (int i = st0; i < st1; i += 2) {      // an even number of stages
            S2_STAGE(buf0, buf1, ra1, rb1, i);
            S2_STAGE(buf1, buf0, ra0, rb0, i + 1);
        }
#undef S2_STAGE
#undef S2_PIPE
#undef S2_SYNC
#undef S2_LWRITE
#undef S2_GLOAD
#undef S2_READ
#undef S2_READ1
#undef S2_TR
#undef S2_MFMA
        __syncthreads();

        if (slab >= 0) {
            // ---- a K-slice: the raw sums go to this slice's slab; the tile's last slice to arrive
            // adds the slabs up in slice order (hand-off: every wave drains its stores, barrier, ONE
            // agent-scope release ahead of one lane's ticket; the reducer takes ONE agent-scope acquire
            // ahead of its plain loads — right for any placement of the slices over CUs / XCDs)
            float* mine = P.slabs + (int64_t)slab * (S2_T * S2_T);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        f32x4 v = {acc[mi][ni][4 * e4], acc[mi][ni][4 * e4 + 1], acc[mi][ni][4 * e4 + 2],
                                   acc[mi][ni][4 * e4 + 3]};
                        *(f32x4*)(mine + ((((mi * NI + ni) * 4 + e4) * NT + tid) << 2)) = v;
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            unsigned* flag = (unsigned*)lds;                 // (the staging LDS is free now)
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned got = __hip_atomic_fetch_add(&P.tickets[t - P.full], 1u, __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_AGENT);
                const bool last = got == (unsigned)(P.S - 1);
                if (last) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                *flag = last ? 1u : 0u;
            }
            __syncthreads();
            const bool last = *flag != 0u;
            __syncthreads();
            if (!last) continue;
            const float* base = P.slabs + (int64_t)(t - P.full) * P.S * (S2_T * S2_T);
            for (int sl = 0; sl < P.S; ++sl) {
                const float* src = base + (int64_t)sl * (S2_T * S2_T);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) {
                            const f32x4 v = *(const f32x4*)(src + ((((mi * NI + ni) * 4 + e4) * NT + tid) << 2));
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                acc[mi][ni][4 * e4 + c] = sl == 0 ? v[c] : acc[mi][ni][4 * e4 + c] + v[c];
                        }
            }
        }

        // ---- epilogue (as above): H = beta H + alpha acc, the value kept for the mirror
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int64_t gc = rowB0 + wn * (32 * NI) + ni * 32 + r;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t gr = rowA0 + wm * 128 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    float v = alpha * acc[mi][ni][e];
                    if (gr < C && gc < C) {
                        if (beta_l != 0.f) v += beta_l * Hout[gr * C + gc];
                        Hout[gr * C + gc] = v;
                    }
                    acc[mi][ni][e] = v;
                }
            }
        if (bi == bj) continue;
        float* ldsT = (float*)lds + wave * (32 * 65);          // per wave [32 cols][64 rows + 1]
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int hm = 0; hm < 2; ++hm) {                   // rows 64 hm .. 64 hm + 63 of the wave's 128
                __syncthreads();                               // the previous part is out
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        ldsT[r * 65 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h] = acc[2 * hm + mi][ni][e];
                __syncthreads();
                const int64_t gr = rowA0 + wm * 128 + 64 * hm + lane;      // original row -> mirrored column
                for (int c = 0; c < 32; ++c) {
                    const int64_t gc = rowB0 + wn * (32 * NI) + ni * 32 + c;
                    if (gc < C && gr < C) Hout[gc * C + gr] = ldsT[c * 65 + lane];
                }
            }
    }
}

static int syrk_cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            return 256;                       // no device (symbol checks on a CPU box): MI355X's count
        n = cus;
    }
    return n;
}

// the plan's numbers for `cols` columns and Kpad (slabs / tickets not filled in)
static Syrk256Plan syrk256_plan(int64_t cols, int64_t kpad) {
    Syrk256Plan p{};
    const int64_t n2 = (cols + S2_T - 1) / S2_T;
    p.ntiles = (int)(n2 * (n2 + 1) / 2);
    const int cus = syrk_cu_count();
    const int npairs = (int)(kpad / (2 * S2_KC));
    p.G = p.ntiles < cus ? p.ntiles : cus;
    p.full = p.ntiles / p.G * p.G;
    p.left = p.ntiles - p.full;
    p.S = 0;
    if (p.left > 0) {                        // one slice per workgroup at most
        p.S = p.G / p.left;
        if (p.S > 8) p.S = 8;
        if (p.S > npairs) p.S = npairs;
        if (p.S < 1) p.S = 1;
    }
    return p;
}

// H = beta * H + P[0] + P[1] + ... (slice order), 4 floats per thread
__global__ __launch_bounds__(256) void syrk_combine_kernel(float* __restrict__ H, const float* __restrict__ P,
                                                           int64_t n4, int64_t hstride, int S, float beta) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 a = ((const f32x4*)P)[i];
    for (int y = 1; y < S; ++y) {
        const f32x4 b = ((const f32x4*)(P + (int64_t)y * hstride))[i];
        a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
    }
    if (beta != 0.f) {
        const f32x4 h = ((const f32x4*)H)[i];
        a[0] += beta * h[0]; a[1] += beta * h[1]; a[2] += beta * h[2]; a[3] += beta * h[3];
    }
    ((f32x4*)H)[i] = a;
}

// Few tiles and a long K (C = 1408 at 8 samples per call): K-slices into partial Hessians + one combine
// pass.  -> number of slices (1 = none): as many as give every CU a 256-wide tile's slice, at most 16
static inline int syrk_kslices(int64_t cols, int64_t kpad) {
    if (!(cols >= 1024 && cols < 2048 && cols % 4 == 0 && kpad >= 4096)) return 1;
    const int64_t n2 = (cols + 255) / 256, nt = n2 * (n2 + 1) / 2;
    int S = (int)(syrk_cu_count() / nt);
    if (S > 16) S = 16;
    if (S > (int)(kpad / 128)) S = (int)(kpad / 128);
    return S < 2 ? 1 : S;
}

static inline int64_t syrk_kpad(int64_t tokens) { return (tokens + SY_KC - 1) / SY_KC * SY_KC; }

static inline size_t syrk_xt_bytes(int64_t tokens, int64_t cols) {
    return ((size_t)cols * (size_t)syrk_kpad(tokens) * 2 + 255) / 256 * 256;      // Xt, 16-bit
}

// 256-wide tiles: wide matrices with a long K.  More tiles than CUs (C = 6144: 300) need the
// K-sliced left-overs, whose slabs only pay at kpad >= 4096; at most one tile per CU (C = 5120:
// 210) nothing is sliced and kpad >= 2048 is enough.  (Measured, us per call old / new kernel:
// [16448, 6144] 1224 / 793, [3072, 5120] 177 / 152, [2056, 6144] 200 / 218, [384, 5120] 74 / 78.)
static inline bool syrk_use256(int64_t cols, int64_t kpad) {
    static const int64_t min_cols = getenv("ECOFLAP_SYRK_MINCOLS") ? atoll(getenv("ECOFLAP_SYRK_MINCOLS")) : 4097;
    static const int64_t min_k = getenv("ECOFLAP_SYRK_MINK") ? atoll(getenv("ECOFLAP_SYRK_MINK")) : 2048;
    if (cols < min_cols || kpad < min_k) return false;
    const int64_t n2 = (cols + S2_T - 1) / S2_T;
    return n2 * (n2 + 1) / 2 <= syrk_cu_count() || kpad >= 2 * min_k;
}

extern "C" size_t ecoflap_hessian_workspace_bytes(int64_t tokens, int64_t cols) {
    if (tokens <= 0 || cols <= 0) return 0;
    size_t n = syrk_xt_bytes(tokens, cols);
    const int ks = syrk_kslices(cols, syrk_kpad(tokens));
    if (ks > 1) n += 256 + (size_t)ks * (((size_t)cols * cols * sizeof(float) + 255) / 256 * 256);
    if (syrk_use256(cols, syrk_kpad(tokens))) {              // K-slice slabs + tickets of the 256-wide kernel
        const Syrk256Plan p = syrk256_plan(cols, syrk_kpad(tokens));
        n += (size_t)p.left * p.S * S2_T * S2_T * sizeof(float) + (16 + (size_t)(p.left + 1) * sizeof(unsigned) + 255) / 256 * 256;
    }
    return n;
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
    static const bool form2 = !getenv("ECOFLAP_SYRK_FORM") || atoi(getenv("ECOFLAP_SYRK_FORM")) == 2;
    static const bool no256_ = getenv("ECOFLAP_SYRK_NO256") != nullptr;
    const bool direct = form2 && !no256_ && cols % 8 == 0 && aligned16(x) &&
                        (syrk_use256(cols, kpad) || syrk_kslices(cols, kpad) > 1);
    if (direct) {
        // the 256-wide kernel reads X as it lies (no transposed copy)
    } else if (cols % 8 == 0 && aligned16(x))
        hipLaunchKernelGGL(syrk_transpose8_kernel, dim3((unsigned)((cols + 63) / 64), (unsigned)(kpad / 64)),
                           dim3(256), 0, s, (const uint16_t*)x, tokens, cols, kpad, xt);
    else
        hipLaunchKernelGGL(syrk_transpose_kernel, dim3((unsigned)((cols + 63) / 64), (unsigned)(kpad / 64)),
                           dim3(256), 0, s, (const uint16_t*)x, tokens, cols, kpad, xt);
    ECO_CHECK_LAUNCH();
    // (:79-81) H *= n/(n+b); n += b; inp = sqrt(2/n) x  ->  alpha = 2/n_new on x^T x
    const float beta = (float)((double)nsamples_before / (double)(nsamples_before + batch));
    const float alpha = (float)(2.0 / (double)(nsamples_before + batch));
    static const bool no256 = getenv("ECOFLAP_SYRK_NO256") != nullptr;     // A/B switch
    if (syrk_use256(cols, kpad) && !no256) {     // (short K: the slabs' hand-off costs more than the balance returns)
        Syrk256Plan p = syrk256_plan(cols, kpad);
        char* after = (char*)workspace + syrk_xt_bytes(tokens, cols);
        p.slabs = (float*)after;
        char* zt = after + (size_t)p.left * p.S * S2_T * S2_T * sizeof(float);
        p.zero16 = zt;                                   // 16 zero bytes, then the tickets: one memset
        p.tickets = (unsigned*)(zt + 16);
        {
            const hipError_t e = hipMemsetAsync(zt, 0, 16 + (size_t)(p.left + 1) * sizeof(unsigned), s);
            if (e != hipSuccess) return (int)e;
        }
        static const int form = getenv("ECOFLAP_SYRK_FORM") ? atoi(getenv("ECOFLAP_SYRK_FORM")) : 2;   // A/B: 0 = 4 waves, 1 = 8 waves, 2 (default) = 8 waves on X as it lies
        const uint16_t* xin = (const uint16_t*)x;
#define S2_GO(DT_, NW_, TRX_, SRC_, K_) hipLaunchKernelGGL((syrk256_kernel<DT_, NW_, TRX_>), dim3((unsigned)p.G), dim3(64 * NW_), 0, s, SRC_, cols, K_, H, beta, alpha, p)
        if (form == 2 && direct) {
            if (dtype == ECOFLAP_F16) S2_GO(ECOFLAP_F16, 8, true, xin, tokens); else S2_GO(ECOFLAP_BF16, 8, true, xin, tokens);
        } else if (form >= 1) {
            if (dtype == ECOFLAP_F16) S2_GO(ECOFLAP_F16, 8, false, xt, kpad); else S2_GO(ECOFLAP_BF16, 8, false, xt, kpad);
        } else {
            if (dtype == ECOFLAP_F16) S2_GO(ECOFLAP_F16, 4, false, xt, kpad); else S2_GO(ECOFLAP_BF16, 4, false, xt, kpad);
        }
#undef S2_GO
        ECO_CHECK_LAUNCH();
        return 0;
    }
    const int ks = syrk_kslices(cols, kpad);
    if (ks > 1) {
        // few tiles, long K: K-slices into partial Hessians, then one combine pass (slice order)
        const int64_t hstride = (int64_t)(((size_t)cols * cols * sizeof(float) + 255) / 256 * 256 / sizeof(float));
        char* after = (char*)workspace + syrk_xt_bytes(tokens, cols);
        float* part = (float*)(after + 256);                       // (16 zero bytes first)
        const bool direct8 = direct;
        if (direct8) {
            // the 8-wave 256-wide kernel on X as it lies, one (tile, slice) per workgroup
            const hipError_t e = hipMemsetAsync(after, 0, 16, s);
            if (e != hipSuccess) return (int)e;
            Syrk256Plan p{};
            const int64_t n2 = (cols + S2_T - 1) / S2_T;
            p.ntiles = (int)(n2 * (n2 + 1) / 2);
            p.G = p.ntiles * ks;
            p.full = p.G;
            p.zero16 = after;
            p.part_S = ks;
            p.part = part;
            p.part_stride = hstride;
            if (dtype == ECOFLAP_F16)
                hipLaunchKernelGGL((syrk256_kernel<ECOFLAP_F16, 8, true>), dim3((unsigned)p.G), dim3(512), 0, s, (const uint16_t*)x, cols, tokens, H, beta, alpha, p);
            else
                hipLaunchKernelGGL((syrk256_kernel<ECOFLAP_BF16, 8, true>), dim3((unsigned)p.G), dim3(512), 0, s, (const uint16_t*)x, cols, tokens, H, beta, alpha, p);
        } else {
            const int64_t n128 = (cols + 127) / 128;
            const int nt128s = (int)(n128 * (n128 + 1) / 2);
            const int64_t kslice = (kpad / SY_KC + ks - 1) / ks * SY_KC;
            const dim3 g((unsigned)((nt128s + 7) / 8 * 8), (unsigned)ks);
            if (dtype == ECOFLAP_F16)
                hipLaunchKernelGGL((syrk_kernel<ECOFLAP_F16, 2>), g, dim3(256), 0, s, xt, cols, kpad, part, 0.f, alpha, nt128s, kslice, hstride);
            else
                hipLaunchKernelGGL((syrk_kernel<ECOFLAP_BF16, 2>), g, dim3(256), 0, s, xt, cols, kpad, part, 0.f, alpha, nt128s, kslice, hstride);
        }
        ECO_CHECK_LAUNCH();
        const int64_t n4 = cols * cols / 4;
        hipLaunchKernelGGL(syrk_combine_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, H, part, n4, hstride, ks, beta);
        ECO_CHECK_LAUNCH();
        return 0;
    }
    const int ntiles = (int)(nt * (nt + 1) / 2);
    const dim3 grid((unsigned)((ntiles + 7) / 8 * 8));
#define SYRK_GO(DT_, WT_) hipLaunchKernelGGL((syrk_kernel<DT_, WT_>), grid, dim3(256), 0, s, xt, cols, kpad, H, beta, alpha, ntiles, kpad, (int64_t)0)
    if (dtype == ECOFLAP_F16) { if (wt == 2) SYRK_GO(ECOFLAP_F16, 2); else SYRK_GO(ECOFLAP_F16, 1); }
    else { if (wt == 2) SYRK_GO(ECOFLAP_BF16, 2); else SYRK_GO(ECOFLAP_BF16, 1); }
#undef SYRK_GO
    ECO_CHECK_LAUNCH();
    return 0;
}
