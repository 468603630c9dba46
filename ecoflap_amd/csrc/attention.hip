// attention.hip — multi-head self-attention of the EVA ViT-g blocks for gfx950 (plumbing for the
// build's shape modules, include/ecoflap_shape_ops.h; not part of the pruner ABI).
//
// The zeroth-order loop is forward-bound, and the library's fused attention runs the ViT-g shape
// (257 tokens, 16 heads of 88) at 67 TFLOP/s: 0.7 ms per block at 16 concatenated evaluations, a
// sixth of a ViT-g matrix's step.  This kernel serves exactly that shape family (N <= 288 tokens,
// head width <= 96, fp16, no mask, no dropout):
//
//   * one workgroup (4 waves) per (image, head); the head's K [N, D] and V^T [D, N] live in LDS
//     for the whole workgroup (K rows at a 16*odd-byte pitch, V^T rows likewise: every
//     ds_read_b128 of 16 consecutive rows covers all 64 banks once);
//   * a wave takes 32 queries at a time: S^T = K Q^T by v_mfma_f32_32x32x16_f16 (queries on the
//     lanes, keys in the registers: a query's softmax is a per-lane loop plus one exchange with
//     lane ^ 32), all N scores of the tile in registers, exact two-pass softmax in fp32;
//   * O^T = V^T P^T takes the P tile straight from those registers as the B operand (the MFMA's k
//     order inside a 16-key step is 8(j>>2) + 4h + (j&3): V^T is stored with its keys in that order,
//     so the A operand is one ds_read_b128 too);
//   * reads qkv as the qkv Linear wrote it ([B, N, 3, H, D]) and writes [B, N, H*D]: no permute /
//     contiguous copies around it.
// Per (image, head) independent: batch invariant by construction.
#include <stdlib.h>

#include "common.h"
#include "../../include/ecoflap_shape_ops.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define AT_NT 9                 // tiles of 32 keys / queries (N <= 288)
#define AT_ROWS (AT_NT * 32)
#define AT_DC 6                 // 16-wide chunks of the head dimension (D <= 96)
#define AT_DT 3                 // 32-wide tiles of the head dimension
#define AT_KPITCH_MAX 208       // bytes per K row in LDS: 2*D rounded up to 16 * odd
#define AT_VPITCH 192           // bytes per V row in LDS (96 halves: 4 consecutive rows x 64 B cover all banks once)
#define AT_FILL 6               // 16-byte vectors of K and of V per thread and fill round

struct AttnArgs {
    const _Float16* qkv;        // [B, N, 3, H, D]
    _Float16* out;              // [B, N, H * D]
    int B, N, H, D;
    int kpitch;                 // bytes
    float scale_log2e;          // softmax scale * log2(e)
    int debug;                  // ablation switches for tools/attention_launches.py (0 in production)
};

static __device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__global__ __launch_bounds__(256) void vit_attention_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char Ks[AT_ROWS * AT_KPITCH_MAX];
    __shared__ __attribute__((aligned(16))) unsigned char Vs[AT_ROWS * AT_VPITCH];
    const int N = a.N, D = a.D, H = a.H, kpitch = a.kpitch;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row_stride = (int64_t)3 * H * D;
    const _Float16* base = a.qkv + (int64_t)b * N * row_stride + (int64_t)h * D;
    const int nt = (N + 31) >> 5;
    // ---- K and V of this head into LDS, both row-major.  Rows past N: V's must be zero (their
    // probabilities are, and 0 * garbage must stay 0); K's may hold anything (masked below) ----
    {
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int i = N * (AT_VPITCH / 16) + tid; i < AT_ROWS * (AT_VPITCH / 16); i += 256) ((u32x4*)Vs)[i] = z;
    }
    const int r = lane & 31, hh = lane >> 5;
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    // the first tile's Q^T fragments are requested before the fill (their latency hides under it)
    f16x8 qf[AT_DC];
    {
        const int q0 = 32 * wave + r;
        const int qc0 = q0 < N ? q0 : N - 1;
#pragma unroll
        for (int c = 0; c < AT_DC; ++c) {
            const int d0 = 16 * c + 8 * hh;
            qf[c] = (d0 < D && wave < nt) ? *(const f16x8*)(base + (int64_t)qc0 * row_stride + d0) : zero8;
        }
    }
    if (!(a.debug & 1)) {
        // all of a round's global loads are in flight before the first LDS write (one memory
        // latency per round of AT_FILL vectors per thread instead of one per vector)
        const int vpr = D >> 3;
        const int total = N * vpr;
        for (int v0 = tid; v0 < total; v0 += 256 * AT_FILL) {
            u32x4 kv[AT_FILL], vv[AT_FILL];
            int keys[AT_FILL], cs[AT_FILL];
#pragma unroll
            for (int j = 0; j < AT_FILL; ++j) {
                const int v = v0 + 256 * j;
                const int vv_ = v < total ? v : total - 1;
                keys[j] = vv_ / vpr;
                cs[j] = vv_ - keys[j] * vpr;
                const _Float16* src = base + (int64_t)keys[j] * row_stride + 8 * cs[j];
                kv[j] = *(const u32x4*)(src + (int64_t)H * D);
                vv[j] = *(const u32x4*)(src + (int64_t)2 * H * D);
            }
#pragma unroll
            for (int j = 0; j < AT_FILL; ++j) {
                if (v0 + 256 * j >= total) continue;
                const int key = keys[j], c = cs[j];
                *(u32x4*)(Ks + key * kpitch + 16 * c) = kv[j];
                *(u32x4*)(Vs + key * AT_VPITCH + 16 * c) = vv[j];
            }
        }
    }
    __syncthreads();
    for (int qt = wave; qt < nt; qt += 4) {
        const int q = 32 * qt + r;
        // Q^T fragments (B operand: k = d on the half-lanes, column = query on the lanes) of the
        // NEXT tile of this wave, requested now, used after this tile's products
        f16x8 qn[AT_DC];
        {
            const int qx = 32 * (qt + 4) + r;
            const int qcx = qx < N ? qx : N - 1;
#pragma unroll
            for (int c = 0; c < AT_DC; ++c) {
                const int d0 = 16 * c + 8 * hh;
                qn[c] = (d0 < D && qt + 4 < nt) ? *(const f16x8*)(base + (int64_t)qcx * row_stride + d0) : zero8;
            }
        }
        // S^T = K Q^T: tile t holds keys 32t .. 32t+31 (rows, in the registers) x 32 queries (lanes).
        // d-chunk outermost: nine INDEPENDENT accumulators per step (no MFMA waits for the one
        // before it) and the next chunk's nine K fragments are in flight while this chunk's
        // products issue (one wave per SIMD: nobody else hides the LDS latency).  All AT_NT tiles
        // are computed whatever N is (K rows past N are zero; those keys are masked below).
        f32x16 s[AT_NT];
#pragma unroll
        for (int t = 0; t < AT_NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) s[t][i] = 0.f;
        if (!(a.debug & 2)) {
            const unsigned char* krow = Ks + r * kpitch + 16 * hh;
            f16x8 kf[2][AT_NT];
#pragma unroll
            for (int t = 0; t < AT_NT; ++t) kf[0][t] = *(const f16x8*)(krow + 32 * t * kpitch);
#pragma unroll
            for (int c = 0; c < AT_DC; ++c) {
                if (c + 1 < AT_DC) {
#pragma unroll
                    for (int t = 0; t < AT_NT; ++t)
                        kf[(c + 1) & 1][t] = *(const f16x8*)(krow + 32 * t * kpitch + 32 * (c + 1));
                }
                const bool live = 16 * c + 8 * hh < D;      // (a row's last chunk may be half padding)
#pragma unroll
                for (int t = 0; t < AT_NT; ++t)
                    s[t] = mfma16(live ? kf[c & 1][t] : zero8, qf[c], s[t]);
                // (keep the scheduler from hoisting every later chunk's loads up here: 54 fragments
                // in flight at once spill)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // exact softmax over the keys of this lane's query: own registers, then lane ^ 32.
        // Four independent chains for the maximum and for the sum (one wave per SIMD: a serial
        // chain of 144 dependent ops would expose every VALU latency).
        float mx[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
        if (!(a.debug & 4)) {
#pragma unroll
            for (int t = 0; t < AT_NT; ++t) {
                if (32 * t + 32 > N) {          // uniform: only the tile(s) reaching past N mask
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int key = 32 * t + (i & 3) + 8 * (i >> 2) + 4 * hh;
                        s[t][i] = key < N ? s[t][i] : -__builtin_inff();
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) mx[i & 3] = __builtin_fmaxf(mx[i & 3], s[t][i]);
            }
        }
        float m = __builtin_fmaxf(__builtin_fmaxf(mx[0], mx[1]), __builtin_fmaxf(mx[2], mx[3]));
        m = __builtin_fmaxf(m, __shfl_xor(m, 32, 64));
        const float c2 = a.scale_log2e, mc = m * c2;
        float ls[4] = {0.f, 0.f, 0.f, 0.f};
        if (!(a.debug & 4)) {
#pragma unroll
            for (int t = 0; t < AT_NT; ++t) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float p = __builtin_amdgcn_exp2f(s[t][i] * c2 - mc);
                    s[t][i] = p;
                    ls[i & 3] += p;
                }
            }
        }
        float l = (ls[0] + ls[1]) + (ls[2] + ls[3]);
        l += __shfl_xor(l, 32, 64);
        // O^T = V^T P^T: P tile t, k-step st = registers 8 st .. 8 st + 7 of s[t], as they stand
        f32x16 o[AT_DT];
#pragma unroll
        for (int dt = 0; dt < AT_DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
        if (!(a.debug & 8)) {
            // three independent accumulators per step.  V stays row-major in LDS; its transposed
            // A fragments come from ds_read_b64_tr_b16: lane 4q+p of a 16-lane group addresses row q,
            // columns 4p..4p+3 of a 4-key x 16-column block and receives column (lane & 15) of the 4
            // keys - exactly the MFMA's k order (keys 4h..4h+3 and 8+4h..8+4h+3 of the step).
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
            const int grp = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
            const unsigned char* vb = Vs + (4 * hh + qq) * AT_VPITCH + 2 * (16 * (grp & 1) + 4 * pp);
            u32x4 vf[2][AT_DT];
#define AT_TR(KS_, DT_, HALF_) __builtin_amdgcn_ds_read_tr16_b64_v4i16(                          \
        (lds_s16x4)(vb + (16 * (KS_) + 8 * (HALF_)) * AT_VPITCH + 64 * (DT_)))
#define AT_LOADV(BUF_, KS_)                                                                      \
    _Pragma("unroll") for (int dt = 0; dt < AT_DT; ++dt) {                                       \
        const s16x4 x0 = AT_TR(KS_, dt, 0), x1 = AT_TR(KS_, dt, 1);                               \
        const uint2 u0 = __builtin_bit_cast(uint2, x0), u1 = __builtin_bit_cast(uint2, x1);       \
        vf[BUF_][dt] = u32x4{u0.x, u0.y, u1.x, u1.y};                                             \
    }
            AT_LOADV(0, 0)
#pragma unroll
            for (int ks = 0; ks < 2 * AT_NT; ++ks) {          // ks = 2 t + st: 16 keys per step
                if (ks + 1 < 2 * AT_NT) { AT_LOADV((ks + 1) & 1, ks + 1) }
                const int t = ks >> 1, st = ks & 1;
                u32x4 pk;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    pk[j] = Vec<ECOFLAP_F16>::f2h_pk(s[t][8 * st + 2 * j], s[t][8 * st + 2 * j + 1]);
                const f16x8 pf = __builtin_bit_cast(f16x8, pk);
#pragma unroll
                for (int dt = 0; dt < AT_DT; ++dt)
                    o[dt] = mfma16(__builtin_bit_cast(f16x8, vf[ks & 1][dt]), pf, o[dt]);
                if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
#undef AT_LOADV
#undef AT_TR
        }
        if (q < N && !(a.debug & 16)) {
            const float inv = 1.0f / l;
            _Float16* orow = a.out + ((int64_t)b * N + q) * H * D + (int64_t)h * D;
#pragma unroll
            for (int dt = 0; dt < AT_DT; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = 32 * dt + 8 * g + 4 * hh;
                    if (d < D) {
                        uint2 w2;
                        w2.x = Vec<ECOFLAP_F16>::f2h_pk(o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv);
                        w2.y = Vec<ECOFLAP_F16>::f2h_pk(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
                        *(uint2*)(orow + d) = w2;
                    }
                }
        }
#pragma unroll
        for (int c = 0; c < AT_DC; ++c) qf[c] = qn[c];
    }
}

extern "C" int ecoflap_vit_attention(const void* qkv, void* out, int64_t batch, int64_t tokens,
                                     int64_t heads, int64_t head_dim, float scale, int dtype,
                                     void* stream) {
    if (dtype != ECOFLAP_F16) return ECOFLAP_EDTYPE;
    if (batch <= 0 || tokens <= 0 || tokens > AT_ROWS || heads <= 0 || head_dim < 8 ||
        head_dim > 16 * AT_DC || (head_dim % 8) != 0 || batch * heads > 0x7fffffffLL)
        return ECOFLAP_ESIZE;
    if (!qkv || !out) return ECOFLAP_ENULL;
    if (!aligned16(qkv) || (((uintptr_t)out) & 7u)) return ECOFLAP_EALIGN;
    AttnArgs a;
    a.qkv = (const _Float16*)qkv;
    a.out = (_Float16*)out;
    a.B = (int)batch; a.N = (int)tokens; a.H = (int)heads; a.D = (int)head_dim;
    int kp = 2 * (int)head_dim;                    // 16-byte vectors per row must be odd
    if (((kp / 16) & 1) == 0) kp += 16;
    a.kpitch = kp;
    a.scale_log2e = scale * 1.4426950408889634f;
    {
        const char* dbg = getenv("ECOFLAP_ATTN_DEBUG");
        a.debug = dbg ? atoi(dbg) : 0;
    }
    hipLaunchKernelGGL(vit_attention_kernel, dim3((unsigned)(batch * heads)), dim3(256), 0,
                       (hipStream_t)stream, a);
    ECO_CHECK_LAUNCH();
    return 0;
}
