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
#define AT_PF 15                // 16-byte vectors of K and of V a prefetching thread holds

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

// One 32-query tile of one (image, head): S^T = K Q^T, exact softmax, O^T = V^T P^T, store.
static __device__ __forceinline__ void attn_tile(const AttnArgs& a, const unsigned char* Ks,
                                                 const unsigned char* Vs, const _Float16* base,
                                                 int64_t row_stride, int b, int h, int qt, int lane) {
    const int N = a.N, D = a.D, H = a.H, kpitch = a.kpitch;
    const int r = lane & 31, hh = lane >> 5;
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const int q = 32 * qt + r;
    const int qc = q < N ? q : N - 1;
    // Q^T fragments (B operand: k = d on the half-lanes, column = query on the lanes)
    f16x8 qf[AT_DC];
#pragma unroll
    for (int c = 0; c < AT_DC; ++c) {
        const int d0 = 16 * c + 8 * hh;
        qf[c] = d0 < D ? *(const f16x8*)(base + (int64_t)qc * row_stride + d0) : zero8;
    }
    // S^T = K Q^T: tile t holds keys 32t .. 32t+31 (rows, in the registers) x 32 queries (lanes).
    // d-chunk outermost: nine INDEPENDENT accumulators per step (no MFMA waits for the one before
    // it); a K fragment's register is reloaded for the next chunk as soon as its product has
    // issued (one wave per SIMD: nobody else hides the LDS latency).  All AT_NT tiles are computed
    // whatever N is (keys past N are masked below).
    f32x16 s[AT_NT];
#pragma unroll
    for (int t = 0; t < AT_NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) s[t][i] = 0.f;
    if (!(a.debug & 2)) {
        const unsigned char* krow = Ks + r * kpitch + 16 * hh;
        f16x8 kf[AT_NT];
#pragma unroll
        for (int t = 0; t < AT_NT; ++t) kf[t] = *(const f16x8*)(krow + 32 * t * kpitch);
#pragma unroll
        for (int c = 0; c < AT_DC; ++c) {
            const bool live = 16 * c + 8 * hh < D;      // (a row's last chunk may be half padding)
#pragma unroll
            for (int t = 0; t < AT_NT; ++t) {
                s[t] = mfma16(live ? kf[t] : zero8, qf[c], s[t]);
                if (c + 1 < AT_DC) kf[t] = *(const f16x8*)(krow + 32 * t * kpitch + 32 * (c + 1));
            }
            // (keeps the scheduler from hoisting every later chunk's loads up here: they spill)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // exact softmax over the keys of this lane's query: own registers, then lane ^ 32.  Four
    // independent chains for the maximum and for the sum (a serial chain of 144 dependent ops
    // would expose every VALU latency).
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
        // three independent accumulators per step.  V stays row-major in LDS; its transposed A
        // fragments come from ds_read_b64_tr_b16: lane 4q+p of a 16-lane group addresses row q,
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
}

// PERSISTENT workgroups (one per CU: the LDS image allows no second one) walk the (image, head)
// items.  A head has ceil(N / 32) query tiles for 4 waves; in the last round of tiles the waves
// WITHOUT a tile (three of four at N = 257) fetch the NEXT item while the others finish: its K goes
// straight into the second K buffer in LDS (nobody reads that one), its V waits in registers
// until everybody is done with the current V image.  The fill's memory latency hides under the
// last tile instead of standing alone in front of every item.
#define AT_KBYTES 46080         // one K image: N * kpitch (257 * 176 = 45232)
__global__ __launch_bounds__(256) void vit_attention_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char Kb[2 * AT_KBYTES];
    __shared__ __attribute__((aligned(16))) unsigned char Vs[AT_ROWS * AT_VPITCH];
    const int N = a.N, D = a.D, H = a.H, kpitch = a.kpitch;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the branches on it are uniform
    const int64_t row_stride = (int64_t)3 * H * D;
    const int nt = (N + 31) >> 5;
    const int nitems = a.B * H;
    const int vpr = D >> 3, total = N * vpr;
    const int rounds = (nt + 3) >> 2;
    const int busy_last = nt - 4 * (rounds - 1);               // waves with a tile in the last round
    const int nfree = 4 - busy_last;
    const bool overlap = nfree > 0 && (total + 64 * nfree - 1) / (64 * nfree) <= AT_PF;
    const bool dma = kpitch == 2 * D && ((total + 63) >> 6) * 1024 <= AT_KBYTES;
    // Rows past N: V's must be zero (their probabilities are, and 0 * garbage must stay 0); K's may
    // hold anything readable (those keys are masked).  N is the same for every item: once.
    {
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int i = N * (AT_VPITCH / 16) + tid; i < AT_ROWS * (AT_VPITCH / 16); i += 256) ((u32x4*)Vs)[i] = z;
    }
    // all threads: K and V of `item` into LDS (loads of a round in flight together)
#define AT_FILL_ALL(ITEM_, KDST_)                                                                \
    do {                                                                                         \
        const _Float16* nb_ = a.qkv + (int64_t)((ITEM_) / H) * N * row_stride + (int64_t)((ITEM_) % H) * D; \
        for (int v0_ = tid; v0_ < total; v0_ += 256 * 6) {                                       \
            u32x4 kv_[6], vv_[6];                                                                \
            _Pragma("unroll") for (int j = 0; j < 6; ++j) {                                      \
                const int v_ = v0_ + 256 * j < total ? v0_ + 256 * j : total - 1;                \
                const int key_ = v_ / vpr, c_ = v_ - key_ * vpr;                                 \
                const _Float16* src_ = nb_ + (int64_t)key_ * row_stride + 8 * c_;                \
                kv_[j] = *(const u32x4*)(src_ + (int64_t)H * D);                                 \
                vv_[j] = *(const u32x4*)(src_ + (int64_t)2 * H * D);                             \
            }                                                                                    \
            _Pragma("unroll") for (int j = 0; j < 6; ++j) {                                      \
                const int v_ = v0_ + 256 * j;                                                    \
                if (v_ < total) {                                                                \
                    const int key_ = v_ / vpr, c_ = v_ - key_ * vpr;                             \
                    *(u32x4*)((KDST_) + key_ * kpitch + 16 * c_) = kv_[j];                       \
                    *(u32x4*)(Vs + key_ * AT_VPITCH + 16 * c_) = vv_[j];                         \
                }                                                                                \
            }                                                                                    \
        }                                                                                        \
    } while (0)
    if ((int)blockIdx.x < nitems && !(a.debug & 1)) AT_FILL_ALL((int)blockIdx.x, Kb);
    __syncthreads();
    int cur = 0;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int b = item / H, h = item % H;
        const _Float16* base = a.qkv + (int64_t)b * N * row_stride + (int64_t)h * D;
        const unsigned char* Ks = Kb + cur * AT_KBYTES;
        unsigned char* Kn = Kb + (cur ^ 1) * AT_KBYTES;
        const int next = item + (int)gridDim.x;
        const bool has_next = next < nitems && !(a.debug & 1);
        u32x4 pv[AT_PF];
        const int fi = (wave - busy_last) * 64 + lane;         // index among the prefetching threads
        for (int rd = 0; rd + 1 < rounds; ++rd)                // rounds in which every wave has a tile
            attn_tile(a, Ks, Vs, base, row_stride, b, h, 4 * rd + wave, lane);
        {
            const int qt = 4 * (rounds - 1) + wave;
            if (qt < nt) {
                attn_tile(a, Ks, Vs, base, row_stride, b, h, qt, lane);
            } else if (overlap && has_next) {
                const _Float16* nb = a.qkv + (int64_t)(next / H) * N * row_stride + (int64_t)(next % H) * D;
#pragma unroll
                for (int j = 0; j < AT_PF; ++j) {              // V: requested first, kept in registers
                    const int v0 = fi + 64 * nfree * j;
                    const int v = v0 < total ? v0 : total - 1;
                    const int key = v / vpr, c = v - key * vpr;
                    pv[j] = *(const u32x4*)(nb + (int64_t)key * row_stride + 8 * c + (int64_t)2 * H * D);
                }
                if (dma) {
                    // K: DMA straight into the idle buffer, no registers (a K image without row
                    // padding is contiguous: vector v of the head lives at byte 16 v, so a wave's 64
                    // consecutive vectors are one 1-KiB piece at a wave-uniform LDS address)
                    const int fw = wave - busy_last;
                    for (int blk = fw; 64 * blk < total; blk += nfree) {
                        const int v0 = 64 * blk + lane;
                        const int v = v0 < total ? v0 : total - 1;
                        const int key = v / vpr, c = v - key * vpr;
                        const _Float16* src = nb + (int64_t)key * row_stride + 8 * c + (int64_t)H * D;
                        __builtin_amdgcn_global_load_lds(
                            (const __attribute__((address_space(1))) void*)src,
                            (__attribute__((address_space(3))) void*)(Kn + 1024 * blk), 16, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int j0 = 0; j0 < AT_PF; j0 += 5) {    // K: through registers into the idle buffer
                        u32x4 kk[5];
#pragma unroll
                        for (int j = 0; j < 5; ++j) {
                            const int v0 = fi + 64 * nfree * (j0 + j);
                            const int v = v0 < total ? v0 : total - 1;
                            const int key = v / vpr, c = v - key * vpr;
                            kk[j] = *(const u32x4*)(nb + (int64_t)key * row_stride + 8 * c + (int64_t)H * D);
                        }
#pragma unroll
                        for (int j = 0; j < 5; ++j) {
                            const int v = fi + 64 * nfree * (j0 + j);
                            if (v < total) {
                                const int key = v / vpr, c = v - key * vpr;
                                *(u32x4*)(Kn + key * kpitch + 16 * c) = kk[j];
                            }
                        }
                    }
                }
                __builtin_amdgcn_s_waitcnt(0);       // the DMA pieces have landed before the barrier
            }
        }
        __syncthreads();                     // every wave is done with this item's LDS image
        if (has_next) {
            if (overlap) {
                if (wave >= busy_last) {
#pragma unroll
                    for (int j = 0; j < AT_PF; ++j) {
                        const int v = fi + 64 * nfree * j;
                        if (v < total) {
                            const int key = v / vpr, c = v - key * vpr;
                            *(u32x4*)(Vs + key * AT_VPITCH + 16 * c) = pv[j];
                        }
                    }
                }
            } else {
                AT_FILL_ALL(next, Kn);
            }
        }
        __syncthreads();
        cur ^= 1;
    }
#undef AT_FILL_ALL
}

extern "C" int ecoflap_vit_attention(const void* qkv, void* out, int64_t batch, int64_t tokens,
                                     int64_t heads, int64_t head_dim, float scale, int dtype,
                                     void* stream) {
    if (dtype != ECOFLAP_F16) return ECOFLAP_EDTYPE;
    if (batch <= 0 || tokens <= 0 || tokens > AT_ROWS || heads <= 0 || head_dim < 8 ||
        head_dim > 16 * AT_DC || (head_dim % 8) != 0 || batch * heads > 0x7fffffffLL)
        return ECOFLAP_ESIZE;
    int kp = 2 * (int)head_dim;                    // 16-byte vectors per K row in LDS must be odd
    if (((kp / 16) & 1) == 0) kp += 16;
    if (tokens * kp > AT_KBYTES) return ECOFLAP_ESIZE;     // one K image of the double buffer
    if (!qkv || !out) return ECOFLAP_ENULL;
    if (!aligned16(qkv) || (((uintptr_t)out) & 7u)) return ECOFLAP_EALIGN;
    AttnArgs a;
    a.qkv = (const _Float16*)qkv;
    a.out = (_Float16*)out;
    a.B = (int)batch; a.N = (int)tokens; a.H = (int)heads; a.D = (int)head_dim;
    a.kpitch = kp;
    a.scale_log2e = scale * 1.4426950408889634f;
    {
        // phase-ablation switches of tools/attention_launches.py (read once; 0 in production)
        static const int debug = [] { const char* d = getenv("ECOFLAP_ATTN_DEBUG"); return d ? atoi(d) : 0; }();
        a.debug = debug;
    }
    // one persistent workgroup per CU (its LDS image allows no second one) walking the items
    int64_t grid = batch * heads;
    if (grid > 256) grid = 256;
    hipLaunchKernelGGL(vit_attention_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    ECO_CHECK_LAUNCH();
    return 0;
}
