// sparsegpt.hip — SparseGPT block step for gfx950.
//
// Replaces the inner part of SparseGPT.fasterprune
//   LAVIS/lavis/compression/pruners/sparsegpt_pruner.py:172-216
// for one block of <= 128 columns [i1, i2):
//   tmp    = W1**2 / diag(Hinv1)**2                               (:186)
//   thresh = sort(tmp.flatten())[int(tmp.numel() * sparsity)]     (:187)
//   mask1  = tmp <= thresh                                        (:188)
//   for i in range(count):  q = w masked; err = (w - q) / d;
//       W1[:, i:] -= err (x) Hinv1[i, i:]                         (:192-210)
//   W[:, i1:i2] = Q1 ;  Err1 kept for  W[:, i2:] -= Err1 @ Hinv[i1:i2, i2:]   (:212-216)
//
// The Hessian accumulation (:71-82), the two Cholesky factorisations (:118-160) and the
// trailing update GEMM are plain library calls on the host side (torch.addmm / rocSOLVER);
// what is fused here is the part torch runs as ~1000 tiny kernels per block:
//   * threshold: k-th order statistic of tmp over rows x count by three histogram passes
//     (same scheme as the Wanda matrix mode: LDS histograms, integer atomics, every
//     workgroup resolves earlier passes itself);
//   * sweep: rows are independent, so ONE 64-lane wave owns a row: its <=128 block values
//     live in two registers per lane, Hinv1 sits in LDS, each of the `count` sequential steps
//     is a lane broadcast + a fully rounded fp32 divide + one multiply and one subtract per
//     lane (two roundings, as torch's outer-product-then-subtract; no fma contraction).
#include "common.h"

struct SgptSelState {
    uint32_t hist[3][2048];
};

static __device__ __forceinline__ uint32_t sg_block_scan_256(uint32_t v, uint32_t* lds_wave4,
                                                             uint32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    __syncthreads();
    if (lane == 63) lds_wave4[wave] = x;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < wave) base += lds_wave4[k];
    total = lds_wave4[0] + lds_wave4[1] + lds_wave4[2] + lds_wave4[3];
    return x + base;
}

static __device__ __forceinline__ void sg_pick(const uint32_t* __restrict__ hist, int bins,
                                               uint32_t remaining, uint32_t* wave4, uint32_t* out2) {
    uint32_t carry = 0;
    for (int base = 0; base < bins; base += 256) {
        const uint32_t cnt = hist[base + threadIdx.x];
        uint32_t total;
        const uint32_t incl = sg_block_scan_256(cnt, wave4, total) + carry;
        const uint32_t excl = incl - cnt;
        if (excl < remaining && remaining <= incl) {
            out2[0] = (uint32_t)(base + threadIdx.x);
            out2[1] = remaining - excl;
        }
        carry += total;
    }
    __syncthreads();
}

static __device__ __forceinline__ void sg_resolve(const SgptSelState* st, int upto, uint32_t rank0,
                                                  uint32_t* wave4, uint32_t* out2, uint32_t& prefix,
                                                  uint32_t& remaining) {
    prefix = 0;
    remaining = rank0;
    if (upto >= 1) { sg_pick(st->hist[0], 2048, remaining, wave4, out2); prefix |= out2[0] << 21; remaining = out2[1]; __syncthreads(); }
    if (upto >= 2) { sg_pick(st->hist[1], 2048, remaining, wave4, out2); prefix |= out2[0] << 10; remaining = out2[1]; __syncthreads(); }
    if (upto >= 3) { sg_pick(st->hist[2], 1024, remaining, wave4, out2); prefix |= out2[0]; remaining = out2[1]; __syncthreads(); }
}

// tmp = w^2 / d^2 with torch's three roundings
static __device__ __forceinline__ uint32_t sg_metric_bits(float w, float d) {
    const float a = w * w;
    const float b = d * d;
    return __float_as_uint(a / b);
}

template <int PASS>
__global__ __launch_bounds__(256) void sgpt_hist_kernel(const float* __restrict__ W, int64_t rows,
                                                        int64_t ldw, const float* __restrict__ Hinv,
                                                        int64_t ldh, int64_t i1, int count,
                                                        uint32_t rank0, SgptSelState* st) {
    constexpr int SHIFT = PASS == 0 ? 21 : (PASS == 1 ? 10 : 0);
    constexpr int BITS = PASS == 2 ? 10 : 11;
    constexpr uint32_t HI_MASK = PASS == 0 ? 0u : (PASS == 1 ? 0xffe00000u : 0xfffffc00u);
    __shared__ uint32_t h[2048];
    __shared__ float diag[128];
    __shared__ uint32_t wave4[4];
    __shared__ uint32_t out2[2];
    for (int i = threadIdx.x; i < 2048; i += 256) h[i] = 0;
    if (threadIdx.x < count) diag[threadIdx.x] = Hinv[(i1 + threadIdx.x) * ldh + i1 + threadIdx.x];
    uint32_t prefix, remaining;
    sg_resolve(st, PASS, rank0, wave4, out2, prefix, remaining);
    __syncthreads();
    const int64_t n = rows * count;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / count;
        const int c = (int)(e - r * count);
        const uint32_t b = sg_metric_bits(W[r * ldw + i1 + c], diag[c]);
        if ((b & HI_MASK) == prefix) atomicAdd(&h[(b >> SHIFT) & ((1u << BITS) - 1u)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (1 << BITS); i += 256)
        if (h[i]) atomicAdd(&st->hist[PASS][i], h[i]);
}

// one wave per row; workgroup = 4 rows sharing Hinv1 in LDS
__global__ __launch_bounds__(256) void sgpt_sweep_kernel(float* W, int64_t rows, int64_t ldw,
                                                         const float* __restrict__ Hinv, int64_t ldh,
                                                         int64_t i1, int count, uint32_t rank0,
                                                         const SgptSelState* st, int use_thresh,
                                                         const uint8_t* __restrict__ mask_in,
                                                         float* __restrict__ Err, uint8_t* mask_out,
                                                         int nm_n, int nm_m) {
    // upper triangle of Hinv1, packed: row i starts at i*count - i*(i-1)/2, holds j = i..count-1
    extern __shared__ __attribute__((aligned(16))) float Hs[];
    __shared__ uint32_t wave4[4];
    __shared__ uint32_t out2[2];
#define HS(i, j) Hs[(i) * count - (((i) * ((i) - 1)) >> 1) + ((j) - (i))]
    for (int e = threadIdx.x; e < count * count; e += 256) {
        const int i = e / count, j = e - i * count;
        if (j >= i) HS(i, j) = Hinv[(i1 + i) * ldh + i1 + j];
    }
    uint32_t thres_bits = 0, remaining;
    if (use_thresh) sg_resolve(st, 3, rank0, wave4, out2, thres_bits, remaining);
    __syncthreads();
    const float thresh = __uint_as_float(thres_bits);
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* wrow = W + row * ldw + i1;
    float w[2], q[2] = {0.f, 0.f}, er[2] = {0.f, 0.f};
    bool mk[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int j = lane + 64 * t;
        w[t] = (j < count) ? wrow[j] : 0.f;
        if (j < count) {
            if (use_thresh) {
                const float d = HS(j, j);
                mk[t] = __uint_as_float(sg_metric_bits(w[t], d)) <= thresh;     // tmp <= thresh (:188)
            } else {
                mk[t] = mask_in ? mask_in[row * count + j] != 0 : false;    // (n:m: filled in during the sweep)
            }
        } else {
            mk[t] = false;
        }
    }
    for (int i = 0; i < count; ++i) {
        if (nm_n != 0 && i % nm_m == 0) {
            // n:m (:196-198): the n smallest W1[:, i:i+m]**2 / diag**2 of this row, on the sweep's
            // current values, join the mask; ranks by counting over the group's lanes
            const int len = count - i < nm_m ? count - i : nm_m;
            uint32_t myk[2];
            bool in[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int j = lane + 64 * t;
                in[t] = j >= i && j < i + len;
                myk[t] = 0xffffffffu;
                if (in[t]) {
                    const uint32_t b = sg_metric_bits(w[t], HS(j, j));
                    myk[t] = (b & 0x7fffffffu) > 0x7f800000u ? 0xffffffffu : b;      // NaN: the largest
                }
            }
            int rank[2] = {0, 0};
            for (int p = 0; p < len; ++p) {
                const int jj = i + p;
                const uint32_t kp = (jj < 64) ? __shfl(myk[0], jj & 63, 64) : __shfl(myk[1], jj & 63, 64);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int j = lane + 64 * t;
                    if (in[t] && jj != j && (kp < myk[t] || (kp == myk[t] && jj < j))) ++rank[t];
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
                if (in[t] && rank[t] < nm_n) mk[t] = true;
        }
        const int src = i & 63;
        const float wi = (i < 64) ? __shfl(w[0], src, 64) : __shfl(w[1], src, 64);
        const int mi = (i < 64) ? __shfl((int)mk[0], src, 64) : __shfl((int)mk[1], src, 64);
        const float d = HS(i, i);
        const float qi = mi ? 0.f : wi;               // q[mask1[:, i]] = 0            (:200-201)
        const float err = (wi - qi) / d;              // err1 = (w - q) / d            (:206)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int j = lane + 64 * t;
            if (j >= i && j < count) {
                const float p = err * HS(i, j);   // err1 (x) Hinv1[i, i:]     (:207)
                w[t] = w[t] - p;
            }
            if (j == i) { q[t] = qi; er[t] = err; }
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int j = lane + 64 * t;
        if (j < count) {
            wrow[j] = q[t];                               // W[:, i1:i2] = Q1            (:212)
            Err[row * count + j] = er[t];
            if (mask_out) mask_out[row * count + j] = mk[t] ? 1 : 0;
        }
    }
}

extern "C" size_t ecoflap_sparsegpt_workspace_bytes(void) { return sizeof(SgptSelState); }

// W: float[rows, ldw] (fp32 working copy of the weight), Hinv: float[cols, ldh] upper Cholesky
// factor of the inverse Hessian; block columns [i1, i1+count); k = int(rows*count*sparsity).
// mask_in (optional): uint8[rows*count] mask to use instead of the threshold (not used by the
// reference's unstructured path).  err_out: float[rows*count]; mask_out (optional).
extern "C" int ecoflap_sparsegpt_block(float* W, int64_t rows, int64_t ldw, const float* Hinv,
                                       int64_t ldh, int64_t i1, int count, int64_t k,
                                       const uint8_t* mask_in, float* err_out, uint8_t* mask_out,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    if (rows <= 0 || count <= 0 || count > 128 || ldw < i1 + count || ldh < i1 + count || i1 < 0)
        return ECOFLAP_ESIZE;
    const int64_t n = rows * count;
    if (!mask_in && (k < 0 || k >= n || n >= (int64_t)0xffffffffLL)) return ECOFLAP_ESIZE;
    if (!W || !Hinv || !err_out || !workspace) return ECOFLAP_ENULL;
    if (workspace_bytes < sizeof(SgptSelState)) return ECOFLAP_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    SgptSelState* st = (SgptSelState*)workspace;
    const uint32_t rank0 = (uint32_t)(k + 1);
    if (!mask_in) {
        hipError_t e = hipMemsetAsync(st, 0, sizeof(SgptSelState), s);
        if (e != hipSuccess) return (int)e;
        int64_t b = (n + 256 * 8 - 1) / (256 * 8);
        if (b < 1) b = 1;
        if (b > 512) b = 512;
        const dim3 grid((unsigned)b), blk(256);
        hipLaunchKernelGGL((sgpt_hist_kernel<0>), grid, blk, 0, s, W, rows, ldw, Hinv, ldh, i1, count, rank0, st);
        hipLaunchKernelGGL((sgpt_hist_kernel<1>), grid, blk, 0, s, W, rows, ldw, Hinv, ldh, i1, count, rank0, st);
        hipLaunchKernelGGL((sgpt_hist_kernel<2>), grid, blk, 0, s, W, rows, ldw, Hinv, ldh, i1, count, rank0, st);
        ECO_CHECK_LAUNCH();
    }
    const size_t lds = ((size_t)count * (count + 1) / 2) * sizeof(float);   // <= 33 KB
    hipLaunchKernelGGL(sgpt_sweep_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), lds, s, W, rows,
                       ldw, Hinv, ldh, i1, count, rank0, st, mask_in ? 0 : 1, mask_in, err_out,
                       mask_out, 0, 0);
    ECO_CHECK_LAUNCH();
    return 0;
}

// The same block step under n:m (sparsegpt_pruner.py:190, :196-198; dead in the reference's shipped
// configs): no threshold, the mask grows during the sweep.  0 < n <= m <= 16; a group cut short by
// the block's end must hold at least n columns (the reference's topk raises otherwise).
extern "C" int ecoflap_sparsegpt_block_nm(float* W, int64_t rows, int64_t ldw, const float* Hinv,
                                          int64_t ldh, int64_t i1, int count, int n, int m,
                                          float* err_out, uint8_t* mask_out, void* stream) {
    if (rows <= 0 || count <= 0 || count > 128 || ldw < i1 + count || ldh < i1 + count || i1 < 0)
        return ECOFLAP_ESIZE;
    if (n <= 0 || m < n || m > 16 || (count % m != 0 && count % m < n)) return ECOFLAP_ESIZE;
    if (!W || !Hinv || !err_out) return ECOFLAP_ENULL;
    const size_t lds = ((size_t)count * (count + 1) / 2) * sizeof(float);
    hipLaunchKernelGGL(sgpt_sweep_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), lds,
                       (hipStream_t)stream, W, rows, ldw, Hinv, ldh, i1, count, 0u,
                       (const SgptSelState*)nullptr, 0, (const uint8_t*)nullptr, err_out, mask_out, n, m);
    ECO_CHECK_LAUNCH();
    return 0;
}
