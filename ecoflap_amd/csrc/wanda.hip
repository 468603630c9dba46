// wanda.hip — K6 / K7 / K8: Wanda calibration statistic, metric + selection +
// zeroing, and mask apply, for gfx950.
//
// Replaces
//   K6  WrappedGPT.add_batch           LAVIS/lavis/compression/pruners/wanda_pruner.py:71-84
//   K7  |W| * sqrt(scaler_row), sort, select, zero
//         rows mode   (T5/BERT)         wanda_pruner.py:260, :272-279
//         matrix mode (ViT)             wanda_pruner.py:541, :555-558
//   K8  grad *= mask                    UPop/ecoflap_compression_vqa.py:124-129
//
// All HBM-bound byte/element work: 16-byte coalesced loads, selection by radix
// select on the fp32 metric's bit pattern (metrics are >= +0 so the unsigned
// order of the bits IS the float order; NaN sorts last like torch.sort), LDS
// histograms, no GEMM reshaping.
// Algorithmic bytes: K6 tokens*cols*s_x; K7 2*s per element + 4*cols.
#include "common.h"

// =====================================================================================
// K6  column sum of squares -> running mean
// =====================================================================================
// stage 1: partial[chunk][col] = sum over the chunk's rows of x^2 (fp32)
template <int DT, bool VECTOR>
__global__ __launch_bounds__(256) void colsq_partial_kernel(const void* __restrict__ x,
                                                            int64_t tokens, int64_t cols,
                                                            int rows_per_chunk,
                                                            float* __restrict__ partial) {
    constexpr int N = VECTOR ? Vec<DT>::N : 1;
    __shared__ float lds[4][64 * 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t cvec = (int64_t)blockIdx.x * 64 + lane;  // column vector handled by this lane
    const int64_t ncvec = cols / N;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
    int64_t r1 = r0 + rows_per_chunk;
    if (r1 > tokens) r1 = tokens;
    float acc[N];
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = 0.f;
    if (cvec < ncvec) {
        // four independent 16-byte loads in flight per lane before the first use
        int64_t r = r0 + wave;
        for (; r + 12 < r1; r += 16) {
            float f0[N], f1[N], f2[N], f3[N];
            if (VECTOR) {
                const u32x4 a = ld16(x, r * ncvec + cvec), b = ld16(x, (r + 4) * ncvec + cvec);
                const u32x4 c = ld16(x, (r + 8) * ncvec + cvec), d = ld16(x, (r + 12) * ncvec + cvec);
                Vec<DT>::unpack(a, f0); Vec<DT>::unpack(b, f1);
                Vec<DT>::unpack(c, f2); Vec<DT>::unpack(d, f3);
            } else {
                f0[0] = Vec<DT>::load1(x, r * cols + cvec);
                f1[0] = Vec<DT>::load1(x, (r + 4) * cols + cvec);
                f2[0] = Vec<DT>::load1(x, (r + 8) * cols + cvec);
                f3[0] = Vec<DT>::load1(x, (r + 12) * cols + cvec);
            }
#pragma unroll
            for (int i = 0; i < N; ++i) {   // same order as the one-row-at-a-time loop below
                acc[i] += f0[i] * f0[i];
                acc[i] += f1[i] * f1[i];
                acc[i] += f2[i] * f2[i];
                acc[i] += f3[i] * f3[i];
            }
        }
        for (; r < r1; r += 4) {
            float f[N];
            if (VECTOR) {
                Vec<DT>::unpack(ld16(x, r * ncvec + cvec), f);
            } else {
                f[0] = Vec<DT>::load1(x, r * cols + cvec);
            }
#pragma unroll
            for (int i = 0; i < N; ++i) acc[i] += f[i] * f[i];
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) lds[wave][lane * N + i] = acc[i];
    __syncthreads();
    if (wave == 0 && cvec < ncvec) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float s = (lds[0][lane * N + i] + lds[1][lane * N + i]) +
                            (lds[2][lane * N + i] + lds[3][lane * N + i]);
            partial[(int64_t)blockIdx.y * cols + cvec * N + i] = s;
        }
    }
}

// stage 2: fixed-order sum over chunks, then the reference's update (W:80-84)
__global__ __launch_bounds__(256) void colsq_final_kernel(float* __restrict__ scaler_row,
                                                          const float* __restrict__ partial,
                                                          int64_t cols, int nchunks, float decay,
                                                          float n_new) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int k = 0; k < nchunks; ++k) s += partial[(int64_t)k * cols + c];
    const float nrm = __builtin_sqrtf(s);  // torch.norm(...): sqrt of the sum of squares
    const float sq = nrm * nrm;            // ... ** 2
    const float r = scaler_row[c] * decay;  // scaler_row *= n / (n + b)
    scaler_row[c] = r + sq / n_new;         // += ... / nsamples
}

// the same update with the sample count kept on the device (graph-replayable: nothing in the
// launch arguments changes from call to call); decay and n are formed as the host form does
__global__ __launch_bounds__(256) void colsq_final_dev_kernel(float* __restrict__ scaler_row,
                                                              const float* __restrict__ partial,
                                                              int64_t cols, int nchunks,
                                                              const int64_t* __restrict__ n_dev,
                                                              int64_t batch) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const int64_t n0 = n_dev[0];
    const float decay = (float)((double)n0 / (double)(n0 + batch));
    const float n_new = (float)(n0 + batch);
    float s = 0.f;
    for (int k = 0; k < nchunks; ++k) s += partial[(int64_t)k * cols + c];
    const float nrm = __builtin_sqrtf(s);
    const float sq = nrm * nrm;
    const float r = scaler_row[c] * decay;
    scaler_row[c] = r + sq / n_new;
}
__global__ void colsq_bump_kernel(int64_t* n_dev, int64_t batch) { n_dev[0] += batch; }

static inline int colsq_rows_per_chunk(int64_t tokens, int64_t cols) {
    // ~512 workgroups over the 256 CUs, at least 64 rows (16 per wave) per workgroup so every
    // lane keeps several loads in flight and the second stage sums few partials
    const int64_t colblocks = (cols / 4 + 63) / 64 + 1;
    int64_t want_chunks = 512 / colblocks;
    if (want_chunks < 1) want_chunks = 1;
    int64_t rpc = (tokens + want_chunks - 1) / want_chunks;
    if (rpc < 64) rpc = 64;
    return (int)rpc;
}
static inline int colsq_nchunks(int64_t tokens, int64_t cols) {
    const int rpc = colsq_rows_per_chunk(tokens, cols);
    return (int)((tokens + rpc - 1) / rpc);
}

extern "C" size_t ecoflap_colsqnorm_workspace_bytes(int64_t tokens, int64_t cols) {
    if (tokens <= 0 || cols <= 0) return 0;
    return (size_t)colsq_nchunks(tokens, cols) * (size_t)cols * sizeof(float);
}

static int colsq_accum_impl(float* scaler_row, const void* x, int64_t tokens, int64_t cols,
                            int dtype, int64_t nsamples_before, int64_t* nsamples_dev,
                            int64_t batch, void* workspace, size_t workspace_bytes, void* stream);

extern "C" int ecoflap_colsqnorm_accum(float* scaler_row, const void* x, int64_t tokens,
                                       int64_t cols, int dtype, int64_t nsamples_before,
                                       int64_t batch, void* workspace, size_t workspace_bytes,
                                       void* stream) {
    return colsq_accum_impl(scaler_row, x, tokens, cols, dtype, nsamples_before, nullptr, batch,
                            workspace, workspace_bytes, stream);
}

extern "C" int ecoflap_colsqnorm_accum_dev(float* scaler_row, const void* x, int64_t tokens,
                                           int64_t cols, int dtype, int64_t* nsamples_dev,
                                           int64_t batch, void* workspace, size_t workspace_bytes,
                                           void* stream) {
    if (!nsamples_dev) return ECOFLAP_ENULL;
    return colsq_accum_impl(scaler_row, x, tokens, cols, dtype, 0, nsamples_dev, batch, workspace,
                            workspace_bytes, stream);
}

static int colsq_accum_impl(float* scaler_row, const void* x, int64_t tokens, int64_t cols,
                            int dtype, int64_t nsamples_before, int64_t* nsamples_dev,
                            int64_t batch, void* workspace, size_t workspace_bytes, void* stream) {
    if (!dtype_ok(dtype)) return ECOFLAP_EDTYPE;
    if (tokens <= 0 || cols <= 0 || nsamples_before < 0 || batch <= 0) return ECOFLAP_ESIZE;
    if (!scaler_row || !x || !workspace) return ECOFLAP_ENULL;
    if (workspace_bytes < ecoflap_colsqnorm_workspace_bytes(tokens, cols)) return ECOFLAP_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int rpc = colsq_rows_per_chunk(tokens, cols);
    const int nchunks = colsq_nchunks(tokens, cols);
    const int nvec = dtype == ECOFLAP_F32 ? 4 : 8;
    const bool vector = (cols % nvec == 0) && aligned16(x);
    const int64_t ncv = vector ? cols / nvec : cols;
    const dim3 grid((unsigned)((ncv + 63) / 64), (unsigned)nchunks);
    float* partial = (float*)workspace;
#define COLSQ(DT_)                                                                              \
    if (vector)                                                                                 \
        hipLaunchKernelGGL((colsq_partial_kernel<DT_, true>), grid, dim3(256), 0, s, x, tokens, \
                           cols, rpc, partial);                                                 \
    else                                                                                        \
        hipLaunchKernelGGL((colsq_partial_kernel<DT_, false>), grid, dim3(256), 0, s, x, tokens, \
                           cols, rpc, partial);
    if (dtype == ECOFLAP_F32) { COLSQ(ECOFLAP_F32) }
    else if (dtype == ECOFLAP_F16) { COLSQ(ECOFLAP_F16) }
    else { COLSQ(ECOFLAP_BF16) }
#undef COLSQ
    ECO_CHECK_LAUNCH();
    if (nsamples_dev) {
        hipLaunchKernelGGL(colsq_final_dev_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0,
                           s, scaler_row, partial, cols, nchunks, nsamples_dev, batch);
        hipLaunchKernelGGL(colsq_bump_kernel, dim3(1), dim3(1), 0, s, nsamples_dev, batch);
        ECO_CHECK_LAUNCH();
        return 0;
    }
    const float decay = (float)((double)nsamples_before / (double)(nsamples_before + batch));
    const float n_new = (float)(nsamples_before + batch);
    hipLaunchKernelGGL(colsq_final_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, s,
                       scaler_row, partial, cols, nchunks, decay, n_new);
    ECO_CHECK_LAUNCH();
    return 0;
}

// =====================================================================================
// K7 common: sqrt(scaler_row) once per matrix (correctly rounded, = torch.sqrt)
// =====================================================================================
__global__ __launch_bounds__(256) void sqrt_cols_kernel(const float* __restrict__ scaler_row,
                                                        float* __restrict__ sq, int64_t cols) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c < cols) sq[c] = __builtin_sqrtf(scaler_row[c]);
}

template <int DT>
static __device__ __forceinline__ uint32_t metric_bits(const void* w, int64_t i, float sq) {
    return __float_as_uint(__builtin_fabsf(Vec<DT>::load1(w, i)) * sq);
}

// inclusive scan of one value per thread over a 256-thread block; returns (inclusive, total)
static __device__ __forceinline__ uint32_t block_scan_256(uint32_t v, uint32_t* lds_wave4,
                                                          uint32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    __syncthreads();  // protect lds_wave4 reuse
    if (lane == 63) lds_wave4[wave] = x;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < wave) base += lds_wave4[k];
    }
    total = lds_wave4[0] + lds_wave4[1] + lds_wave4[2] + lds_wave4[3];
    return x + base;
}

// =====================================================================================
// K7 rows mode: one 256-thread workgroup per row; the row's metric bits live in LDS.
// =====================================================================================
#define WANDA_ROWS_MAX_COLS 15360  // metric bits + histogram stay under 64 KiB of LDS

template <int DT>
__global__ __launch_bounds__(256) void wanda_rows_kernel(void* w, const float* __restrict__ sq,
                                                         int64_t cols, int64_t k,
                                                         uint8_t* mask_out) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* m = smem;                 // [cols] metric bits
    uint32_t* hist = smem + cols;       // [256]
    uint32_t* wave4 = hist + 256;       // [4]
    uint32_t* sel = wave4 + 4;          // [2] selected digit / remaining rank
    const int64_t row = blockIdx.x;
    void* wrow = (char*)w + row * cols * Vec<DT>::BYTES;
    const int tid = threadIdx.x;

    for (int64_t c = tid; c < cols; c += 256) m[c] = metric_bits<DT>(wrow, c, sq[c]);
    __syncthreads();

    // (k)-th smallest, 1-indexed, by 4 passes of 8-bit MSB-first radix select
    uint32_t prefix = 0, prefix_mask = 0;
    uint32_t remaining = (uint32_t)k;
    if (k > 0 && k < cols) {
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            hist[tid] = 0;
            __syncthreads();
            for (int64_t c = tid; c < cols; c += 256) {
                const uint32_t b = m[c];
                if ((b & prefix_mask) == prefix) atomicAdd(&hist[(b >> shift) & 255u], 1u);
            }
            __syncthreads();
            uint32_t total;
            const uint32_t cnt = hist[tid];
            const uint32_t incl = block_scan_256(cnt, wave4, total);
            const uint32_t excl = incl - cnt;
            if (excl < remaining && remaining <= incl) {  // exactly one thread
                sel[0] = (uint32_t)tid;
                sel[1] = remaining - excl;
            }
            __syncthreads();
            prefix |= sel[0] << shift;
            prefix_mask |= 255u << shift;
            remaining = sel[1];
            __syncthreads();
        }
    }
    // prefix = bits of the k-th smallest metric T; `remaining` of the elements equal to T are
    // pruned, lowest column first (stable sort order, W:272-277).
    if (k <= 0) return;
    const bool all = (k >= cols);
    // count of elements equal to T decides whether column order matters
    uint32_t running = 0;  // equal-to-T elements in columns before this 256-column tile
    for (int64_t c0 = 0; c0 < cols; c0 += 256) {
        const int64_t c = c0 + tid;
        const uint32_t b = (c < cols) ? m[c] : 0xffffffffu;
        const bool eq = !all && (c < cols) && (b == prefix);
        uint32_t total;
        const uint32_t incl = block_scan_256(eq ? 1u : 0u, wave4, total);
        const bool prune = (c < cols) && (all || b < prefix || (eq && (running + incl) <= remaining));
        if (prune) Vec<DT>::store1(wrow, c, 0.0f);
        if (mask_out && c < cols) mask_out[row * cols + c] = prune ? 1 : 0;
        running += total;
    }
}

// -------------------------------------------------------------------------------------
// K7 rows mode, register form (cols a multiple of the 16-byte vector): the row's metric
// bits live in registers (NV vectors per lane), W is read with 16-byte loads and written
// back once; the k-th smallest is found by a bitwise binary search on the bit pattern —
// per bit one compare-count over the lane's registers, a 64-wide shuffle reduction and one
// barrier (LDS only carries the four wave sums).  No atomics, no histogram.
// -------------------------------------------------------------------------------------
static __device__ __forceinline__ uint32_t block_count_256(uint32_t v, uint32_t* lds8, int phase) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    uint32_t* buf = lds8 + 4 * (phase & 1);     // double-buffered: one barrier per call
    if ((threadIdx.x & 63) == 0) buf[threadIdx.x >> 6] = v;
    __syncthreads();
    return buf[0] + buf[1] + buf[2] + buf[3];
}

template <int DT, int NV>
__global__ __launch_bounds__(256) void wanda_rows_reg_kernel(void* w, const float* __restrict__ sq,
                                                             int64_t cols, int64_t k,
                                                             uint8_t* mask_out) {
    constexpr int N = Vec<DT>::N;
    __shared__ uint32_t lds8[8];
    __shared__ uint32_t wave4[4];
    const int tid = threadIdx.x;
    const int64_t row = blockIdx.x;
    const int64_t nvec = cols / N;
    void* wrow = (char*)w + row * cols * Vec<DT>::BYTES;
    u32x4 wv[NV];
    uint32_t m[NV][N];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int64_t v = tid + 256 * j;
        if (v < nvec) {
            wv[j] = ld16(wrow, v);
            float f[N];
            Vec<DT>::unpack(wv[j], f);
#pragma unroll
            for (int q = 0; q < N / 4; ++q) {
                const u32x4 s4 = ld16(sq, v * (N / 4) + q);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    m[j][4 * q + i] = __float_as_uint(__builtin_fabsf(f[4 * q + i]) * __uint_as_float(s4[i]));
            }
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i) m[j][i] = 0xffffffffu;   // padding: never among the k smallest
        }
    }
    const bool all = (k >= cols);
    uint32_t T = 0xffffffffu, take_equal = 0, total_equal = 0;
    if (!all) {
        // k-th smallest (1-indexed): largest T with #(m < T) < k
        uint32_t prefix = 0;
        int phase = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t cand = prefix | (1u << bit);
            uint32_t c = 0;
#pragma unroll
            for (int j = 0; j < NV; ++j)
#pragma unroll
                for (int i = 0; i < N; ++i) c += (m[j][i] < cand) ? 1u : 0u;
            c = block_count_256(c, lds8, phase++);
            if (c < (uint32_t)k) prefix = cand;
        }
        T = prefix;
        uint32_t less = 0, eq = 0;
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int i = 0; i < N; ++i) {
                less += (m[j][i] < T) ? 1u : 0u;
                eq += (m[j][i] == T) ? 1u : 0u;
            }
        less = block_count_256(less, lds8, phase++);
        total_equal = block_count_256(eq, lds8, phase++);
        take_equal = (uint32_t)k - less;          // >= 1 of the elements equal to T are pruned
    }
    const bool ordered = !all && (take_equal < total_equal);   // ties cut by column order
    uint32_t running = 0;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int64_t v = tid + 256 * j;
        uint32_t rank0 = 0;
        if (ordered) {     // block-uniform: equal-to-T elements in lower columns come first
            uint32_t mine = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) mine += (m[j][i] == T) ? 1u : 0u;
            uint32_t total;
            const uint32_t incl = block_scan_256(mine, wave4, total);
            rank0 = running + incl - mine;
            running += total;
        }
        if (v < nvec) {
            float f[N];
            Vec<DT>::unpack(wv[j], f);
            uint32_t bytes_lo = 0, bytes_hi = 0;
            uint32_t seen = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const bool eqT = (m[j][i] == T);
                bool prune = all || (m[j][i] < T) || (eqT && (!ordered || (rank0 + seen) < take_equal));
                seen += eqT ? 1u : 0u;
                if (prune) f[i] = 0.0f;
                if (i < 4) bytes_lo |= (prune ? 1u : 0u) << (8 * i);
                else bytes_hi |= (prune ? 1u : 0u) << (8 * (i - 4));
            }
            st16(wrow, v, Vec<DT>::pack(f));
            if (mask_out) {
                uint8_t* mrow = mask_out + row * cols + v * N;
                *(uint32_t*)mrow = bytes_lo;
                if (N == 8) *(uint32_t*)(mrow + 4) = bytes_hi;
            }
        }
    }
}

// -------------------------------------------------------------------------------------
// K7 rows mode, wave form (rows of up to 64*12 vectors): ONE 64-lane wave per row, four rows
// per workgroup.  Counts are ballots: v_cmp writes the lane mask, s_bcnt1 counts it on the
// scalar unit — the bitwise search needs no shuffle, no LDS and no barrier at all.
// -------------------------------------------------------------------------------------
template <int DT, int NV>
__global__ __launch_bounds__(256) void wanda_rows_wave_kernel(void* w, const float* __restrict__ sq,
                                                              int64_t rows, int64_t cols, int64_t k,
                                                              uint8_t* mask_out) {
    constexpr int N = Vec<DT>::N;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;                   // whole wave leaves together
    const int64_t nvec = cols / N;
    void* wrow = (char*)w + row * cols * Vec<DT>::BYTES;
    u32x4 wv[NV];
    uint32_t m[NV][N];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int64_t v = lane + 64 * j;
        if (v < nvec) {
            wv[j] = ld16(wrow, v);
            float f[N];
            Vec<DT>::unpack(wv[j], f);
#pragma unroll
            for (int q = 0; q < N / 4; ++q) {
                const u32x4 s4 = ld16(sq, v * (N / 4) + q);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    m[j][4 * q + i] = __float_as_uint(__builtin_fabsf(f[4 * q + i]) * __uint_as_float(s4[i]));
            }
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i) m[j][i] = 0xffffffffu;
        }
    }
    const bool all = (k >= cols);
    uint32_t T = 0xffffffffu, take_equal = 0, total_equal = 0;
    if (!all) {
        uint32_t prefix = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t cand = prefix | (1u << bit);
            uint32_t c = 0;
#pragma unroll
            for (int j = 0; j < NV; ++j)
#pragma unroll
                for (int i = 0; i < N; ++i) c += (uint32_t)__popcll(__ballot(m[j][i] < cand));
            if (c < (uint32_t)k) prefix = cand;
        }
        T = prefix;
        uint32_t less = 0;
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int i = 0; i < N; ++i) {
                less += (uint32_t)__popcll(__ballot(m[j][i] < T));
                total_equal += (uint32_t)__popcll(__ballot(m[j][i] == T));
            }
        take_equal = (uint32_t)k - less;
    }
    const bool ordered = !all && (take_equal < total_equal);
    uint32_t running = 0;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int64_t v = lane + 64 * j;
        uint32_t rank0 = 0;
        if (ordered) {      // wave-uniform
            uint32_t mine = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) mine += (m[j][i] == T) ? 1u : 0u;
            uint32_t x = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t y = __shfl_up(x, off, 64);
                if (lane >= off) x += y;
            }
            rank0 = running + x - mine;
            running += __shfl(x, 63, 64);
        }
        if (v < nvec) {
            float f[N];
            Vec<DT>::unpack(wv[j], f);
            uint32_t bytes_lo = 0, bytes_hi = 0, seen = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const bool eqT = (m[j][i] == T);
                const bool prune = all || (m[j][i] < T) ||
                                   (eqT && (!ordered || (rank0 + seen) < take_equal));
                seen += eqT ? 1u : 0u;
                if (prune) f[i] = 0.0f;
                if (i < 4) bytes_lo |= (prune ? 1u : 0u) << (8 * i);
                else bytes_hi |= (prune ? 1u : 0u) << (8 * (i - 4));
            }
            st16(wrow, v, Vec<DT>::pack(f));
            if (mask_out) {
                uint8_t* mrow = mask_out + row * cols + v * N;
                *(uint32_t*)mrow = bytes_lo;
                if (N == 8) *(uint32_t*)(mrow + 4) = bytes_hi;
            }
        }
    }
}

template <int DT>
static int launch_rows_wave(void* w, const float* sq, int64_t rows, int64_t cols, int64_t k,
                            uint8_t* mask_out, hipStream_t s) {
    constexpr int N = Vec<DT>::N;
    const int64_t nvec = cols / N;
    const int nv = (int)((nvec + 63) / 64);
    const dim3 grid((unsigned)((rows + 3) / 4)), blk(256);
#define ROWS_WAVE(NV_)                                                                   \
    hipLaunchKernelGGL((wanda_rows_wave_kernel<DT, NV_>), grid, blk, 0, s, w, sq, rows, cols, k, \
                       mask_out)
    if (nv <= 1) ROWS_WAVE(1);
    else if (nv <= 2) ROWS_WAVE(2);
    else if (nv <= 4) ROWS_WAVE(4);
    else if (nv <= 6) ROWS_WAVE(6);
    else if (nv <= 8) ROWS_WAVE(8);
    else if (nv <= 10) ROWS_WAVE(10);
    else if (nv <= 12) ROWS_WAVE(12);
    else return 1;
#undef ROWS_WAVE
    return 0;
}

template <int DT>
static int launch_rows_reg(void* w, const float* sq, int64_t rows, int64_t cols, int64_t k,
                           uint8_t* mask_out, hipStream_t s) {
    constexpr int N = Vec<DT>::N;
    const int64_t nvec = cols / N;
    const int nv = (int)((nvec + 255) / 256);
    const dim3 grid((unsigned)rows), blk(256);
#define ROWS_REG(NV_)                                                                          \
    hipLaunchKernelGGL((wanda_rows_reg_kernel<DT, NV_>), grid, blk, 0, s, w, sq, cols, k, mask_out)
    if (nv <= 1) ROWS_REG(1);
    else if (nv <= 2) ROWS_REG(2);
    else if (nv <= 3) ROWS_REG(3);
    else if (nv <= 4) ROWS_REG(4);
    else if (nv <= 6) ROWS_REG(6);
    else if (nv <= 8) ROWS_REG(8);
    else if (nv <= 12) ROWS_REG(12);
    else if (nv <= 16) ROWS_REG(16);
    else return 1;
#undef ROWS_REG
    return 0;
}

// =====================================================================================
// K7 matrix mode: global (k+1)-th smallest over rows*cols by 3 histogram passes
// (11 + 11 + 10 bits), then zero metric <= threshold.
// =====================================================================================
struct MatrixSelState {
    uint32_t hist[3][2048];
};

// Every workgroup resolves the previous passes itself from the global histograms (8 KB each,
// L2-resident): bin = first bin whose inclusive count reaches `remaining`.  Identical in every
// workgroup, so no separate "pick" launch and no inter-workgroup hand-off is needed.
static __device__ __forceinline__ void pick_bin(const uint32_t* __restrict__ hist, int bins,
                                                uint32_t remaining, uint32_t* wave4,
                                                uint32_t* out2 /* LDS: bin, new remaining */) {
    uint32_t carry = 0;
    for (int base = 0; base < bins; base += 256) {
        const uint32_t cnt = hist[base + threadIdx.x];
        uint32_t total;
        const uint32_t incl = block_scan_256(cnt, wave4, total) + carry;
        const uint32_t excl = incl - cnt;
        if (excl < remaining && remaining <= incl) {
            out2[0] = (uint32_t)(base + threadIdx.x);
            out2[1] = remaining - excl;
        }
        carry += total;
    }
    __syncthreads();
}

// prefix (selected high bits) and remaining rank after `upto` resolved passes
static __device__ __forceinline__ void resolve(const MatrixSelState* st, int upto, uint32_t rank0,
                                               uint32_t* wave4, uint32_t* out2, uint32_t& prefix,
                                               uint32_t& remaining) {
    prefix = 0;
    remaining = rank0;
    if (upto >= 1) {
        pick_bin(st->hist[0], 2048, remaining, wave4, out2);
        prefix |= out2[0] << 21; remaining = out2[1];
        __syncthreads();
    }
    if (upto >= 2) {
        pick_bin(st->hist[1], 2048, remaining, wave4, out2);
        prefix |= out2[0] << 10; remaining = out2[1];
        __syncthreads();
    }
    if (upto >= 3) {
        pick_bin(st->hist[2], 1024, remaining, wave4, out2);
        prefix |= out2[0]; remaining = out2[1];
        __syncthreads();
    }
}

template <int DT, int PASS, bool VECTOR>
__global__ __launch_bounds__(256) void wanda_matrix_hist_kernel(const void* __restrict__ w,
                                                                const float* __restrict__ sq,
                                                                int64_t rows, int64_t cols,
                                                                uint32_t rank0, MatrixSelState* st) {
    constexpr int SHIFT = PASS == 0 ? 21 : (PASS == 1 ? 10 : 0);
    constexpr int BITS = PASS == 2 ? 10 : 11;
    constexpr uint32_t HI_MASK = PASS == 0 ? 0u : (PASS == 1 ? 0xffe00000u : 0xfffffc00u);
    constexpr int N = Vec<DT>::N;
    __shared__ uint32_t h[2048];
    __shared__ uint32_t wave4[4];
    __shared__ uint32_t out2[2];
    for (int i = threadIdx.x; i < 2048; i += 256) h[i] = 0;
    uint32_t prefix, remaining;
    resolve(st, PASS, rank0, wave4, out2, prefix, remaining);
    __syncthreads();
    if (VECTOR) {
        const int64_t vpr = cols / N;                 // vectors per row
        const int64_t nvec = rows * vpr;
        for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
            const int64_t c0 = (v % vpr) * N;
            float f[N];
            Vec<DT>::unpack(ld16(w, v), f);
#pragma unroll
            for (int q = 0; q < N / 4; ++q) {
                const u32x4 s4 = ld16(sq, c0 / 4 + q);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t b = __float_as_uint(__builtin_fabsf(f[4 * q + i]) * __uint_as_float(s4[i]));
                    if ((b & HI_MASK) == prefix) atomicAdd(&h[(b >> SHIFT) & ((1u << BITS) - 1u)], 1u);
                }
            }
        }
    } else {
        for (int64_t r = blockIdx.x; r < rows; r += gridDim.x)
            for (int64_t c = threadIdx.x; c < cols; c += 256) {
                const uint32_t b = metric_bits<DT>(w, r * cols + c, sq[c]);
                if ((b & HI_MASK) == prefix) atomicAdd(&h[(b >> SHIFT) & ((1u << BITS) - 1u)], 1u);
            }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (1 << BITS); i += 256)
        if (h[i]) atomicAdd(&st->hist[PASS][i], h[i]);
}

template <int DT, bool VECTOR>
__global__ __launch_bounds__(256) void wanda_matrix_apply_kernel(void* w,
                                                                 const float* __restrict__ sq,
                                                                 int64_t rows, int64_t cols,
                                                                 uint32_t rank0,
                                                                 const MatrixSelState* st,
                                                                 uint8_t* mask_out) {
    constexpr int N = Vec<DT>::N;
    __shared__ uint32_t wave4[4];
    __shared__ uint32_t out2[2];
    uint32_t thres_bits, remaining;
    resolve(st, 3, rank0, wave4, out2, thres_bits, remaining);
    const float thres = __uint_as_float(thres_bits);
    if (VECTOR) {
        const int64_t vpr = cols / N;
        const int64_t nvec = rows * vpr;
        for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
            const int64_t c0 = (v % vpr) * N;
            float f[N];
            Vec<DT>::unpack(ld16(w, v), f);
            uint32_t lo = 0, hi = 0;
            bool any = false;
#pragma unroll
            for (int q = 0; q < N / 4; ++q) {
                const u32x4 s4 = ld16(sq, c0 / 4 + q);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = 4 * q + i;
                    // W_metric <= thres (W:556): false for NaN metrics, as in torch
                    const bool prune = (__builtin_fabsf(f[e]) * __uint_as_float(s4[i])) <= thres;
                    if (prune) { f[e] = 0.0f; any = true; }
                    if (e < 4) lo |= (prune ? 1u : 0u) << (8 * e);
                    else hi |= (prune ? 1u : 0u) << (8 * (e - 4));
                }
            }
            if (any) st16(w, v, Vec<DT>::pack(f));
            if (mask_out) {
                uint8_t* m = mask_out + v * N;
                *(uint32_t*)m = lo;
                if (N == 8) *(uint32_t*)(m + 4) = hi;
            }
        }
    } else {
        for (int64_t r = blockIdx.x; r < rows; r += gridDim.x)
            for (int64_t c = threadIdx.x; c < cols; c += 256) {
                const int64_t i = r * cols + c;
                const float mval = __uint_as_float(metric_bits<DT>(w, i, sq[c]));
                const bool prune = mval <= thres;
                if (prune) Vec<DT>::store1(w, i, 0.0f);
                if (mask_out) mask_out[i] = prune ? 1 : 0;
            }
    }
}

extern "C" size_t ecoflap_wanda_workspace_bytes(int64_t rows, int64_t cols) {
    (void)rows;
    if (cols <= 0) return 0;
    // sqrt table (padded to 256 B) + matrix-mode selection state
    return (((size_t)cols * sizeof(float) + 255) / 256) * 256 + sizeof(MatrixSelState);
}

static inline size_t sq_bytes(int64_t cols) { return (((size_t)cols * sizeof(float) + 255) / 256) * 256; }

extern "C" int ecoflap_wanda_prune_rows(void* w, const float* scaler_row, int64_t rows,
                                        int64_t cols, int dtype, int64_t k, uint8_t* mask_out,
                                        void* workspace, size_t workspace_bytes, void* stream) {
    if (!dtype_ok(dtype)) return ECOFLAP_EDTYPE;
    if (rows <= 0 || cols <= 0 || k < 0 || cols > WANDA_ROWS_MAX_COLS) return ECOFLAP_ESIZE;
    if (!w || !scaler_row || !workspace) return ECOFLAP_ENULL;
    if (workspace_bytes < ecoflap_wanda_workspace_bytes(rows, cols)) return ECOFLAP_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* sq = (float*)workspace;
    hipLaunchKernelGGL(sqrt_cols_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, s,
                       scaler_row, sq, cols);
    ECO_CHECK_LAUNCH();
    const int nvec_elems = dtype == ECOFLAP_F32 ? 4 : 8;
    // mask rows are written with 4/8-byte stores: cols % 8 keeps them aligned for 16-bit dtypes
    if (cols % nvec_elems == 0 && aligned16(w) && aligned16(sq) &&
        (!mask_out || (((uintptr_t)mask_out) & 7u) == 0)) {
        // measured on MI355X: the wave form wins up to 256 vectors per row (2048 bf16 columns),
        // the workgroup form beyond
        int miss = 1;
        if (cols / nvec_elems <= 256) {
            if (dtype == ECOFLAP_F32) miss = launch_rows_wave<ECOFLAP_F32>(w, sq, rows, cols, k, mask_out, s);
            else if (dtype == ECOFLAP_F16) miss = launch_rows_wave<ECOFLAP_F16>(w, sq, rows, cols, k, mask_out, s);
            else miss = launch_rows_wave<ECOFLAP_BF16>(w, sq, rows, cols, k, mask_out, s);
        }
        if (!miss) {
            ECO_CHECK_LAUNCH();
            return 0;
        }
        if (dtype == ECOFLAP_F32) miss = launch_rows_reg<ECOFLAP_F32>(w, sq, rows, cols, k, mask_out, s);
        else if (dtype == ECOFLAP_F16) miss = launch_rows_reg<ECOFLAP_F16>(w, sq, rows, cols, k, mask_out, s);
        else miss = launch_rows_reg<ECOFLAP_BF16>(w, sq, rows, cols, k, mask_out, s);
        if (!miss) {
            ECO_CHECK_LAUNCH();
            return 0;
        }
    }
    const size_t lds = ((size_t)cols + 256 + 4 + 4) * sizeof(uint32_t);
    if (dtype == ECOFLAP_F32)
        hipLaunchKernelGGL((wanda_rows_kernel<ECOFLAP_F32>), dim3((unsigned)rows), dim3(256), lds, s,
                           w, sq, cols, k, mask_out);
    else if (dtype == ECOFLAP_F16)
        hipLaunchKernelGGL((wanda_rows_kernel<ECOFLAP_F16>), dim3((unsigned)rows), dim3(256), lds, s,
                           w, sq, cols, k, mask_out);
    else
        hipLaunchKernelGGL((wanda_rows_kernel<ECOFLAP_BF16>), dim3((unsigned)rows), dim3(256), lds,
                           s, w, sq, cols, k, mask_out);
    ECO_CHECK_LAUNCH();
    return 0;
}

template <int DT>
static int wanda_matrix_launch(void* w, const float* sq, int64_t n, int64_t cols, int64_t k,
                               MatrixSelState* st, uint8_t* mask_out, hipStream_t s) {
    const int64_t rows = n / cols;
    const bool vector = (cols % Vec<DT>::N == 0) && aligned16(w) && aligned16(sq) &&
                        (!mask_out || (((uintptr_t)mask_out) & 7u) == 0);
    int64_t b = vector ? (n / Vec<DT>::N + 255) / 256 : rows;
    if (b < 1) b = 1;
    if (b > 1024) b = 1024;
    const dim3 grid((unsigned)b), blk(256);
    const uint32_t rank0 = (uint32_t)(k + 1);      // sorted[k], 0-indexed -> (k+1)-th smallest
#define MATRIX_PASSES(V)                                                                        \
    hipLaunchKernelGGL((wanda_matrix_hist_kernel<DT, 0, V>), grid, blk, 0, s, w, sq, rows, cols, rank0, st); \
    hipLaunchKernelGGL((wanda_matrix_hist_kernel<DT, 1, V>), grid, blk, 0, s, w, sq, rows, cols, rank0, st); \
    hipLaunchKernelGGL((wanda_matrix_hist_kernel<DT, 2, V>), grid, blk, 0, s, w, sq, rows, cols, rank0, st); \
    hipLaunchKernelGGL((wanda_matrix_apply_kernel<DT, V>), grid, blk, 0, s, w, sq, rows, cols, rank0, st, mask_out)
    if (vector) { MATRIX_PASSES(true); } else { MATRIX_PASSES(false); }
#undef MATRIX_PASSES
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

extern "C" int ecoflap_wanda_prune_matrix(void* w, const float* scaler_row, int64_t rows,
                                          int64_t cols, int dtype, int64_t k, uint8_t* mask_out,
                                          void* workspace, size_t workspace_bytes, void* stream) {
    if (!dtype_ok(dtype)) return ECOFLAP_EDTYPE;
    const int64_t n = rows * cols;
    // the reference indexes sorted[k]: k == numel raises IndexError there (W:555)
    if (rows <= 0 || cols <= 0 || k < 0 || k >= n || n >= (int64_t)0xffffffffLL) return ECOFLAP_ESIZE;
    if (!w || !scaler_row || !workspace) return ECOFLAP_ENULL;
    if (workspace_bytes < ecoflap_wanda_workspace_bytes(rows, cols)) return ECOFLAP_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* sq = (float*)workspace;
    MatrixSelState* st = (MatrixSelState*)((char*)workspace + sq_bytes(cols));
    hipError_t e = hipMemsetAsync(st, 0, sizeof(MatrixSelState), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(sqrt_cols_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, s,
                       scaler_row, sq, cols);
    ECO_CHECK_LAUNCH();
    if (dtype == ECOFLAP_F32) return wanda_matrix_launch<ECOFLAP_F32>(w, sq, n, cols, k, st, mask_out, s);
    if (dtype == ECOFLAP_F16) return wanda_matrix_launch<ECOFLAP_F16>(w, sq, n, cols, k, st, mask_out, s);
    return wanda_matrix_launch<ECOFLAP_BF16>(w, sq, n, cols, k, st, mask_out, s);
}

// =====================================================================================
// K8  grad *= mask
// =====================================================================================
template <int DT>
__global__ __launch_bounds__(256) void mask_mul_kernel(void* g, const uint8_t* __restrict__ keep,
                                                       int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float m = keep[i] ? 1.0f : 0.0f;
        Vec<DT>::store1(g, i, Vec<DT>::load1(g, i) * m);
    }
}

extern "C" int ecoflap_mask_mul(void* g, const uint8_t* keep_mask, int64_t n, int dtype,
                                void* stream) {
    if (!dtype_ok(dtype)) return ECOFLAP_EDTYPE;
    if (n < 0) return ECOFLAP_ESIZE;
    if (n == 0) return 0;
    if (!g || !keep_mask) return ECOFLAP_ENULL;
    int64_t b = (n + 256 * 4 - 1) / (256 * 4);
    if (b > 2048) b = 2048;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == ECOFLAP_F32)
        hipLaunchKernelGGL((mask_mul_kernel<ECOFLAP_F32>), dim3((unsigned)b), dim3(256), 0, s, g, keep_mask, n);
    else if (dtype == ECOFLAP_F16)
        hipLaunchKernelGGL((mask_mul_kernel<ECOFLAP_F16>), dim3((unsigned)b), dim3(256), 0, s, g, keep_mask, n);
    else
        hipLaunchKernelGGL((mask_mul_kernel<ECOFLAP_BF16>), dim3((unsigned)b), dim3(256), 0, s, g, keep_mask, n);
    ECO_CHECK_LAUNCH();
    return 0;
}
