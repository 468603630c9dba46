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
// Phase clock for the K6 / K7 matrix-mode kernels, compiled in only by tools/diag/k7_clock.sh
// (-DECO_K7_CLOCK, its own .so): workgroup 0 stamps the 100 MHz wall clock at its phase boundaries,
// every workgroup folds its entry / exit into a min / max.
#ifdef ECO_K7_CLOCK
__device__ unsigned long long eco_k7_clk[8][16];
#define K7_STAMP(k, i) do { if (threadIdx.x == 0 && blockIdx.x == 0) eco_k7_clk[k][i] = wall_clock64(); } while (0)
#define K7_ENTER(k) do { if (threadIdx.x == 0) atomicMin(&eco_k7_clk[k][14], wall_clock64()); } while (0)
#define K7_EXIT(k) do { if (threadIdx.x == 0) atomicMax(&eco_k7_clk[k][15], wall_clock64()); } while (0)
extern "C" int ecoflap_debug_k7_clock_reset(void) {
    unsigned long long z[8][16];
    for (int k = 0; k < 8; ++k) for (int i = 0; i < 16; ++i) z[k][i] = i == 14 ? ~0ull : 0ull;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(eco_k7_clk), z, sizeof(z));
}
extern "C" int ecoflap_debug_k7_clock_read(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(eco_k7_clk), sizeof(unsigned long long) * 8 * 16);
}
#else
#define K7_STAMP(k, i) do { } while (0)
#define K7_ENTER(k) do { } while (0)
#define K7_EXIT(k) do { } while (0)
#endif

// =====================================================================================
// ONE launch per hooked input.  Workgroup (column block, row chunk) writes
// partial[chunk][col] = sum over the chunk's rows of x^2 (fp32); the LAST chunk of a column block
// to finish (ticket counter in the workspace) sums that block's partials in fixed chunk order —
// so the result does not depend on which workgroup came last — and applies the reference's
// update (W:80-84).  Tickets reset themselves: the workspace is zeroed once by its owner.
// FROM_DEV: the sample count lives in device memory (graph-replayable: no launch argument
// changes from call to call); the last workgroup of the whole grid then adds `batch` to it.
struct ColsqArgs {
    const void* x;
    int64_t tokens, cols;
    int rows_per_chunk, nchunks;
    float* partial;          // [nchunks][cols]
    unsigned* tickets;       // [colblocks + 1], zero between launches
    float* scaler_row;
    float decay, n_new;      // host form
    int64_t* n_dev;          // device form (nullptr: host form)
    int64_t batch;
    int colblocks;           // column blocks of this input (= gridDim.x of the one-input launch)
    int first_wg;            // multi-input launch: first workgroup of this input
    int vector;              // 16-byte column vectors usable
    int raw;                 // 1: scaler_row[c] = ||x_c||^2 of THIS input only (no running mean)
    int small;               // 1: few rows — one workgroup per 64 columns reduces ALL rows (colsq_small_body)
};

// the work of workgroup (bx = column block, by = row chunk) of one hooked input
template <int DT, bool VECTOR>
__device__ __forceinline__ void colsq_body(const ColsqArgs& a, const int bx, const int by,
                                           float (*lds)[64 * 8], unsigned* last) {
    constexpr int N = VECTOR ? Vec<DT>::N : 1;
    const void* __restrict__ x = a.x;
    const int64_t tokens = a.tokens, cols = a.cols;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t cvec = (int64_t)bx * 64 + lane;  // column vector handled by this lane
    const int64_t ncvec = cols / N;
    const int64_t r0 = (int64_t)by * a.rows_per_chunk;
    int64_t r1 = r0 + a.rows_per_chunk;
    if (r1 > tokens) r1 = tokens;
    float acc[N];
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = 0.f;
    K7_ENTER(5); K7_STAMP(5, 0);
    if (cvec < ncvec) {
        // four independent 16-byte loads in flight per lane before the first use
        int64_t r = r0 + wave;
        for (; r + 12 < r1; r += 16) {
            float f0[N], f1[N], f2[N], f3[N];
            if (VECTOR) {
                const u32x4 va = ld16(x, r * ncvec + cvec), vb = ld16(x, (r + 4) * ncvec + cvec);
                const u32x4 vc = ld16(x, (r + 8) * ncvec + cvec), vd = ld16(x, (r + 12) * ncvec + cvec);
                Vec<DT>::unpack(va, f0); Vec<DT>::unpack(vb, f1);
                Vec<DT>::unpack(vc, f2); Vec<DT>::unpack(vd, f3);
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
    K7_STAMP(5, 1);
#pragma unroll
    for (int i = 0; i < N; ++i) lds[wave][lane * N + i] = acc[i];
    __syncthreads();
    if (wave == 0 && cvec < ncvec) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float s = (lds[0][lane * N + i] + lds[1][lane * N + i]) +
                            (lds[2][lane * N + i] + lds[3][lane * N + i]);
            // device-scope (sc1) store: written through this XCD's L2, so the workgroup that
            // finishes the column block — possibly on another XCD — reads it without anyone
            // having to write back or invalidate a whole L2 (what a release fence would do)
            __hip_atomic_store(&a.partial[(int64_t)by * cols + cvec * N + i], s,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ---- last chunk of this column block finishes the update ------------------------------
    __builtin_amdgcn_s_waitcnt(0);         // this wave's partial stores have been performed
    __syncthreads();
    K7_STAMP(5, 2);
    if (threadIdx.x == 0)
        *last = (__hip_atomic_fetch_add(&a.tickets[bx], 1u, __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_AGENT) == (unsigned)a.nchunks - 1u);
    __syncthreads();
    K7_STAMP(5, 3);
    K7_EXIT(5);
    if (!*last) return;
#ifdef ECO_K7_CLOCK
    if (threadIdx.x == 0) { atomicMin(&eco_k7_clk[6][14], wall_clock64()); }
#endif
    float decay = a.decay, n_new = a.n_new;
    if (a.n_dev) {                         // formed as the host form does (double, then float)
        const int64_t n0 = a.n_dev[0];
        decay = (float)((double)n0 / (double)(n0 + a.batch));
        n_new = (float)(n0 + a.batch);
    }
    const int64_t c0 = (int64_t)bx * 64 * N;
    for (int64_t c = c0 + threadIdx.x; c < c0 + 64 * N && c < cols; c += 256) {
        float s = 0.f;
        int k = 0;
        for (; k + 8 <= a.nchunks; k += 8) {     // eight loads in flight, summed in chunk order
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)          // device-scope loads: never a stale L2 line
                t[j] = __hip_atomic_load(&a.partial[(int64_t)(k + j) * cols + c], __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += t[j];
        }
        for (; k < a.nchunks; ++k)
            s += __hip_atomic_load(&a.partial[(int64_t)k * cols + c], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        const float nrm = __builtin_sqrtf(s);  // torch.norm(...): sqrt of the sum of squares
        const float sq = nrm * nrm;            // ... ** 2
        if (a.raw) {
            a.scaler_row[c] = sq;              // this input's own statistic (data-parallel replay)
        } else {
            const float r = a.scaler_row[c] * decay;  // scaler_row *= n / (n + b)
            a.scaler_row[c] = r + sq / n_new;         // += ... / nsamples
        }
    }
    // every wave of this workgroup has read n_dev (and used it) before thread 0 takes the
    // grid-level ticket: the workgroup that takes the LAST one bumps the count
    __syncthreads();
    K7_EXIT(6);
    if (threadIdx.x == 0) {
        __hip_atomic_store(&a.tickets[bx], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.n_dev) {                     // every column block has read n_dev before its ticket
            if (__hip_atomic_fetch_add(&a.tickets[a.colblocks], 1u, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT) == (unsigned)a.colblocks - 1u) {
                __hip_atomic_store(&a.tickets[a.colblocks], 0u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                a.n_dev[0] += a.batch;
            }
        }
    }
}

// Few rows (tokens <= COLSQ_SMALL_TOKENS: one calibration batch of a FlanT5 block is 128-384
// rows of 2048-5120 columns, 0.5-4 MB): the two-stage scheme above is all latency there — partial
// sums to memory, a ticket, the last arriver's reload (12.5 us for a block's 13 MB).  Here a
// workgroup owns 64 columns (8 lanes x 8) and reduces ALL rows itself: lane (row slot, column
// vector) takes rows wave * 8 + slot + 32 p — at most 16 16-byte loads, four in flight at a time —, the wave's 8
// row slots are added by a fixed butterfly, the 4 waves through LDS in a fixed order, and the same
// workgroup applies the update: no partials, no ticket to wait for.  One path per (tokens, cols):
// a given input is always summed in the same order.
#define COLSQ_SMALL_TOKENS 512
template <int DT>
__device__ __forceinline__ void colsq_small_body(const ColsqArgs& a, const int bx, float (*lds)[64 * 8]) {
    constexpr int N = Vec<DT>::N;
    const void* __restrict__ x = a.x;
    const int64_t tokens = a.tokens, cols = a.cols;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane & 7, slot = lane >> 3;
    const int64_t ncvec = cols / N;
    const int64_t cvec = (int64_t)bx * 8 + cl;
    float acc[N];
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = 0.f;
    if (cvec < ncvec) {
        int64_t r = wave * 8 + slot;
        for (; r + 96 < tokens; r += 128) {          // four loads in flight (all sixteen at once: slower, measured)
            float f0[N], f1[N], f2[N], f3[N];
            const u32x4 va = ld16(x, r * ncvec + cvec), vb = ld16(x, (r + 32) * ncvec + cvec);
            const u32x4 vc = ld16(x, (r + 64) * ncvec + cvec), vd = ld16(x, (r + 96) * ncvec + cvec);
            Vec<DT>::unpack(va, f0); Vec<DT>::unpack(vb, f1);
            Vec<DT>::unpack(vc, f2); Vec<DT>::unpack(vd, f3);
#pragma unroll
            for (int i = 0; i < N; ++i) {            // same order as the one-row-at-a-time loop below
                acc[i] += f0[i] * f0[i];
                acc[i] += f1[i] * f1[i];
                acc[i] += f2[i] * f2[i];
                acc[i] += f3[i] * f3[i];
            }
        }
        for (; r < tokens; r += 32) {
            float f[N];
            Vec<DT>::unpack(ld16(x, r * ncvec + cvec), f);
#pragma unroll
            for (int i = 0; i < N; ++i) acc[i] += f[i] * f[i];
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float v = acc[i];
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (slot == 0) lds[wave][cl * N + i] = v;
    }
    __syncthreads();
    float decay = a.decay, n_new = a.n_new;
    if (a.n_dev) {                         // formed as the host form does (double, then float)
        const int64_t n0 = a.n_dev[0];
        decay = (float)((double)n0 / (double)(n0 + a.batch));
        n_new = (float)(n0 + a.batch);
    }
    const int64_t c = (int64_t)bx * 8 * N + threadIdx.x;
    if (threadIdx.x < 8 * N && c < cols) {
        const float s = (lds[0][threadIdx.x] + lds[1][threadIdx.x]) + (lds[2][threadIdx.x] + lds[3][threadIdx.x]);
        const float nrm = __builtin_sqrtf(s);  // torch.norm(...): sqrt of the sum of squares
        const float sq = nrm * nrm;            // ... ** 2
        if (a.raw) {
            a.scaler_row[c] = sq;
        } else {
            const float r = a.scaler_row[c] * decay;  // scaler_row *= n / (n + b)
            a.scaler_row[c] = r + sq / n_new;         // += ... / nsamples
        }
    }
    if (a.n_dev) {
        // every wave of this workgroup has read n_dev before thread 0 takes the grid-level
        // ticket: the workgroup that takes the LAST one bumps the count
        __syncthreads();
        if (threadIdx.x == 0 &&
            __hip_atomic_fetch_add(&a.tickets[a.colblocks], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
                (unsigned)a.colblocks - 1u) {
            __hip_atomic_store(&a.tickets[a.colblocks], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a.n_dev[0] += a.batch;
        }
    }
}

template <int DT, bool VECTOR>
__global__ __launch_bounds__(256) void colsq_kernel(const ColsqArgs a) {
    __shared__ float lds[4][64 * 8];
    __shared__ unsigned last;
    if (VECTOR && a.small) colsq_small_body<DT>(a, blockIdx.x, lds);
    else colsq_body<DT, VECTOR>(a, blockIdx.x, blockIdx.y, lds, &last);
}

// ALL hooked inputs of a transformer block for one calibration sample in ONE launch (the
// reference calls add_batch from one forward hook per Linear, W:240-252 / :521-533): the
// per-input launches are pure latency (11-17 us for 1.6-25 MB).  Workgroups of the inputs are
// laid out one input after another; same arithmetic per input as the one-input kernel.
struct ColsqMultiArgs {
    ColsqArgs it[ECOFLAP_COLSQ_MAX_ITEMS];
    int first[ECOFLAP_COLSQ_MAX_ITEMS];      // first workgroup of input i (INT_MAX past the last): one
    int n;                                   // vector load, no walk over the 100-byte records
};

template <int DT>
__global__ __launch_bounds__(256) void colsq_multi_kernel(const ColsqMultiArgs m) {
    __shared__ float lds[4][64 * 8];
    __shared__ unsigned last;
    int i = 0;
    const int wg = blockIdx.x;
#pragma unroll
    for (int j = 1; j < ECOFLAP_COLSQ_MAX_ITEMS; ++j) i += wg >= m.first[j] ? 1 : 0;     // (ascending)
    const ColsqArgs& a = m.it[i];
    const int local = wg - a.first_wg;
    const int bx = local % a.colblocks, by = local / a.colblocks;
    if (a.small) colsq_small_body<DT>(a, bx, lds);
    else if (a.vector) colsq_body<DT, true>(a, bx, by, lds, &last);
    else colsq_body<DT, false>(a, bx, by, lds, &last);
}

// The running mean replayed from per-batch statistics (raw rows of the kernels above), in
// global batch order: what the one-process loop computes, bit for bit, when the batches were
// reduced on different ranks.  sq: [J][ld] fp32; batches: int64[J] (device).
__global__ __launch_bounds__(256) void colsq_replay_kernel(float* __restrict__ scaler_row,
                                                           const float* __restrict__ sq,
                                                           const int64_t* __restrict__ batches,
                                                           int J, int64_t cols, int64_t ld,
                                                           int64_t n_before) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float row = scaler_row[c];
    int64_t n0 = n_before;
    for (int j = 0; j < J; ++j) {
        const int64_t b = batches[j];
        const float decay = (float)((double)n0 / (double)(n0 + b));
        const float n_new = (float)(n0 + b);
        const float r = row * decay;
        row = r + sq[(int64_t)j * ld + c] / n_new;
        n0 += b;
    }
    scaler_row[c] = row;
}

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
// ticket area at the head of the workspace: FIXED size, so that calls with different shapes that
// share one workspace never see another call's partial sums where they expect zeroed tickets
#define COLSQ_MAX_COLBLOCKS 4095
static inline size_t colsq_ticket_bytes(int64_t) { return (COLSQ_MAX_COLBLOCKS + 1) * sizeof(unsigned); }

extern "C" size_t ecoflap_colsqnorm_workspace_bytes(int64_t tokens, int64_t cols) {
    if (tokens <= 0 || cols <= 0) return 0;
    return colsq_ticket_bytes(cols) + (size_t)colsq_nchunks(tokens, cols) * (size_t)cols * sizeof(float);
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

static int colsq_fill(ColsqArgs& a, float* scaler_row, const void* x, int64_t tokens,
                      int64_t cols, int dtype, int64_t nsamples_before, int64_t* nsamples_dev,
                      int64_t batch, int raw, void* workspace) {
    if (tokens <= 0 || cols <= 0 || nsamples_before < 0 || batch <= 0) return ECOFLAP_ESIZE;
    if (!scaler_row || !x || !workspace) return ECOFLAP_ENULL;
    const int nvec = dtype == ECOFLAP_F32 ? 4 : 8;
    const bool vector = (cols % nvec == 0) && aligned16(x);
    const int64_t ncv = vector ? cols / nvec : cols;
    if ((ncv + 63) / 64 > COLSQ_MAX_COLBLOCKS) return ECOFLAP_ESIZE;
    a.x = x; a.tokens = tokens; a.cols = cols;
    a.rows_per_chunk = colsq_rows_per_chunk(tokens, cols);
    a.nchunks = colsq_nchunks(tokens, cols);
    a.tickets = (unsigned*)workspace;
    a.partial = (float*)((char*)workspace + colsq_ticket_bytes(cols));
    a.scaler_row = scaler_row;
    a.decay = (float)((double)nsamples_before / (double)(nsamples_before + batch));
    a.n_new = (float)(nsamples_before + batch);
    a.n_dev = nsamples_dev;
    a.batch = batch;
    a.colblocks = (int)((ncv + 63) / 64);
    a.first_wg = 0;
    a.vector = vector ? 1 : 0;
    a.raw = raw;
    a.small = 0;
    if (vector && tokens <= COLSQ_SMALL_TOKENS && (ncv + 7) / 8 <= COLSQ_MAX_COLBLOCKS) {
        a.small = 1;                       // one workgroup per 64 (fp32: 32) columns, all rows: colsq_small_body
        a.colblocks = (int)((ncv + 7) / 8);
        a.nchunks = 1;
        a.rows_per_chunk = (int)tokens;
    }
    return 0;
}

static int colsq_accum_impl(float* scaler_row, const void* x, int64_t tokens, int64_t cols,
                            int dtype, int64_t nsamples_before, int64_t* nsamples_dev,
                            int64_t batch, void* workspace, size_t workspace_bytes, void* stream) {
    if (!dtype_ok(dtype)) return ECOFLAP_EDTYPE;
    if (tokens <= 0 || cols <= 0) return ECOFLAP_ESIZE;
    if (workspace_bytes < ecoflap_colsqnorm_workspace_bytes(tokens, cols)) return ECOFLAP_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    ColsqArgs a;
    const int rc = colsq_fill(a, scaler_row, x, tokens, cols, dtype, nsamples_before, nsamples_dev,
                              batch, 0, workspace);
    if (rc) return rc;
    const dim3 grid((unsigned)a.colblocks, (unsigned)a.nchunks);
#define COLSQ(DT_)                                                                                \
    do {                                                                                          \
        if (a.vector) hipLaunchKernelGGL((colsq_kernel<DT_, true>), grid, dim3(256), 0, s, a);    \
        else hipLaunchKernelGGL((colsq_kernel<DT_, false>), grid, dim3(256), 0, s, a);            \
    } while (0)
    if (dtype == ECOFLAP_F32) COLSQ(ECOFLAP_F32);
    else if (dtype == ECOFLAP_F16) COLSQ(ECOFLAP_F16);
    else COLSQ(ECOFLAP_BF16);
#undef COLSQ
    ECO_CHECK_LAUNCH();
    return 0;
}

// Workspace of the multi-input call: ECOFLAP_COLSQ_MAX_ITEMS ticket areas of FIXED size at the
// head (input i always finds its tickets at the same place, whatever the shapes of the call, so
// a zeroed-once workspace can serve calls of different shapes one after another), then the
// inputs' partial sums.
static inline size_t colsq_item_partial_bytes(int64_t tokens, int64_t cols) {
    if (tokens <= 0 || cols <= 0) return 0;
    return ((size_t)colsq_nchunks(tokens, cols) * (size_t)cols * sizeof(float) + 255) & ~(size_t)255;
}

extern "C" size_t ecoflap_colsqnorm_multi_workspace_bytes(const ecoflap_colsq_item* items, int n) {
    if (!items || n <= 0) return 0;
    size_t total = (size_t)ECOFLAP_COLSQ_MAX_ITEMS * colsq_ticket_bytes(0);
    for (int i = 0; i < n; ++i) total += colsq_item_partial_bytes(items[i].tokens, items[i].cols);
    return total;
}

extern "C" int ecoflap_colsqnorm_accum_multi(const ecoflap_colsq_item* items, int n, int dtype,
                                             void* workspace, size_t workspace_bytes,
                                             void* stream) {
    if (!dtype_ok(dtype)) return ECOFLAP_EDTYPE;
    if (!items || !workspace) return ECOFLAP_ENULL;
    if (n <= 0 || n > ECOFLAP_COLSQ_MAX_ITEMS) return ECOFLAP_ESIZE;
    if (workspace_bytes < ecoflap_colsqnorm_multi_workspace_bytes(items, n)) return ECOFLAP_EWORKSPACE;
    ColsqMultiArgs m;
    m.n = n;
    size_t off = (size_t)ECOFLAP_COLSQ_MAX_ITEMS * colsq_ticket_bytes(0);
    int wg = 0;
    for (int i = 0; i < n; ++i) {
        const ecoflap_colsq_item& it = items[i];
        const int rc = colsq_fill(m.it[i], it.scaler_row, it.x, it.tokens, it.cols, dtype,
                                  it.nsamples_dev ? 0 : it.nsamples_before, it.nsamples_dev,
                                  it.batch, it.raw ? 1 : 0, workspace);
        if (rc) return rc;
        m.it[i].tickets = (unsigned*)((char*)workspace + (size_t)i * colsq_ticket_bytes(0));
        m.it[i].partial = (float*)((char*)workspace + off);
        m.it[i].first_wg = wg;
        m.first[i] = wg;
        wg += m.it[i].colblocks * m.it[i].nchunks;
        off += colsq_item_partial_bytes(it.tokens, it.cols);
    }
    for (int i = n; i < ECOFLAP_COLSQ_MAX_ITEMS; ++i) m.first[i] = 0x7fffffff;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == ECOFLAP_F32)
        hipLaunchKernelGGL((colsq_multi_kernel<ECOFLAP_F32>), dim3((unsigned)wg), dim3(256), 0, s, m);
    else if (dtype == ECOFLAP_F16)
        hipLaunchKernelGGL((colsq_multi_kernel<ECOFLAP_F16>), dim3((unsigned)wg), dim3(256), 0, s, m);
    else
        hipLaunchKernelGGL((colsq_multi_kernel<ECOFLAP_BF16>), dim3((unsigned)wg), dim3(256), 0, s, m);
    ECO_CHECK_LAUNCH();
    return 0;
}

extern "C" int ecoflap_colsq_replay(float* scaler_row, const float* sq, const int64_t* batches_dev,
                                    int n_batches, int64_t cols, int64_t ld,
                                    int64_t nsamples_before, void* stream) {
    if (!scaler_row || !sq || !batches_dev) return ECOFLAP_ENULL;
    if (n_batches <= 0 || cols <= 0 || ld < cols || nsamples_before < 0) return ECOFLAP_ESIZE;
    hipLaunchKernelGGL(colsq_replay_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, scaler_row, sq, batches_dev, n_batches, cols, ld,
                       nsamples_before);
    ECO_CHECK_LAUNCH();
    return 0;
}

// =====================================================================================
// K7 common: sqrt(scaler_row) once per matrix (correctly rounded, = torch.sqrt)
// =====================================================================================
// Block-level launches: every K7 kernel below serves up to ECOFLAP_WANDA_MAX_ITEMS matrices (all
// the Linears of one transformer block) from ONE launch; the item of a workgroup / wave is found
// from a prefix table in the kernel arguments (wave-uniform scalar search).
#define WMAX ECOFLAP_WANDA_MAX_ITEMS

template <typename G>
static __device__ __forceinline__ int group_item(const G& g, int64_t idx) {
    int it = 0;
    while (it + 1 < g.n && idx >= g.start[it + 1]) ++it;
    return it;
}

struct SqrtGroup {
    int n;
    int32_t start[WMAX + 1];      // workgroup prefix (256 columns each)
    const float* src[WMAX];
    float* dst[WMAX];
    int64_t cols[WMAX];
    uint32_t* zero[WMAX];         // matrix mode: the item's selection state, cleared here (no memset launch)
    int zero_words;
};
__global__ __launch_bounds__(256) void sqrt_cols_kernel(const SqrtGroup g) {
    const int it = group_item(g, blockIdx.x);
    const int lb = blockIdx.x - g.start[it], nb = g.start[it + 1] - g.start[it];
    const int64_t c = (int64_t)lb * 256 + threadIdx.x;
    if (c < g.cols[it]) g.dst[it][c] = __builtin_sqrtf(g.src[it][c]);
    if (g.zero[it])
        for (int i = lb * 256 + threadIdx.x; i < g.zero_words; i += nb * 256) g.zero[it][i] = 0u;
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

struct RowsGroup {
    int n;
    int32_t start[WMAX + 1];      // row prefix over the group's matrices
    void* w[WMAX];
    const float* sq[WMAX];
    int64_t cols[WMAX];
    int64_t k[WMAX];
    uint8_t* mask[WMAX];
};

// ---- helpers shared by the two register forms of the rows mode ----------------------------
// wave-wide sum on the DPP network (row shifts inside the four 16-lane rows, then the two row
// broadcasts gfx9 has): 6 full-rate adds, no LDS crossbar, result uniform
static __device__ __forceinline__ uint32_t wave_sum_dpp(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);   // row_bcast:15
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);   // row_bcast:31
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// pruned elements -> +0 on the stored bits themselves (kept elements are not re-encoded), and the
// row's uint8 mask bytes
template <int DT>
static __device__ __forceinline__ void zero_pruned(u32x4& wv, const bool* prune) {
    if (Vec<DT>::N == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) wv[i] = prune[i] ? 0u : wv[i];
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            wv[q] &= (prune[2 * q] ? 0u : 0xffffu) | (prune[2 * q + 1] ? 0u : 0xffff0000u);
    }
}
// `m <= T` goes: the common ending of a row (no mask bytes wanted, no tie cut by column order).
// 16-bit dtypes: one compare and ONE sub-dword select per element — v_cndmask_b32_sdwa writes the
// element's half of the packed register and preserves the other (the compiler's form is compare,
// select, select, or, and per PAIR plus the bool bookkeeping: ~5 per element in the listing).
// Element halves are visited low halves first, high halves second: consecutive writes never
// touch the same register (the 16-bit-destination forwarding hazard of this ISA).
template <int DT>
static __device__ __forceinline__ void zero_le(u32x4& wv, const uint32_t* m, uint32_t T) {
    if constexpr (Vec<DT>::N == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) wv[i] = m[i] <= T ? 0u : wv[i];
    } else {
        const uint32_t zero = 0u;
        uint32_t r0 = wv[0], r1 = wv[1], r2 = wv[2], r3 = wv[3];
#define ECO_SEL_LO(R_, M_) asm volatile("v_cmp_lt_u32 vcc, %2, %1\n\tv_cndmask_b32_sdwa %0, %3, %0, vcc dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:WORD_0" : "+v"(R_) : "v"(M_), "s"(T), "v"(zero) : "vcc")
#define ECO_SEL_HI(R_, M_) asm volatile("v_cmp_lt_u32 vcc, %2, %1\n\tv_cndmask_b32_sdwa %0, %3, %0, vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:WORD_1" : "+v"(R_) : "v"(M_), "s"(T), "v"(zero) : "vcc")
        ECO_SEL_LO(r0, m[0]); ECO_SEL_LO(r1, m[2]); ECO_SEL_LO(r2, m[4]); ECO_SEL_LO(r3, m[6]);
        ECO_SEL_HI(r0, m[1]); ECO_SEL_HI(r1, m[3]); ECO_SEL_HI(r2, m[5]); ECO_SEL_HI(r3, m[7]);
#undef ECO_SEL_LO
#undef ECO_SEL_HI
        wv[0] = r0; wv[1] = r1; wv[2] = r2; wv[3] = r3;
    }
}

template <int N>
static __device__ __forceinline__ void store_mask_bytes(uint8_t* mrow, const bool* prune) {
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (i < 4) lo |= (prune[i] ? 1u : 0u) << (8 * i);
        else hi |= (prune[i] ? 1u : 0u) << (8 * (i - 4));
    }
    *(uint32_t*)mrow = lo;
    if (N == 8) *(uint32_t*)(mrow + 4) = hi;
}

// count of row elements below cand: per-lane compare + add-with-carry, one DPP reduction
template <int NV, int N>
static __device__ __forceinline__ uint32_t wave_count_less(const uint32_t (&m)[NV][N], uint32_t cand) {
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int i = 0; i < N; ++i) c += (m[j][i] < cand) ? 1u : 0u;
    return wave_sum_dpp(c);
}

// The rounds on bits 31..16 only look at the HIGH HALVES of the metrics (cand has no low bits
// there), two elements per register: hi16(x) >= c for both halves of a word with ONE subtraction,
//   d = (x | 0x80008000) - (c & 0x7fff) * 0x10001   (no borrow leaves a half; bit 15 / 31 of d is
//   set iff low15(x) >= low15(c));   c < 0x8000: x >= c <=> bit15(x | d);  else bit15(x & d),
// counted with v_bcnt: 3 instructions per two elements instead of 4 (compare + add-with-carry each,
// and the v_cmp -> carry-in hazard of this ISA costs a wait state per compare on top).
template <int NV, int N>
static __device__ __forceinline__ void gather_hi16(const uint32_t (&m)[NV][N], uint32_t (&x)[NV * N / 2]) {
#pragma unroll
    for (int w = 0; w < NV * N / 2; ++w) {
        const int e = 2 * w;
        // bytes [b.3 b.2 a.3 a.2]: element e in the low half, e+1 in the high half
        x[w] = __builtin_amdgcn_perm(m[(e + 1) / N][(e + 1) % N], m[e / N][e % N], 0x07060302u);
    }
}
template <int W, bool KEEP>
static __device__ __forceinline__ uint32_t wave_count_hi16_ge(const uint32_t (&x)[W], const uint32_t (&xg)[KEEP ? W : 1],
                                                              uint32_t c) {
    const uint32_t cl = (c & 0x7fffu) * 0x10001u;
    const uint32_t K = 0x80008000u;
    uint32_t cnt = 0;
    if (c & 0x8000u) {         // wave-uniform
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const uint32_t d = (KEEP ? xg[KEEP ? w : 0] : (x[w] | K)) - cl;
            const uint32_t f = __builtin_amdgcn_bitop3_b32(d, x[w], K, 0x80);      // d & x & K
            asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(cnt) : "v"(f));
        }
    } else {
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const uint32_t d = (KEEP ? xg[KEEP ? w : 0] : (x[w] | K)) - cl;
            const uint32_t f = __builtin_amdgcn_bitop3_b32(d, x[w], K, 0xa8);      // (d | x) & K
            asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(cnt) : "v"(f));
        }
    }
    return wave_sum_dpp(cnt);
}

// sum (or max) of one wave-uniform value per wave over the workgroup's four waves, result uniform:
// lane 0 of each wave writes its value, one barrier (double-buffered by call parity), four reads
static __device__ __forceinline__ uint32_t block_sum4(uint32_t wave_value, uint32_t* lds8, int phase,
                                                      int wave, int lane) {
    uint32_t* buf = lds8 + 4 * (phase & 1);
    if (lane == 0) buf[wave] = wave_value;
    __syncthreads();
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)(buf[0] + buf[1] + buf[2] + buf[3]));
}
static __device__ __forceinline__ uint32_t block_max4(uint32_t wave_value, uint32_t* lds8, int phase,
                                                      int wave, int lane) {
    uint32_t* buf = lds8 + 4 * (phase & 1);
    if (lane == 0) buf[wave] = wave_value;
    __syncthreads();
    const uint32_t a = buf[0] > buf[1] ? buf[0] : buf[1], b = buf[2] > buf[3] ? buf[2] : buf[3];
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)(a > b ? a : b));
}

// ---- the k-th smallest by ONE histogram pass instead of a bisection (round 4) --------------
// profiles/r04_secondary/k7_rows_phases.md: of the ~41 VALU instructions per element of the
// bisection form, ~21 were the bisection itself (71 per probe and lane, 9-14 probes), 5 the
// compaction, 3 the low bits.  Here the top of the search costs 3 VALU + one LDS atomic per
// element: the metrics' bit patterns are binned RELATIVE TO THE ROW MAXIMUM,
//     bin = sat_sub(m, mx - (2^28 - 1)) >> 18          (1024 bins of 2^18 codes = 1/32 octave,
//                                                        32 octaves below the maximum; what lies
//                                                        further down shares bin 0)
// into an LDS histogram (ds_add_u32), the bin holding the k-th smallest is found by one scan
// (16 bins per lane + a wave prefix), and only that bin's elements (a few dozen of a 2048- or
// 5120-wide row) go on to the 18 low-bit rounds.  Exact: the bins partition the values in
// order, every count is a count of elements.  A bin with more than RH_CAND elements (massive
// ties) or a k-th smallest in the shared bin 0 takes the generic bit-by-bit search over all
// elements.
#define RH_BINS 1024
#define RH_SHIFT 18
#define RH_SPAN ((1u << 28) - 1u)         // RH_BINS << RH_SHIFT, minus one
#define RH_CAND 128                        // candidates one wave finishes (two per lane)

static __device__ __forceinline__ uint32_t rh_lowbound(uint32_t mx) { return mx > RH_SPAN ? mx - RH_SPAN : 0u; }
static __device__ __forceinline__ uint32_t rh_bin(uint32_t m, uint32_t lowbound) {
    return __builtin_elementwise_sub_sat(m, lowbound) >> RH_SHIFT;
}
// wave-wide inclusive prefix sum on the DPP network; `total` = the wave's sum (uniform)
static __device__ __forceinline__ uint32_t wave_scan_dpp(uint32_t v, uint32_t& total) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);   // row_bcast:15
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);   // row_bcast:31
    total = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
    return v;
}
// One wave scans the RH_BINS counters (lane l: bins [16 l, 16 l + 16)): the bin holding the k-th
// smallest (1-indexed), the number of elements in lower bins, and that bin's population — all
// wave-uniform.  The counters sum to >= k.
static __device__ __forceinline__ void rh_find(const uint32_t* hist, uint32_t k, int lane, uint32_t& bin,
                                               uint32_t& less_below, uint32_t& ncand) {
    constexpr int PER = RH_BINS / 64;
    uint32_t c[PER];
#pragma unroll
    for (int q = 0; q < PER / 4; ++q) {
        const u32x4 v = ((const u32x4*)hist)[lane * (PER / 4) + q];
#pragma unroll
        for (int i = 0; i < 4; ++i) c[4 * q + i] = v[i];
    }
#pragma unroll
    for (int j = 1; j < PER; ++j) c[j] += c[j - 1];               // local inclusive sums
    uint32_t total;
    const uint32_t incl = wave_scan_dpp(c[PER - 1], total);
    const uint32_t P = incl - c[PER - 1];                        // elements in the lanes before
    const int32_t kk = (int32_t)k - (int32_t)P;                  // <= 0: the bin lies in an earlier lane
    uint32_t cnt = 0, below = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const bool lt = (int32_t)c[j] < kk;
        cnt += lt ? 1u : 0u;
        below = lt ? c[j] : below;
    }
    const uint64_t has = __ballot(incl >= k);                    // first lane whose range reaches k
    const int L = __builtin_ctzll(has);
    const uint32_t cntL = (uint32_t)__builtin_amdgcn_readlane((int)cnt, L);
    bin = (uint32_t)L * PER + cntL;
    less_below = (uint32_t)__builtin_amdgcn_readlane((int)(P + below), L);
    ncand = hist[bin];                                           // uniform address: broadcast read
    ncand = (uint32_t)__builtin_amdgcn_readfirstlane((int)ncand);
}

// One 256-thread workgroup per row (rows beyond the wave form's reach, and the long rows of a
// block whose short rows run in the wave form): the same search — bisection over the packed high
// halves, then the compacted bucket — with the per-probe count summed over the four waves through
// LDS (one barrier per probe).
template <int DT, int NV, bool HIST>
static __device__ __forceinline__ void rows_reg_body(const RowsGroup& g, int64_t grow, uint32_t* lds8,
                                                     uint32_t* wave4, uint32_t* cand_lds, uint32_t* res3,
                                                     uint32_t* hist) {
    constexpr int N = Vec<DT>::N;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int it = group_item(g, grow);
    const int64_t row = grow - g.start[it];
    const int64_t cols = g.cols[it], k = g.k[it];
    const float* __restrict__ sq = g.sq[it];
    uint8_t* mask_out = g.mask[it];
    const int64_t nvec = cols / N;
    void* wrow = (char*)g.w[it] + row * cols * Vec<DT>::BYTES;
    u32x4 wv[NV];
    uint32_t m[NV][N];
#pragma unroll
    for (int j = 0; j < NV; ++j)
        if (tid + 256 * j < nvec) wv[j] = ld16(wrow, tid + 256 * j);
    if constexpr (HIST) {                      // the row's histogram (+ the candidate counter behind it)
        const u32x4 zero = {0u, 0u, 0u, 0u};
        static_assert(RH_BINS == 1024, "one 16-byte clear per thread");
        ((u32x4*)hist)[tid] = zero;
        if (tid == 0) hist[RH_BINS] = 0u;
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int64_t v = tid + 256 * j;
        if (v < nvec) {
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
    int phase = 0;
    if (!all) {
        uint32_t mx = 0;
#pragma unroll
        for (int j = 0; j < NV; ++j)
            if (tid + 256 * j < nvec) {
#pragma unroll
                for (int i = 0; i < N; ++i) mx = m[j][i] > mx ? m[j][i] : mx;
            }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t o = __shfl_xor(mx, off, 64);
            mx = o > mx ? o : mx;
        }
        // (the barrier inside also orders the histogram's clears before the adds below)
        mx = block_max4((uint32_t)__builtin_amdgcn_readfirstlane((int)mx), lds8, phase++, wave, lane);
        uint32_t lo, less_below, ncand;        // the interval [lo, lo + 2^bits) holding the k-th smallest
        int bits;
        bool crowded;
        if constexpr (HIST) {
            const uint32_t lowbound = rh_lowbound(mx);
#pragma unroll
            for (int j = 0; j < NV; ++j)
                if (tid + 256 * j < nvec) {
#pragma unroll
                    for (int i = 0; i < N; ++i)
                        __hip_atomic_fetch_add(&hist[rh_bin(m[j][i], lowbound)], 1u, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            __syncthreads();
            // every wave scans the counters for itself: the same uniform result in all four,
            // no second barrier to hand it round
            uint32_t bin;
            rh_find(hist, (uint32_t)k, lane, bin, less_below, ncand);
            if (bin == 0 && lowbound > 0) {
                lo = 0; bits = 32; less_below = 0; crowded = true;
            } else {
                lo = lowbound + (bin << RH_SHIFT); bits = RH_SHIFT; crowded = ncand > RH_CAND;
            }
        } else {
            const int top = mx ? 31 - __builtin_clz(mx) : -1;
            constexpr int WH = NV * N / 2;
            constexpr uint32_t ALL = 256 * NV * N;
            uint32_t xh[WH], xg[WH];
            gather_hi16(m, xh);
#pragma unroll
            for (int w = 0; w < WH; ++w) xg[w] = xh[w] | 0x80008000u;
            const uint32_t mx16 = mx >> 16;
            uint32_t blo = 0, flo = 0, bhi = 65536u, fhi = ALL;
            if (mx16 < 0xffffu) { bhi = mx16 + 1; fhi = (uint32_t)cols; }
            while (bhi - blo > 1) {
                const uint32_t mid = (blo + bhi) >> 1;
                const uint32_t fm = ALL - block_sum4(wave_count_hi16_ge<WH, true>(xh, xg, mid), lds8, phase++, wave, lane);
                if (fm < (uint32_t)k) { blo = mid; flo = fm; } else { bhi = mid; fhi = fm; }
            }
            lo = blo << 16;
            less_below = flo;
            ncand = fhi - flo;
            bits = (top < 15 ? top : 15) + 1;
            crowded = ncand > RH_CAND;
        }
        if (!crowded) {                                   // block-uniform
            // compact the interval's elements; ONE wave finishes the low bits on them with
            // ballots — `bits` rounds without a block barrier or a pass over the registers
            const uint32_t span = 1u << bits;
            if constexpr (HIST) {
                // order among the candidates does not matter: a counter hands out the slots
#pragma unroll
                for (int j = 0; j < NV; ++j)
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        const uint32_t rel = m[j][i] - lo;
                        if (rel < span) {
                            const uint32_t p_ = __hip_atomic_fetch_add(&hist[RH_BINS], 1u, __ATOMIC_RELAXED,
                                                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                            cand_lds[p_] = rel;
                        }
                    }
            } else {
                uint32_t mine = 0;
#pragma unroll
                for (int j = 0; j < NV; ++j)
#pragma unroll
                    for (int i = 0; i < N; ++i) mine += ((m[j][i] - lo) < span) ? 1u : 0u;
                uint32_t total;
                uint32_t pos = block_scan_256(mine, wave4, total) - mine;
                if (mine) {
#pragma unroll
                    for (int j = 0; j < NV; ++j)
#pragma unroll
                        for (int i = 0; i < N; ++i)
                            if ((m[j][i] - lo) < span) cand_lds[pos++] = m[j][i] - lo;
                }
            }
            __syncthreads();
            if (wave == 0) {
                const bool v0 = (uint32_t)lane < ncand, v1 = (uint32_t)lane + 64 < ncand;
                const uint32_t c0 = v0 ? cand_lds[lane] : 0u, c1 = v1 ? cand_lds[lane + 64] : 0u;
                uint32_t off = 0;
                for (int bit = bits - 1; bit >= 0; --bit) {
                    const uint32_t cand = off | (1u << bit);
                    const uint32_t c = less_below + (uint32_t)__popcll(__ballot(v0 && c0 < cand)) +
                                       (uint32_t)__popcll(__ballot(v1 && c1 < cand));
                    if (c < (uint32_t)k) off = cand;
                }
                const uint32_t less = less_below + (uint32_t)__popcll(__ballot(v0 && c0 < off)) +
                                      (uint32_t)__popcll(__ballot(v1 && c1 < off));
                const uint32_t eq = (uint32_t)__popcll(__ballot(v0 && c0 == off)) +
                                    (uint32_t)__popcll(__ballot(v1 && c1 == off));
                if (lane == 0) { res3[0] = lo + off; res3[1] = less; res3[2] = eq; }
            }
            __syncthreads();
            T = (uint32_t)__builtin_amdgcn_readfirstlane((int)res3[0]);
            total_equal = (uint32_t)__builtin_amdgcn_readfirstlane((int)res3[2]);
            take_equal = (uint32_t)k - (uint32_t)__builtin_amdgcn_readfirstlane((int)res3[1]);
        } else {                                          // crowded interval: all elements, bit by bit
            uint32_t less = less_below, off = 0;
            for (int bit = bits - 1; bit >= 0; --bit) {
                const uint32_t cand = lo + (off | (1u << bit));
                const uint32_t c = block_sum4(wave_count_less(m, cand), lds8, phase++, wave, lane);
                if (c < (uint32_t)k) { off |= 1u << bit; less = c; }
            }
            T = lo + off;
            uint32_t eq = 0;
#pragma unroll
            for (int j = 0; j < NV; ++j)
#pragma unroll
                for (int i = 0; i < N; ++i) eq += (m[j][i] == T) ? 1u : 0u;
            total_equal = block_sum4(wave_sum_dpp(eq), lds8, phase++, wave, lane);
            take_equal = (uint32_t)k - less;
        }
    }
    const bool ordered = !all && (take_equal < total_equal);   // ties cut by column order
    if (!ordered) {
        const uint32_t Tq = all ? 0xffffffffu : T;
        if (!mask_out) {                       // block-uniform: the production call
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int64_t v = tid + 256 * j;
                if (v < nvec) {
                    zero_le<DT>(wv[j], m[j], Tq);
                    st16(wrow, v, wv[j]);
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int64_t v = tid + 256 * j;
            if (v < nvec) {
                bool prune[N];
#pragma unroll
                for (int i = 0; i < N; ++i) prune[i] = m[j][i] <= Tq;
                zero_pruned<DT>(wv[j], prune);
                st16(wrow, v, wv[j]);
                store_mask_bytes<N>(mask_out + row * cols + v * N, prune);
            }
        }
        return;
    }
    uint32_t running = 0;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int64_t v = tid + 256 * j;
        // block-uniform branch: equal-to-T elements in lower columns come first
        uint32_t mine = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) mine += (m[j][i] == T) ? 1u : 0u;
        uint32_t total;
        const uint32_t incl = block_scan_256(mine, wave4, total);
        const uint32_t rank0 = running + incl - mine;
        running += total;
        if (v < nvec) {
            bool prune[N];
            uint32_t seen = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const bool eqT = (m[j][i] == T);
                prune[i] = (m[j][i] < T) || (eqT && (rank0 + seen) < take_equal);
                seen += eqT ? 1u : 0u;
            }
            zero_pruned<DT>(wv[j], prune);
            st16(wrow, v, wv[j]);
            if (mask_out) store_mask_bytes<N>(mask_out + row * cols + v * N, prune);
        }
    }
}

template <int DT, int NV, bool HIST>
__global__ __launch_bounds__(256) void wanda_rows_reg_kernel(const RowsGroup g) {
    __shared__ uint32_t lds8[8];
    __shared__ uint32_t wave4[4];
    __shared__ uint32_t cand_lds[RH_CAND];
    __shared__ uint32_t res3[3];
    __shared__ __attribute__((aligned(16))) uint32_t hist[HIST ? RH_BINS + 4 : 4];
    rows_reg_body<DT, NV, HIST>(g, blockIdx.x, lds8, wave4, cand_lds, res3, hist);
}

// -------------------------------------------------------------------------------------
// K7 rows mode, wave form (rows of up to 64*12 vectors): ONE 64-lane wave per row, four rows
// per workgroup — the search needs no LDS traffic and no barrier at all.
// -------------------------------------------------------------------------------------
// One 64-lane wave per row, four rows per workgroup.  (Tried and measured slower on MI355X: a
// persistent wave walking a run of rows with the next row's loads in flight during the search —
// all waves then load, search and store in phase and the memory and VALU phases stop overlapping
// across waves; seeding the search from the previous row's threshold through a word in the
// workspace — loads of many waves from one address serialise like atomics, ~30 ns each.)
// FULL: every row of the launch has exactly 64 * NV vectors (FlanT5's 2048-wide rows in bf16):
// no padding values, no bounds tests (the padding's defaults alone were 61 register moves).
template <int DT, int NV, bool HIST, bool FULL = false>
static __device__ __forceinline__ void rows_wave_body(const RowsGroup& g, int64_t first_row,
                                                      uint32_t (*cand_lds_all)[64 * 2],
                                                      uint32_t (*hist_all)[RH_BINS]) {
    constexpr int N = Vec<DT>::N;
    const int lane = threadIdx.x & 63;
    // (the wave index is uniform, which the compiler cannot see through threadIdx: without the
    // readfirstlane every per-row scalar — item, k, cols, the search bounds — lives in VGPRs)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t grow = first_row + wave;
    if (grow >= g.start[g.n]) return;          // whole wave leaves together
    const int it = group_item(g, grow);
    const int64_t row = grow - g.start[it];
    const int64_t cols = g.cols[it], k = g.k[it];
    const float* __restrict__ sq = g.sq[it];
    uint8_t* mask_out = g.mask[it];
    const int64_t nvec = cols / N;
    void* wrow = (char*)g.w[it] + row * cols * Vec<DT>::BYTES;
    uint32_t* cand_lds = cand_lds_all[wave];
    u32x4 wv[NV];
    uint32_t m[NV][N];
#pragma unroll
    for (int j = 0; j < NV; ++j)
        if (FULL || lane + 64 * j < nvec) wv[j] = ld16(wrow, lane + 64 * j);
    if constexpr (HIST) {                      // the wave's histogram, cleared while the loads fly
        const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int q = 0; q < RH_BINS / 256; ++q) ((u32x4*)hist_all[wave])[lane + 64 * q] = zero;
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int64_t v = lane + 64 * j;
        if (FULL || v < nvec) {
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
        // k-th smallest (1-indexed) = largest T with #(m < T) < k.  First an interval
        // [lo, lo + 2^bits) that holds it, with the number of elements below it — from the
        // histogram of the metrics relative to the row maximum (HIST) or from a bisection over the
        // packed high halves —; then only the elements inside the interval (typically ~1 % of a
        // row) can still change a count: they are compacted to at most two per lane and the low
        // bits are found bit by bit with 2 ballots per round instead of a pass over the row.
        uint32_t mx = 0;
#pragma unroll
        for (int j = 0; j < NV; ++j)
            if (FULL || lane + 64 * j < nvec) {
#pragma unroll
                for (int i = 0; i < N; ++i) mx = m[j][i] > mx ? m[j][i] : mx;
            }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t o = __shfl_xor(mx, off, 64);
            mx = o > mx ? o : mx;
        }
        mx = (uint32_t)__builtin_amdgcn_readfirstlane((int)mx);      // uniform: scalar loop control
        uint32_t lo, less_below, ncand;        // less_below = #(m < lo), ncand = population of the interval
        int bits;
        bool crowded;
        if constexpr (HIST) {
            uint32_t* hist = hist_all[wave];
            const uint32_t lowbound = rh_lowbound(mx);
#pragma unroll
            for (int j = 0; j < NV; ++j)
                if (FULL || lane + 64 * j < nvec) {      // (padding must not be binned: it lies above mx)
#pragma unroll
                    for (int i = 0; i < N; ++i)
                        __hip_atomic_fetch_add(&hist[rh_bin(m[j][i], lowbound)], 1u, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            // (one wave's LDS operations are performed in issue order: the clears above, these
            // adds and the reads of the scan need no barrier between them)
            uint32_t bin;
            rh_find(hist, (uint32_t)k, lane, bin, less_below, ncand);
            if (bin == 0 && lowbound > 0) {      // k-th smallest more than 32 octaves below the maximum
                lo = 0; bits = 32; less_below = 0; crowded = true;
            } else {
                lo = lowbound + (bin << RH_SHIFT); bits = RH_SHIFT; crowded = ncand > RH_CAND;
            }
        } else {
            const int top = mx ? 31 - __builtin_clz(mx) : -1;
            constexpr int WH = NV * N / 2;
            constexpr bool KEEPG = NV <= 4;
            constexpr uint32_t ALL = 64 * NV * N;              // padding (0xffff....) is never "less"
            uint32_t xh[WH], xg[KEEPG ? WH : 1];
            gather_hi16(m, xh);
            if (KEEPG) {
#pragma unroll
                for (int w = 0; w < WH; ++w) xg[KEEPG ? w : 0] = xh[w] | 0x80008000u;
            }
            // f(c) = #(hi16(m) < c) is monotone; wanted: p16 = max{c : f(c) < k}.  Bisection on
            // [lo, hi] with f(lo) < k <= f(hi); the last two values give the bucket's population
            const uint32_t mx16 = mx >> 16;
            uint32_t blo = 0, flo = 0, bhi = 65536u, fhi = ALL;
            if (mx16 < 0xffffu) { bhi = mx16 + 1; fhi = (uint32_t)cols; }   // every real element is below
            while (bhi - blo > 1) {
                const uint32_t mid = (blo + bhi) >> 1;
                const uint32_t fm = ALL - wave_count_hi16_ge<WH, KEEPG>(xh, xg, mid);
                if (fm < (uint32_t)k) { blo = mid; flo = fm; } else { bhi = mid; fhi = fm; }
            }
            lo = blo << 16;
            less_below = flo;
            ncand = fhi - flo;
            bits = (top < 15 ? top : 15) + 1;
            crowded = ncand > RH_CAND;
        }
        constexpr uint32_t CMAX = 2;             // candidates per lane (RH_CAND = 64 * CMAX)
        if (!crowded) {                          // wave-uniform
            const uint32_t span = 1u << bits;    // (bits <= 18 here)
            // (tried: a counter handing out the slots, `cand[atomicAdd(counter, 1)] = rel` under the
            // lanes' own branch — the compiler turns it into a wave-aggregated atomic, 11 VALU per
            // element slot with a hit against 6 for the ballot form below, plus an LDS round trip)
            uint32_t pos = 0;
#pragma unroll
            for (int j = 0; j < NV; ++j)
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    const uint32_t rel = m[j][i] - lo;
                    const bool hit = rel < span;
                    const uint64_t mask = __ballot(hit);
                    if (mask) {                  // wave-uniform: a third of the element slots have no candidate
                        if (hit) cand_lds[pos + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u))] = rel;
                        pos += (uint32_t)__popcll(mask);
                    }
                }
            // (same wave wrote and reads: program order, no barrier needed on one SIMD)
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): LDS writes done
            uint32_t cr[CMAX];
            bool cv[CMAX];
#pragma unroll
            for (uint32_t c = 0; c < CMAX; ++c) {
                cv[c] = lane + 64 * c < ncand;
                cr[c] = cv[c] ? cand_lds[lane + 64 * c] : 0u;
            }
            uint32_t off = 0;                    // T = lo + off
            for (int bit = bits - 1; bit >= 0; --bit) {
                const uint32_t cand = off | (1u << bit);
                uint32_t c = less_below;
#pragma unroll
                for (uint32_t q = 0; q < CMAX; ++q) c += (uint32_t)__popcll(__ballot(cv[q] && cr[q] < cand));
                if (c < (uint32_t)k) off = cand;
            }
            T = lo + off;
            uint32_t less = less_below;
#pragma unroll
            for (uint32_t q = 0; q < CMAX; ++q) {
                less += (uint32_t)__popcll(__ballot(cv[q] && cr[q] < off));
                total_equal += (uint32_t)__popcll(__ballot(cv[q] && cr[q] == off));
            }
            take_equal = (uint32_t)k - less;
        } else {                                 // crowded interval (few distinct values): all elements
            uint32_t less = less_below, off = 0;
            for (int bit = bits - 1; bit >= 0; --bit) {
                const uint32_t cand = lo + (off | (1u << bit));
                const uint32_t c = wave_count_less(m, cand);
                if (c < (uint32_t)k) { off |= 1u << bit; less = c; }
            }
            T = lo + off;
#pragma unroll
            for (int j = 0; j < NV; ++j)
#pragma unroll
                for (int i = 0; i < N; ++i)
                    total_equal += (uint32_t)__popcll(__ballot(m[j][i] == T));
            take_equal = (uint32_t)k - less;
        }
    }
    const bool ordered = !all && (take_equal < total_equal);
    if (!ordered) {
        // every element equal to T goes (or, with k >= cols, every element): prune = m <= T
        const uint32_t Tq = all ? 0xffffffffu : T;
        if (!mask_out) {                       // wave-uniform: the production call
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int64_t v = lane + 64 * j;
                if (FULL || v < nvec) {
                    zero_le<DT>(wv[j], m[j], Tq);
                    st16(wrow, v, wv[j]);
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int64_t v = lane + 64 * j;
            if (FULL || v < nvec) {
                bool prune[N];
#pragma unroll
                for (int i = 0; i < N; ++i) prune[i] = m[j][i] <= Tq;
                zero_pruned<DT>(wv[j], prune);
                st16(wrow, v, wv[j]);
                store_mask_bytes<N>(mask_out + row * cols + v * N, prune);
            }
        }
        return;
    }
    uint32_t running = 0;                      // ties cut by column order
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int64_t v = lane + 64 * j;
        uint32_t mine = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) mine += (m[j][i] == T) ? 1u : 0u;
        uint32_t x = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(x, off, 64);
            if (lane >= off) x += y;
        }
        const uint32_t rank0 = running + x - mine;
        running += __shfl(x, 63, 64);
        if (v < nvec) {
            bool prune[N];
            uint32_t seen = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const bool eqT = (m[j][i] == T);
                prune[i] = (m[j][i] < T) || (eqT && (rank0 + seen) < take_equal);
                seen += eqT ? 1u : 0u;
            }
            zero_pruned<DT>(wv[j], prune);
            st16(wrow, v, wv[j]);
            if (mask_out) store_mask_bytes<N>(mask_out + row * cols + v * N, prune);
        }
    }
}

template <int DT, int NV, bool HIST, bool FULL>
__global__ __launch_bounds__(256, 4) void wanda_rows_wave_kernel(const RowsGroup g) {
    __shared__ uint32_t cand_lds_all[4][64 * 2];
    __shared__ __attribute__((aligned(16))) uint32_t hist_all[HIST ? 4 : 1][RH_BINS];
    rows_wave_body<DT, NV, HIST, FULL>(g, (int64_t)blockIdx.x * 4, cand_lds_all, hist_all);
}

// A block's short rows (wave form) and long rows (workgroup form) in ONE grid: the long rows of a
// T5 block (2048 rows of 5120 columns) are too few to fill the chip in a launch of their own
// (measured: 33 us for 22 % of the block's bytes, after the 58 us of the other 78 %), and launches
// on one stream do not overlap; side streams joined by events cost more than they gave.  The
// long-row workgroups come first in the grid so that they are not the tail.
template <int DT, int NVW, int NVR, bool HIST, bool FULL>
__global__ __launch_bounds__(256, 4) void wanda_rows_fused_kernel(const RowsGroup gw, const RowsGroup gr) {
    __shared__ uint32_t cand_lds_all[4][64 * 2];      // wave form: 2 candidates per lane and wave
    __shared__ uint32_t lds8[8];
    __shared__ uint32_t wave4[4];
    __shared__ uint32_t res3[4];
    // wave form: one histogram per wave; workgroup form: the first one (+ its counter in res3's
    // neighbour would alias: the counter lives behind the LAST histogram instead)
    __shared__ __attribute__((aligned(16))) uint32_t hist_all[HIST ? 4 : 1][RH_BINS];
    const int64_t n_long = gr.start[gr.n];
    if ((int64_t)blockIdx.x < n_long)
        rows_reg_body<DT, NVR, HIST>(gr, blockIdx.x, lds8, wave4, &cand_lds_all[0][0], res3, &hist_all[0][0]);
    else
        rows_wave_body<DT, NVW, HIST, FULL>(gw, ((int64_t)blockIdx.x - n_long) * 4, cand_lds_all, hist_all);
}

#define ROWS_WAVE_MAX_NVEC 256                           // 4 vectors per lane (2048 bf16 columns)
#define ROWS_FUSE_MAX_NVEC 1024                          // workgroup-form rows that may share a grid: 4 vectors per thread
static inline int rows_wave_class(int64_t nvec) {       // vectors per lane, one wave per row
    const int nv = (int)((nvec + 63) / 64);
    return nv <= 1 ? 1 : (nv <= 2 ? 2 : (nv <= 4 ? 4 : 0));
}
static inline int rows_reg_class(int64_t nvec) {        // vectors per thread, one workgroup per row
    const int nv = (int)((nvec + 255) / 256);
    static const int classes[] = {1, 2, 3, 4, 6, 8, 12, 16};
    for (int c : classes)
        if (nv <= c) return c;
    return 0;
}

// Which search the rows kernels run: the histogram form (default) or the bisection form of
// rounds 2-3 (ECOFLAP_WANDA_ROWS_SEARCH=bisect: A/B measurements and the parity tests, which run
// both against the oracle).  Read per call: cheap, and a test can flip it inside one process.
static inline bool rows_search_hist() {
    const char* e = getenv("ECOFLAP_WANDA_ROWS_SEARCH");
    return !(e && e[0] == 'b');
}

// every row of the group exactly 64 * nv vectors wide (no padding lanes)?
template <int DT>
static inline bool rows_group_full(const RowsGroup& g, int nv) {
    for (int i = 0; i < g.n; ++i)
        if (g.cols[i] != (int64_t)64 * nv * Vec<DT>::N) return false;
    return true;
}

template <int DT, bool HIST, bool FULL>
static void launch_rows_wave_t(const RowsGroup& g, int nv, hipStream_t s) {
    const dim3 grid((unsigned)((g.start[g.n] + 3) / 4)), blk(256);
    if (nv == 1) hipLaunchKernelGGL((wanda_rows_wave_kernel<DT, 1, HIST, FULL>), grid, blk, 0, s, g);
    else if (nv == 2) hipLaunchKernelGGL((wanda_rows_wave_kernel<DT, 2, HIST, FULL>), grid, blk, 0, s, g);
    else hipLaunchKernelGGL((wanda_rows_wave_kernel<DT, 4, HIST, FULL>), grid, blk, 0, s, g);
}
template <int DT>
static void launch_rows_wave(const RowsGroup& g, int nv, hipStream_t s) {
    const bool full = rows_group_full<DT>(g, nv);
    if (rows_search_hist()) {
        if (full) launch_rows_wave_t<DT, true, true>(g, nv, s);
        else launch_rows_wave_t<DT, true, false>(g, nv, s);
    } else {
        launch_rows_wave_t<DT, false, false>(g, nv, s);
    }
}

template <int DT, bool HIST, bool FULL>
static void launch_rows_fused_t(const RowsGroup& gw, int nvw, const RowsGroup& gr, int nvr, hipStream_t s) {
    const dim3 grid((unsigned)(gr.start[gr.n] + (gw.start[gw.n] + 3) / 4)), blk(256);
#define FUSED(W_, R_) hipLaunchKernelGGL((wanda_rows_fused_kernel<DT, W_, R_, HIST, FULL>), grid, blk, 0, s, gw, gr)
#define FUSED_R(W_)                                                          \
    do {                                                                     \
        if (nvr == 1) FUSED(W_, 1); else if (nvr == 2) FUSED(W_, 2);         \
        else if (nvr == 3) FUSED(W_, 3); else FUSED(W_, 4);                  \
    } while (0)
    if (nvw == 1) FUSED_R(1); else if (nvw == 2) FUSED_R(2); else FUSED_R(4);
#undef FUSED_R
#undef FUSED
}
template <int DT>
static void launch_rows_fused(const RowsGroup& gw, int nvw, const RowsGroup& gr, int nvr, hipStream_t s) {
    const bool full = rows_group_full<DT>(gw, nvw);
    if (rows_search_hist()) {
        if (full) launch_rows_fused_t<DT, true, true>(gw, nvw, gr, nvr, s);
        else launch_rows_fused_t<DT, true, false>(gw, nvw, gr, nvr, s);
    } else {
        launch_rows_fused_t<DT, false, false>(gw, nvw, gr, nvr, s);
    }
}

template <int DT, bool HIST>
static void launch_rows_reg_t(const RowsGroup& g, int nv, hipStream_t s) {
    const dim3 grid((unsigned)g.start[g.n]), blk(256);
#define ROWS_REG(NV_) hipLaunchKernelGGL((wanda_rows_reg_kernel<DT, NV_, HIST>), grid, blk, 0, s, g)
    switch (nv) {
        case 1: ROWS_REG(1); break;
        case 2: ROWS_REG(2); break;
        case 3: ROWS_REG(3); break;
        case 4: ROWS_REG(4); break;
        case 6: ROWS_REG(6); break;
        case 8: ROWS_REG(8); break;
        case 12: ROWS_REG(12); break;
        default: ROWS_REG(16); break;
    }
#undef ROWS_REG
}
template <int DT>
static void launch_rows_reg(const RowsGroup& g, int nv, hipStream_t s) {
    if (rows_search_hist()) launch_rows_reg_t<DT, true>(g, nv, s);
    else launch_rows_reg_t<DT, false>(g, nv, s);
}

// =====================================================================================
// K7 matrix mode: global (k+1)-th smallest over rows*cols by 3 histogram passes
// (11 + 11 + 10 bits), then zero metric <= threshold.
// =====================================================================================
struct MatrixSelState {
    uint32_t hist[3][2048];
    uint32_t resolved[3][2];      // (prefix, remaining rank) after pass 1, 2, 3
    uint32_t pad[58];
};

// Matrix mode runs FEW, LARGE workgroups (1024 threads, 256 over the launch, shared out by size):
// every workgroup merges its LDS histogram into the matrix's global one with one atomic per
// non-empty bin, and those atomics all meet at the memory side — with 1024 workgroups per matrix
// that merge, not the read of W, was the whole cost of a pass (measured: 74 us for a 50 MB ViT-g
// block).  What a pass costs today is mostly fixed: a matrix of 2 M elements takes ~10 us per
// pass like one of 8 M (launch, zeroing + staging LDS, resolving the earlier passes from the
// global histograms, the merge); the streaming itself runs at > 5 TB/s.
#define WM_THREADS 1024
#define WM_WAVES (WM_THREADS / 64)
#define WM_TOTAL_WGS 256     // one workgroup per CU over the whole launch (measured: 128 and 512 both slower)
#define WM_SUB0 4            // interleaved counters per bin in the first pass
#define WM_UNROLL 4
#define WM_SQ_LDS 16384      // columns whose sqrt table is staged in LDS (64 KiB of the CU's 160)

// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() is a release / acquire fence on
// every address space, which the compiler implements as s_waitcnt vmcnt(0): in a kernel that has
// its share of W in flight in registers, EVERY barrier of the prologue then waits for the whole
// stream (measured: 12 us to the first barrier instead of 2).  Here: lgkmcnt(0) (LDS and scalar
// loads; vmcnt 63 / expcnt 7 = do not wait) and s_barrier; the "memory" clobber keeps the compiler
// from moving accesses across it.  Threads that exchange data through GLOBAL memory inside one
// kernel must not use this.
static __device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
template <bool LDS_ONLY>
static __device__ __forceinline__ void wm_barrier() {
    if (LDS_ONLY) lds_barrier();
    else __syncthreads();
}

// sqrt(scaler_row) of the workgroup's matrix into LDS (every vector needs 2 x 16 bytes of it:
// from global memory those were two dependent cache round trips per vector, most of a pass)
static __device__ __forceinline__ const float* stage_sq(const float* __restrict__ sq, int64_t cols,
                                                        float* lds) {
    if (cols > WM_SQ_LDS) return sq;
    for (int64_t c = threadIdx.x; c < cols; c += WM_THREADS) lds[c] = sq[c];
    __syncthreads();
    return lds;
}
// the same, straight from the statistic (sampled-bracket path: no sqrt launch in front of it)
static __device__ __forceinline__ const float* stage_sqrt(const float* __restrict__ sr, int64_t cols,
                                                          float* lds) {
    for (int64_t c = threadIdx.x; c < cols; c += WM_THREADS) lds[c] = __builtin_sqrtf(sr[c]);
    __syncthreads();
    return lds;
}
// the same in two halves, so that the statistic's loads can be issued BEFORE a long run of other
// loads (loads return in order: issued after them, these would wait for all of them) and consumed
// after it: the first WM_SR_PRE * WM_THREADS columns travel through registers
#define WM_SR_PRE 8
struct SqrtPre { float v[WM_SR_PRE]; };
static __device__ __forceinline__ SqrtPre stage_sqrt_issue(const float* __restrict__ sr, int64_t cols) {
    SqrtPre p;
#pragma unroll
    for (int q = 0; q < WM_SR_PRE; ++q) {
        // (index clamped, not predicated: a predicated load becomes a branch, and the compiler
        // then sinks the wait and the sqrt into it — eight serial round trips)
        const int64_t c = threadIdx.x + (int64_t)q * WM_THREADS;
        p.v[q] = sr[c < cols ? c : cols - 1];
    }
    return p;
}
static __device__ __forceinline__ const float* stage_sqrt_finish(const SqrtPre& p, const float* __restrict__ sr,
                                                                 int64_t cols, float* lds) {
#pragma unroll
    for (int q = 0; q < WM_SR_PRE; ++q) {
        const int64_t c = threadIdx.x + (int64_t)q * WM_THREADS;
        if (c < cols) lds[c] = __builtin_sqrtf(p.v[q]);
    }
    for (int64_t c = threadIdx.x + (int64_t)WM_SR_PRE * WM_THREADS; c < cols; c += WM_THREADS)
        lds[c] = __builtin_sqrtf(sr[c]);
    lds_barrier();
    return lds;
}
static __device__ __forceinline__ u32x4 ld_sq4(const float* sq, int64_t i4) {
    return *(const u32x4*)(sq + 4 * i4);      // LDS or global, 16-byte aligned either way
}

// inclusive scan of one value per thread over the 1024-thread block; returns (inclusive, total)
template <bool LDS_ONLY = false>
static __device__ __forceinline__ uint32_t block_scan_wm(uint32_t v, uint32_t* lds_waves,
                                                         uint32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    wm_barrier<LDS_ONLY>();  // protect lds_waves reuse
    if (lane == 63) lds_waves[wave] = x;
    wm_barrier<LDS_ONLY>();
    uint32_t base = 0, t = 0;
#pragma unroll
    for (int k = 0; k < WM_WAVES; ++k) {
        const uint32_t wk = lds_waves[k];
        if (k < wave) base += wk;
        t += wk;
    }
    total = t;
    return x + base;
}

// Every workgroup resolves the LATEST pass itself from its global histogram (8 KB, L2-resident):
// bin = first bin whose inclusive count reaches `remaining`.  Identical in every workgroup, so no
// separate "pick" launch and no inter-workgroup hand-off is needed; the result (prefix, remaining)
// is left in the state by the matrix's first workgroup, so the next pass starts from it instead
// of resolving all earlier passes again.  Two adjacent bins per thread: one block scan per pick.
struct PickLoad {                // the pick's global loads, issued before anything that can wait on memory
    uint32_t c0, c1, prefix, remaining;
};
static __device__ __forceinline__ PickLoad pick_load(const MatrixSelState* st, int upto, uint32_t rank0) {
    PickLoad p;
    p.c0 = p.c1 = 0; p.prefix = 0; p.remaining = rank0;
    if (upto == 0) return p;
    const int bins = upto == 3 ? 1024 : 2048;
    const int b0 = 2 * (int)threadIdx.x;
    if (b0 < bins) {
        const uint2 two = *(const uint2*)(st->hist[upto - 1] + b0);
        p.c0 = two.x; p.c1 = two.y;
    }
    if (upto >= 2) {
        p.prefix = st->resolved[upto - 2][0];
        p.remaining = st->resolved[upto - 2][1];
    }
    return p;
}

// prefix (selected high bits) and remaining rank after `upto` resolved passes
static __device__ __forceinline__ void resolve(MatrixSelState* st, int upto, const PickLoad& p, bool writer,
                                               uint32_t* wave4, uint32_t* out2, uint32_t& prefix,
                                               uint32_t& remaining) {
    prefix = p.prefix;
    remaining = p.remaining;
    if (upto == 0) return;
    uint32_t total;
    const uint32_t incl = block_scan_wm(p.c0 + p.c1, wave4, total);
    const uint32_t excl = incl - (p.c0 + p.c1);
    if (excl < remaining && remaining <= incl) {
        const bool second = remaining > excl + p.c0;
        out2[0] = (uint32_t)(2 * threadIdx.x + (second ? 1 : 0));
        out2[1] = remaining - excl - (second ? p.c0 : 0u);
    }
    __syncthreads();
    const int shift = upto == 1 ? 21 : (upto == 2 ? 10 : 0);
    prefix |= out2[0] << shift;
    remaining = out2[1];
    __syncthreads();
    if (writer && threadIdx.x == 0) {
        st->resolved[upto - 1][0] = prefix;
        st->resolved[upto - 1][1] = remaining;
    }
}

struct MatGroup {
    int n;
    int32_t start[WMAX + 1];      // workgroup prefix over the group's matrices
    void* w[WMAX];
    const float* sq[WMAX];
    int64_t rows[WMAX];
    int64_t cols[WMAX];
    uint32_t rank0[WMAX];
    MatrixSelState* st[WMAX];
    uint8_t* mask[WMAX];
};

template <int DT, int PASS, bool VECTOR, int SUB = 1>
__global__ __launch_bounds__(WM_THREADS) void wanda_matrix_hist_kernel(const MatGroup g) {
    const int it = group_item(g, blockIdx.x);
    const unsigned lb = blockIdx.x - g.start[it], nb = g.start[it + 1] - g.start[it];
    const void* __restrict__ w = g.w[it];
    const float* sq = g.sq[it];
    const int64_t rows = g.rows[it], cols = g.cols[it];
    const uint32_t rank0 = g.rank0[it];
    MatrixSelState* st = g.st[it];
    constexpr int SHIFT = PASS == 0 ? 21 : (PASS == 1 ? 10 : 0);
    constexpr int BITS = PASS == 2 ? 10 : 11;
    constexpr uint32_t HI_MASK = PASS == 0 ? 0u : (PASS == 1 ? 0xffe00000u : 0xfffffc00u);
    constexpr int N = Vec<DT>::N;
    // SUB > 1: SUB interleaved counters per bin, chosen by the lane — the top digit has few
    // distinct values, and lanes of one LDS atomic that meet at one address are served one by one
    __shared__ uint32_t h[2048 * SUB];
    __shared__ uint32_t wave4[WM_WAVES];
    __shared__ uint32_t out2[2];
    const PickLoad pl = pick_load(st, PASS, rank0);   // in flight while the LDS is set up
    for (int i = threadIdx.x; i < 2048 * SUB; i += WM_THREADS) h[i] = 0;
    const uint32_t sub = SUB > 1 ? (threadIdx.x & (SUB - 1)) : 0u;
    __shared__ __attribute__((aligned(16))) float sq_lds[VECTOR ? WM_SQ_LDS : 4];
    if (VECTOR) sq = stage_sq(sq, cols, sq_lds);
    uint32_t prefix, remaining;
    resolve(st, PASS, pl, lb == 0, wave4, out2, prefix, remaining);
    __syncthreads();
    if (VECTOR) {
        const int64_t vpr = cols / N;                 // vectors per row
        const int64_t nvec = rows * vpr;
        // WM_UNROLL independent 16-byte loads in flight per thread (one per stride of the
        // workgroup set): with a single load per iteration the pass ran at the latency bound
        const int64_t stride = (int64_t)nb * WM_THREADS;
        // column of a vector without a 64-bit modulo per vector (that division was most of the
        // pass's instructions): one 32-bit modulo up front, then add-and-wrap per stride
        const uint32_t vpr32 = (uint32_t)vpr, step32 = (uint32_t)(stride % vpr);
        uint32_t cv = (uint32_t)(((int64_t)lb * WM_THREADS + threadIdx.x) % vpr);
        for (int64_t v0 = (int64_t)lb * WM_THREADS + threadIdx.x; v0 < nvec; v0 += stride * WM_UNROLL) {
            u32x4 wv[WM_UNROLL];
#pragma unroll
            for (int j = 0; j < WM_UNROLL; ++j)
                if (v0 + j * stride < nvec) wv[j] = ld16(w, v0 + j * stride);
#pragma unroll
            for (int j = 0; j < WM_UNROLL; ++j) {
                const int64_t v = v0 + j * stride;
                const int64_t c0 = (int64_t)cv * N;
                cv += step32;
                if (cv >= vpr32) cv -= vpr32;
                if (v >= nvec) continue;
                float f[N];
                Vec<DT>::unpack(wv[j], f);
#pragma unroll
                for (int q = 0; q < N / 4; ++q) {
                    const u32x4 s4 = ld_sq4(sq, c0 / 4 + q);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const uint32_t b = __float_as_uint(__builtin_fabsf(f[4 * q + i]) * __uint_as_float(s4[i]));
                        if ((b & HI_MASK) == prefix) atomicAdd(&h[((b >> SHIFT) & ((1u << BITS) - 1u)) * SUB + sub], 1u);
                    }
                }
            }
        }
    } else {
        for (int64_t r = lb; r < rows; r += nb)
            for (int64_t c = threadIdx.x; c < cols; c += WM_THREADS) {
                const uint32_t b = metric_bits<DT>(w, r * cols + c, sq[c]);
                if ((b & HI_MASK) == prefix) atomicAdd(&h[((b >> SHIFT) & ((1u << BITS) - 1u)) * SUB + sub], 1u);
            }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (1 << BITS); i += WM_THREADS) {
        uint32_t c = 0;
#pragma unroll
        for (int q = 0; q < SUB; ++q) c += h[i * SUB + q];
        if (c) atomicAdd(&st->hist[PASS][i], c);
    }
}

template <int DT, bool VECTOR>
__global__ __launch_bounds__(WM_THREADS) void wanda_matrix_apply_kernel(const MatGroup g) {
    const int it = group_item(g, blockIdx.x);
    const unsigned lb = blockIdx.x - g.start[it], nb = g.start[it + 1] - g.start[it];
    void* w = g.w[it];
    const float* sq = g.sq[it];
    const int64_t rows = g.rows[it], cols = g.cols[it];
    const uint32_t rank0 = g.rank0[it];
    MatrixSelState* st = g.st[it];
    uint8_t* mask_out = g.mask[it];
    constexpr int N = Vec<DT>::N;
    __shared__ uint32_t wave4[WM_WAVES];
    __shared__ uint32_t out2[2];
    const PickLoad pl = pick_load(st, 3, rank0);
    __shared__ __attribute__((aligned(16))) float sq_lds[VECTOR ? WM_SQ_LDS : 4];
    if (VECTOR) sq = stage_sq(sq, cols, sq_lds);
    uint32_t thres_bits, remaining;
    resolve(st, 3, pl, false, wave4, out2, thres_bits, remaining);
    const float thres = __uint_as_float(thres_bits);
    if (VECTOR) {
        const int64_t vpr = cols / N;
        const int64_t nvec = rows * vpr;
        const int64_t stride = (int64_t)nb * WM_THREADS;
        const uint32_t vpr32 = (uint32_t)vpr, step32 = (uint32_t)(stride % vpr);
        uint32_t cv = (uint32_t)(((int64_t)lb * WM_THREADS + threadIdx.x) % vpr);
        for (int64_t v0 = (int64_t)lb * WM_THREADS + threadIdx.x; v0 < nvec; v0 += stride * WM_UNROLL) {
            u32x4 wv[WM_UNROLL];
#pragma unroll
            for (int j = 0; j < WM_UNROLL; ++j)
                if (v0 + j * stride < nvec) wv[j] = ld16(w, v0 + j * stride);
#pragma unroll
            for (int j = 0; j < WM_UNROLL; ++j) {
                const int64_t v = v0 + j * stride;
                const int64_t c0 = (int64_t)cv * N;
                cv += step32;
                if (cv >= vpr32) cv -= vpr32;
                if (v >= nvec) continue;
                float f[N];
                Vec<DT>::unpack(wv[j], f);
                uint32_t lo = 0, hi = 0;
                bool any = false;
#pragma unroll
                for (int q = 0; q < N / 4; ++q) {
                    const u32x4 s4 = ld_sq4(sq, c0 / 4 + q);
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
        }
    } else {
        for (int64_t r = lb; r < rows; r += nb)
            for (int64_t c = threadIdx.x; c < cols; c += WM_THREADS) {
                const int64_t i = r * cols + c;
                const float mval = __uint_as_float(metric_bits<DT>(w, i, sq[c]));
                const bool prune = mval <= thres;
                if (prune) Vec<DT>::store1(w, i, 0.0f);
                if (mask_out) mask_out[i] = prune ? 1 : 0;
            }
    }
}

// =====================================================================================
// K7 matrix mode, sampled bracket: 2 reads + 1 write of W instead of 4 + 1, in TWO dependent launches
// =====================================================================================
// The three histogram passes above read W three times to find ONE number.  Here a sample of
// WS_SAMPLE_VECS vectors per matrix (pseudo-randomly placed, one per stride) brackets the (k+1)-th
// smallest metric between two sample order statistics 5 sigma either side of its expected rank
// (~4 % of the elements); ONE full pass counts exactly what lies below the bracket and histograms
// what lies inside it (512 bins: an LDS atomic for 4 % of the elements instead of all of them);
// the apply pass then knows the bin that holds the threshold, zeroes everything below that bin,
// keeps everything above it and appends the bin's own few hundred elements (index, metric bits)
// to a list; the last workgroup to finish selects the threshold among them exactly and settles
// them.  Every count is exact, so the result is the reference's (W:555-558) bit for bit; when the
// bracket misses (probability ~1e-6 per matrix), the bin is too crowded for the list (massive
// ties) or the threshold is not finite, every workgroup of the matrix sees that from the counts
// before it has written anything, and the matrix's first workgroup runs the three-histogram
// selection on it by itself (matrix_fallback_body).
//
// Round 5, after a phase clock on the device (tools/diag/k7_clock.py: sample 14.9 us alone on 4
// workgroups, 1.6 us to the next launch, bracket pass 19 us of which 11 streaming, apply pass 21 us
// of which 4.2 before its first load, 5 us of threshold settling at the end): the sample no longer
// has a launch of its own.  EVERY workgroup of the counting pass takes the sample of its matrix
// itself (2048 vectors, L2 hits after the first workgroup; the result is a pure function of W and
// the statistic, so all workgroups hold the same bracket and nobody waits for anybody — no flag,
// no spin, nothing that could deadlock against another process's launch) while the WS_PRE vectors
// per thread that are its share of the matrix are already on their way into registers; counts go
// to a slot per workgroup (no zeroed state, no atomics meeting at the memory side) which the
// apply pass sums in its prologue, again with its share of W already in flight.
#define WS_SAMPLE_VECS 2048
#define WS_BINS 512
#define WS_CAP 16384                    // list of the threshold bin's elements (LDS of the settling workgroup)
// Matrices of more elements than this keep the three-histogram selection: the bracket holds up to
// ~8.5 % of a matrix (5 sigma of the sample plus a sample bin either side) in 256-512 histogram
// bins, so the threshold's bin of a larger matrix would outgrow the list and every such call
// would end in the one-workgroup fallback.
#define WS_MAX_NUMEL (48ll << 20)
#define WS_PRE 4                        // vectors per thread in registers across the prologue (see below)
#define WS_PART_WORDS (WS_BINS + 8)     // a workgroup's slot: histogram, count below the bracket

struct BracketState {
    uint32_t list_count, ticket, fallback, pad;     // cleared by the counting pass
    uint32_t lo, hi, shift, valid;                  // written by the counting pass
    uint32_t list_idx[WS_CAP];
    uint32_t list_bits[WS_CAP];
};

struct BracketGroup {
    MatGroup m;
    BracketState* bs[WMAX];
    const float* sr[WMAX];          // the raw statistic (sqrt taken while staging it into LDS)
};

// rank (1-based) -> (bin, rank inside the bin) over an LDS histogram of 2048 bins, WM_THREADS threads
static __device__ __forceinline__ void find_rank_wm(const uint32_t* h, uint32_t rank, uint32_t* waves,
                                                    uint32_t* out2) {
    const uint32_t c0 = h[2 * threadIdx.x], c1 = h[2 * threadIdx.x + 1];
    uint32_t total;
    const uint32_t incl = block_scan_wm(c0 + c1, waves, total);
    const uint32_t excl = incl - (c0 + c1);
    if (excl < rank && rank <= incl) {
        const bool second = rank > excl + c0;
        out2[0] = 2 * threadIdx.x + (second ? 1u : 0u);
        out2[1] = rank - excl - (second ? c0 : 0u);
    }
    __syncthreads();
}
// the counting pass: sample -> bracket [lo, hi) (every workgroup, identically), then the exact
// count below the bracket and the WS_BINS-bin histogram inside it over the workgroup's share of W.
// The sample's two order statistics are taken at the resolution of ONE histogram — 4096 bins of
// 1/16 octave (metric bits >> 19): the bracket's edges are bin edges, up to 4 % of the elements wider
// on either side than the exact order statistics would make it, which costs the exact pass a few
// more LDS atomics and saves the prologue a second histogram level (two more scans, five barriers:
// 3.5 us of every workgroup's time by the phase clock).  Nothing about exactness depends on where
// the edges lie.
#define WS_SBINS 4096
#define WS_SSUB 4                      // interleaved counters per sample bin (few dozen live bins)
template <int DT>
__global__ __launch_bounds__(WM_THREADS) void wanda_matrix_sbracket_kernel(const BracketGroup bg,
                                                                           uint32_t* __restrict__ part) {
    const MatGroup& g = bg.m;
    const int it = group_item(g, blockIdx.x);
    const unsigned lb = blockIdx.x - g.start[it], nb = g.start[it + 1] - g.start[it];
    const void* __restrict__ w = g.w[it];
    const int64_t rows = g.rows[it], cols = g.cols[it];
    BracketState* bs = bg.bs[it];
    constexpr int N = Vec<DT>::N;
    constexpr int S = WS_SAMPLE_VECS * N;
    constexpr int VPT = WS_SAMPLE_VECS / WM_THREADS;          // sample vectors per thread
    __shared__ uint32_t h[WS_BINS];
    __shared__ uint32_t red[WM_WAVES];
    __shared__ __attribute__((aligned(16))) float sq_lds[WM_SQ_LDS];
    __shared__ __attribute__((aligned(16))) uint32_t h1s[WS_SBINS * WS_SSUB];
    __shared__ uint32_t waves[WM_WAVES], o_lo[2], o_hi[2], brk[4];
    K7_ENTER(1); K7_STAMP(1, 0);
    const int64_t vpr = cols / N, nvec = rows * vpr;
    const uint32_t vpr32 = (uint32_t)vpr;
    // ---- the prologue's global loads, issued before anything waits --------------------------
    const SqrtPre srp = stage_sqrt_issue(bg.sr[it], cols);
    // the sample: one vector per stride of the matrix, at a pseudo-random place inside its stride
    // (a regular stride can be a multiple of the row length and then sees one column group only)
    const uint32_t step = (uint32_t)(nvec / WS_SAMPLE_VECS);
    u32x4 sv[VPT];
    uint32_t sc0[VPT];
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const uint32_t i = threadIdx.x + WM_THREADS * j;
        const uint32_t jitter = (uint32_t)(((uint64_t)(i * 2654435761u) * (uint64_t)step) >> 32);
        const uint32_t v = i * step + jitter;
        sc0[j] = (v % vpr32) * N;
        sv[j] = ld16(w, (int64_t)v);
    }
    // A compute unit serves its waves' loads in the order they arrive: every wave's prologue loads
    // are queued before any wave's W loads
    lds_barrier();
    // The first WS_PRE vectors per thread of the workgroup's share of W, in registers across the
    // sample.  Not more: a wave cannot run ahead of its own loads by more than the memory pipeline
    // holds — with the whole share (12 vectors) issued up front the waves sat in the ISSUE of those
    // loads until most of the stream had arrived, and the sample waited with them.
    const uint32_t nvec32 = (uint32_t)nvec, stride32 = nb * WM_THREADS;     // (numel < 2^32)
    const uint32_t vfirst32 = lb * WM_THREADS + threadIdx.x;
    u32x4 cur[WS_PRE];
#pragma unroll
    for (int j = 0; j < WS_PRE; ++j) {
        const uint32_t v = vfirst32 + j * stride32;
        cur[j] = ld16(w, (int64_t)(v < nvec32 ? v : nvec32 - 1u));  // (clamped: no branch around a load)
    }
    for (int i = threadIdx.x; i < WS_BINS; i += WM_THREADS) h[i] = 0;
    for (int i = threadIdx.x; i < WS_SBINS * WS_SSUB; i += WM_THREADS) h1s[i] = 0;
    const float* sq = stage_sqrt_finish(srp, bg.sr[it], cols, sq_lds);      // (ends with a barrier)
    K7_STAMP(1, 1);
    // ---- the sample's histogram -------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        float f[N];
        Vec<DT>::unpack(sv[j], f);
#pragma unroll
        for (int e = 0; e < N; ++e) {
            const uint32_t b = __float_as_uint(__builtin_fabsf(f[e]) * sq[sc0[j] + e]);
            const uint32_t top = b >> 19;                       // (sign-bit patterns: the last bin)
            atomicAdd(&h1s[(top < (uint32_t)WS_SBINS ? top : (uint32_t)WS_SBINS - 1u) * WS_SSUB + (threadIdx.x & (WS_SSUB - 1))], 1u);
        }
    }
    lds_barrier();
    K7_STAMP(1, 2);
    // sample ranks 5 sigma either side of the expected one; four bins per thread, one scan
    const double numel = (double)rows * (double)cols;
    const double p = (double)g.rank0[it] / numel;
    const double mid = p * S;
    const double dev = 5.0 * __builtin_sqrt((double)S * p * (1.0 - p)) + 8.0;
    const bool open_lo = mid - dev < 1.0, open_hi = mid + dev > (double)S;
    const uint32_t r_lo = open_lo ? 1u : (uint32_t)(mid - dev);
    const uint32_t r_hi = open_hi ? (uint32_t)S : (uint32_t)(mid + dev);
    {
        uint32_t c4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const u32x4 x = *(const u32x4*)(h1s + (4 * threadIdx.x + k) * WS_SSUB);
            c4[k] = x[0] + x[1] + x[2] + x[3];
        }
        const uint32_t csum = c4[0] + c4[1] + c4[2] + c4[3];
        uint32_t total;
        const uint32_t incl = block_scan_wm<true>(csum, waves, total);
        const uint32_t excl = incl - csum;
        if (excl < r_lo && r_lo <= incl) {
            uint32_t e = excl; int d = 0; bool on = true;      // first bin whose running count reaches the rank
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                on = on && r_lo > e + c4[k];
                if (on) { e += c4[k]; d = k + 1; }
            }
            o_lo[0] = 4 * threadIdx.x + (uint32_t)d;
        }
        if (excl < r_hi && r_hi <= incl) {
            uint32_t e = excl; int d = 0; bool on = true;      // first bin whose running count reaches the rank
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                on = on && r_hi > e + c4[k];
                if (on) { e += c4[k]; d = k + 1; }
            }
            o_hi[0] = 4 * threadIdx.x + (uint32_t)d;
        }
    }
    lds_barrier();
    if (threadIdx.x == 0) {
        const uint32_t lo = open_lo ? 0u : (o_lo[0] << 19);
        const uint64_t hi64 = open_hi ? 0x100000000ull : ((uint64_t)(o_hi[0] + 1u) << 19);
        const uint32_t hi = hi64 > 0xffffffffull ? 0xffffffffu : (uint32_t)hi64;
        const uint32_t width = hi - lo;
        int shift = 0;
        while (shift < 31 && ((width - 1u) >> shift) >= (uint32_t)WS_BINS) ++shift;
        brk[0] = lo; brk[1] = hi; brk[2] = (uint32_t)shift; brk[3] = hi > lo ? 1u : 0u;
        if (lb == 0) {
            bs->list_count = 0u; bs->ticket = 0u; bs->fallback = 0u;
            bs->lo = lo; bs->hi = hi; bs->shift = (uint32_t)shift; bs->valid = hi > lo ? 1u : 0u;
        }
    }
    lds_barrier();
    K7_STAMP(1, 3);
    const uint32_t lo = brk[0], width = brk[1] - brk[0], shift = brk[2];
    // ---- the exact counts over the share of W: the next WS_PRE loads in flight while the ---------
    // ---- current ones are counted                                                       ---------
    uint32_t below = 0;
    if (brk[3]) {
        const uint32_t step32 = stride32 % vpr32;
        uint32_t cv = vfirst32 % vpr32;
        const uint32_t wg_first = lb * WM_THREADS;              // (uniform: the loop's trip count)
        for (uint32_t base = 0; wg_first + base < nvec32; base += WS_PRE * stride32) {
            u32x4 nxt[WS_PRE];
            const bool more = wg_first + base + WS_PRE * stride32 < nvec32;
            if (more) {
#pragma unroll
                for (int j = 0; j < WS_PRE; ++j) {
                    const uint32_t v = vfirst32 + base + (WS_PRE + j) * stride32;
                    nxt[j] = ld16(w, (int64_t)(v < nvec32 ? v : nvec32 - 1u));
                }
            }
#pragma unroll
            for (int j = 0; j < WS_PRE; ++j) {
                const uint32_t c0 = cv * N;
                cv += step32;
                if (cv >= vpr32) cv -= vpr32;
                if (vfirst32 + base + j * stride32 < nvec32) {
                    float f[N];
                    Vec<DT>::unpack(cur[j], f);
#pragma unroll
                    for (int q = 0; q < N / 4; ++q) {
                        const u32x4 s4 = ld_sq4(sq, c0 / 4 + q);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const uint32_t b = __float_as_uint(__builtin_fabsf(f[4 * q + i]) * __uint_as_float(s4[i]));
                            const uint32_t rel = b - lo;
                            below += b < lo ? 1u : 0u;
                            if (b >= lo && rel < width) atomicAdd(&h[rel >> shift], 1u);
                        }
                    }
                }
            }
            if (more) {
#pragma unroll
                for (int j = 0; j < WS_PRE; ++j) cur[j] = nxt[j];
            }
        }
    }
    K7_STAMP(1, 4);
    // the workgroup's slot: its histogram and its count below the bracket
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) below += __shfl_xor(below, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = below;
    lds_barrier();
    uint32_t* slot = part + (size_t)blockIdx.x * WS_PART_WORDS;
    if (threadIdx.x < WS_BINS) slot[threadIdx.x] = h[threadIdx.x];
    if (threadIdx.x == WS_BINS) {
        uint32_t t = 0;
#pragma unroll
        for (int k = 0; k < WM_WAVES; ++k) t += red[k];
        slot[WS_BINS] = t;
    }
    K7_STAMP(1, 5);
    K7_EXIT(1);
}

// The exact finish for a matrix the sampled passes cannot settle (the bracket missed, ~1e-6 per
// matrix; the threshold's bin reaches non-finite bit patterns; the bin is too crowded for the
// list: massive ties), ON THE DEVICE and inside the apply pass: every one of these conditions is
// known from the counts in the prologue, to every workgroup of the matrix alike, BEFORE anything
// of W has been written — so all of them leave, and the matrix's first workgroup runs the whole
// three-histogram selection and the apply pass by itself (LDS histograms, no other workgroup to
// wait for: no grid barrier, nothing that could deadlock with another resident instance, and no
// question of one XCD seeing another's stores).  Slow (one CU streams the matrix four times:
// ~1 ms for 8.6 M elements) and rare.  Rounds 3-4 decided this on the host (a flag read-back and a
// stream round trip per call), round 5 first by a conditional launch of its own (4 us at the end
// of every call for a launch that almost never had anything to do).
// How often it ran, by reason (1: bracket miss / non-finite bin, 2: crowded bin, 3: a list that
// does not match its count — cannot happen, counted so that a test would see it): read and cleared
// by ecoflap_wanda_fallback_counts (tests: ordinary data must not take it).
__device__ unsigned int eco_k7_fallbacks[4];
extern "C" int ecoflap_wanda_fallback_counts(unsigned int* out4, int reset) {
    if (out4) {
        const hipError_t e = hipMemcpyFromSymbol(out4, HIP_SYMBOL(eco_k7_fallbacks), sizeof(unsigned int) * 4);
        if (e != hipSuccess) return (int)e;
    }
    if (reset) {
        const unsigned int z[4] = {0u, 0u, 0u, 0u};
        const hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(eco_k7_fallbacks), z, sizeof(z));
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

template <int DT, int PASS>
static __device__ __forceinline__ void fallback_hist_pass(const void* __restrict__ w, const float* sq,
                                                          int64_t nvec, uint32_t vpr32, uint32_t* h,
                                                          uint32_t prefix) {
    constexpr int N = Vec<DT>::N;
    constexpr int SHIFT = PASS == 0 ? 21 : (PASS == 1 ? 10 : 0);
    constexpr uint32_t DMASK = PASS == 2 ? 1023u : 2047u;
    constexpr uint32_t HI_MASK = PASS == 0 ? 0u : (PASS == 1 ? 0xffe00000u : 0xfffffc00u);
    for (int i = threadIdx.x; i < 2048; i += WM_THREADS) h[i] = 0;
    __syncthreads();
    uint32_t cv = threadIdx.x % vpr32;
    const uint32_t step32 = WM_THREADS % vpr32;
    for (int64_t v = threadIdx.x; v < nvec; v += WM_THREADS) {
        const int64_t c0 = (int64_t)cv * N;
        cv += step32;
        if (cv >= vpr32) cv -= vpr32;
        float f[N];
        Vec<DT>::unpack(ld16(w, v), f);
#pragma unroll
        for (int e = 0; e < N; ++e) {
            const uint32_t b = __float_as_uint(__builtin_fabsf(f[e]) * sq[c0 + e]);
            if ((b & HI_MASK) == prefix) atomicAdd(&h[(b >> SHIFT) & DMASK], 1u);
        }
    }
    __syncthreads();
}

// one workgroup, the whole matrix: h = 2048 words of LDS, sq = the staged sqrt table
template <int DT>
static __device__ void matrix_fallback_body(void* w, const float* sq, int64_t rows, int64_t cols, uint32_t rank0,
                                            uint8_t* mask_out, uint32_t* h, uint32_t* wave4, uint32_t* out2,
                                            unsigned reason) {
    if (threadIdx.x == 0) atomicAdd(&eco_k7_fallbacks[reason & 3u], 1u);
    constexpr int N = Vec<DT>::N;
    const int64_t vpr = cols / N, nvec = rows * vpr;
    const uint32_t vpr32 = (uint32_t)vpr;
    uint32_t prefix = 0, remaining = rank0;
    __syncthreads();
    fallback_hist_pass<DT, 0>(w, sq, nvec, vpr32, h, prefix);
    find_rank_wm(h, remaining, wave4, out2);
    prefix |= out2[0] << 21; remaining = out2[1];
    __syncthreads();
    fallback_hist_pass<DT, 1>(w, sq, nvec, vpr32, h, prefix);
    find_rank_wm(h, remaining, wave4, out2);
    prefix |= out2[0] << 10; remaining = out2[1];
    __syncthreads();
    fallback_hist_pass<DT, 2>(w, sq, nvec, vpr32, h, prefix);
    find_rank_wm(h, remaining, wave4, out2);
    prefix |= out2[0];
    __syncthreads();
    const float thres = __uint_as_float(prefix);     // sorted[k]
    uint32_t cv = threadIdx.x % vpr32;
    const uint32_t step32 = WM_THREADS % vpr32;
    for (int64_t v = threadIdx.x; v < nvec; v += WM_THREADS) {
        const int64_t c0 = (int64_t)cv * N;
        cv += step32;
        if (cv >= vpr32) cv -= vpr32;
        float f[N];
        Vec<DT>::unpack(ld16(w, v), f);
        uint32_t lo = 0, hi = 0;
        bool any = false;
#pragma unroll
        for (int e = 0; e < N; ++e) {
            // W_metric <= thres (W:556): false for NaN metrics, as in torch
            const bool prune = (__builtin_fabsf(f[e]) * sq[c0 + e]) <= thres;
            if (prune) { f[e] = 0.0f; any = true; }
            if (e < 4) lo |= (prune ? 1u : 0u) << (8 * e);
            else hi |= (prune ? 1u : 0u) << (8 * (e - 4));
        }
        if (any) st16(w, v, Vec<DT>::pack(f));
        if (mask_out) {
            uint8_t* m = mask_out + v * N;
            *(uint32_t*)m = lo;
            if (N == 8) *(uint32_t*)(m + 4) = hi;
        }
    }
}

template <int DT>
__global__ __launch_bounds__(WM_THREADS) void wanda_matrix_apply2_kernel(const BracketGroup bg,
                                                                         const uint32_t* __restrict__ part) {
    const MatGroup& g = bg.m;
    const int it = group_item(g, blockIdx.x);
    const unsigned lb = blockIdx.x - g.start[it], nb = g.start[it + 1] - g.start[it];
    void* w = g.w[it];
    const int64_t rows = g.rows[it], cols = g.cols[it];
    const uint32_t rank0 = g.rank0[it];
    BracketState* bs = bg.bs[it];
    uint8_t* mask_out = g.mask[it];
    constexpr int N = Vec<DT>::N;
    __shared__ uint32_t wave4[WM_WAVES];
    __shared__ uint32_t out2[2];
    __shared__ uint32_t is_last, below_s, ln, lbase, binc_s;
    __shared__ __attribute__((aligned(16))) uint32_t arr[WS_CAP];   // (first the slots' sums, 8 x 512)
    __shared__ __attribute__((aligned(16))) uint32_t hsel[4096];
    __shared__ __attribute__((aligned(16))) float sq_lds[WM_SQ_LDS];
    K7_ENTER(2); K7_STAMP(2, 0);
    if (!bs->valid) {                          // (an empty bracket: cannot happen, handled all the same)
        if (lb == 0) {
            const float* sq0 = stage_sqrt(bg.sr[it], cols, sq_lds);
            matrix_fallback_body<DT>(w, sq0, rows, cols, rank0, mask_out, hsel, wave4, out2, 1u);
        }
        return;
    }
    const uint32_t lo = bs->lo, hi = bs->hi, shift = bs->shift;
    // Loads return in order, so the prologue's own loads go FIRST and the workgroup's share of W
    // right behind them: the prologue then runs while W streams in.
    // The matrix's counts: the sum of its workgroups' slots (L2-resident, 2 KB each)
    const uint32_t* pp = part + (size_t)g.start[it] * WS_PART_WORDS;
    // eight groups of 128 threads, 4 bins per thread, one slot per group at a time: 12 slots per
    // group in flight, one latency for up to 96 workgroups (more: further rounds, still ahead of
    // the W loads); the groups' sums meet in LDS
    constexpr int PXG = WM_THREADS / (WS_BINS / 4), PXN = 12;
    const unsigned grp = threadIdx.x / (WS_BINS / 4), b4 = (threadIdx.x % (WS_BINS / 4)) * 4u;
    u32x4 px[PXN];
#pragma unroll
    for (int u = 0; u < PXN; ++u) {
        const unsigned q = grp + (unsigned)PXG * u;
        const u32x4 x = *(const u32x4*)(pp + (size_t)(q < nb ? q : nb - 1u) * WS_PART_WORDS + b4);
        px[u] = q < nb ? x : u32x4{0u, 0u, 0u, 0u};
    }
    // one wave: the slots' counts below the bracket
    const bool tb_wave = threadIdx.x >= WS_BINS && threadIdx.x < WS_BINS + 64;
    uint32_t tb0 = 0, tb1 = 0;
    if (tb_wave) {
        const unsigned q0 = threadIdx.x - WS_BINS, q1 = q0 + 64u;
        const uint32_t x0 = pp[(size_t)(q0 < nb ? q0 : nb - 1u) * WS_PART_WORDS + WS_BINS];
        const uint32_t x1 = pp[(size_t)(q1 < nb ? q1 : nb - 1u) * WS_PART_WORDS + WS_BINS];
        tb0 = q0 < nb ? x0 : 0u;
        tb1 = q1 < nb ? x1 : 0u;
    }
    u32x4 acc = {0u, 0u, 0u, 0u};
    if (nb > (unsigned)(PXG * PXN)) {          // (rare: one matrix on more than 96 workgroups)
        for (unsigned q = grp + (unsigned)(PXG * PXN); q < nb; q += PXG)
            acc += *(const u32x4*)(pp + (size_t)q * WS_PART_WORDS + b4);
        if (tb_wave)
            for (unsigned q = threadIdx.x - WS_BINS + 128u; q < nb; q += 64u) tb0 += pp[(size_t)q * WS_PART_WORDS + WS_BINS];
    }
    const SqrtPre srp = stage_sqrt_issue(bg.sr[it], cols);
    if (threadIdx.x == 0) ln = 0u;
    lds_barrier();          // (every wave's prologue loads are queued before any wave's W loads)
    const int64_t vpr = cols / N;
    const int64_t nvec = rows * vpr;
    const uint32_t nvec32 = (uint32_t)nvec, stride32 = nb * WM_THREADS;
    const uint32_t vfirst32 = lb * WM_THREADS + threadIdx.x;
    const int64_t stride = (int64_t)stride32, vfirst = (int64_t)vfirst32;
    u32x4 cur[WS_PRE];
#pragma unroll
    for (int j = 0; j < WS_PRE; ++j) {
        const uint32_t v = vfirst32 + j * stride32;
        cur[j] = ld16(w, (int64_t)(v < nvec32 ? v : nvec32 - 1u));  // (clamped: no branch around a load)
    }
    {
#pragma unroll
        for (int u = 0; u < PXN; ++u) acc += px[u];
        *(u32x4*)(arr + grp * WS_BINS + b4) = acc;
    }
    if (tb_wave) {
        uint32_t tb = tb0 + tb1;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tb += __shfl_xor(tb, off, 64);
        if (threadIdx.x == WS_BINS) below_s = tb;
    }
    const float* sq = stage_sqrt_finish(srp, bg.sr[it], cols, sq_lds);      // (ends with a barrier)
    uint32_t c = 0;
    if (threadIdx.x < WS_BINS) {
#pragma unroll
        for (int u = 0; u < PXG; ++u) c += arr[u * WS_BINS + threadIdx.x];
    }
    const uint32_t below = below_s;
    uint32_t total;
    const uint32_t incl = block_scan_wm<true>(c, wave4, total);
    const bool miss = rank0 <= below || rank0 - below > total;
    const uint32_t r = rank0 - below;
    if (threadIdx.x == 0) { out2[0] = 0xffffffffu; out2[1] = 0; binc_s = 0; }
    lds_barrier();
    if (!miss && incl - c < r && r <= incl) { out2[0] = threadIdx.x; out2[1] = r - (incl - c); binc_s = c; }
    lds_barrier();
    const uint32_t bin = out2[0], rr = out2[1], binc = binc_s;
    uint64_t binhi64 = (uint64_t)lo + ((uint64_t)(bin + 1u) << shift);
    if (binhi64 > (uint64_t)hi) binhi64 = hi;
    // The cases the two passes cannot settle, known here to every workgroup of the matrix and
    // before any of them has written a byte of W: the bracket missed, the threshold's bin reaches
    // into Inf / NaN bit patterns (`metric <= thres` is not an order on bits there), the bin holds
    // more elements than the list (massive ties).  All leave; the first runs the exact finish.
    const bool unsettled = miss || bin == 0xffffffffu || binhi64 > 0x7f800000ull;
    if (unsettled || binc > (uint32_t)WS_CAP) {
        if (lb == 0) matrix_fallback_body<DT>(w, sq, rows, cols, rank0, mask_out, hsel, wave4, out2, unsettled ? 1u : 2u);
        return;
    }
    const uint32_t binlo = lo + (bin << shift), binw = (uint32_t)binhi64 - binlo;
    K7_STAMP(2, 1);
    {
        const uint32_t vpr32 = (uint32_t)vpr, step32 = stride32 % vpr32;
        uint32_t cv = vfirst32 % vpr32;
        auto apply_vec = [&](const u32x4& x, int64_t v, int64_t c0) {
            float f[N];
            Vec<DT>::unpack(x, f);
            uint32_t lo4 = 0, hi4 = 0;
            bool any = false, open_any = false;
#pragma unroll
            for (int q = 0; q < N / 4; ++q) {
                const u32x4 s4 = ld_sq4(sq, c0 / 4 + q);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = 4 * q + i;
                    const uint32_t b = __float_as_uint(__builtin_fabsf(f[e]) * __uint_as_float(s4[i]));
                    const bool prune = b < binlo;
                    open_any |= (b - binlo) < binw;          // (b < binlo wraps to a huge value)
                    if (prune) { f[e] = 0.0f; any = true; }
                    if (mask_out) {
                        if (e < 4) lo4 |= (prune ? 1u : 0u) << (8 * e);
                        else hi4 |= (prune ? 1u : 0u) << (8 * (e - 4));
                    }
                }
            }
            if (!open_any) {
                if (any) st16(w, v, Vec<DT>::pack(f));
                if (mask_out) {
                    uint8_t* m = mask_out + v * N;
                    *(uint32_t*)m = lo4;
                    if (N == 8) *(uint32_t*)(m + 4) = hi4;
                }
            } else {
                // a vector with an element of the threshold's bin (a few hundred per matrix):
                // element stores only, and never the open element's own bytes (the last
                // workgroup may write them).  Metrics again from the untouched vector.
                float f0[N];
                Vec<DT>::unpack(x, f0);
#pragma unroll
                for (int e = 0; e < N; ++e) {
                    const int64_t idx = v * N + e;
                    const uint32_t b = __float_as_uint(__builtin_fabsf(f0[e]) * sq[c0 + e]);
                    if ((b - binlo) < binw) {
                        // the workgroup's own list first (LDS; `hsel` is free until the settling):
                        // one atomic per WORKGROUP at the memory side instead of one per element —
                        // ~1000 returning atomics on one address were most of this pass's time
                        const uint32_t ls = atomicAdd(&ln, 1u);
                        if (ls < 2048u) { hsel[2 * ls] = (uint32_t)idx; hsel[2 * ls + 1] = b; continue; }
                        const uint32_t slot = atomicAdd(&bs->list_count, 1u);
                        if (slot < (uint32_t)WS_CAP) {
                            __hip_atomic_store(&bs->list_idx[slot], (uint32_t)idx, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(&bs->list_bits[slot], b, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                        }
                    } else {
                        const bool prune = b < binlo;
                        if (prune) Vec<DT>::store1(w, idx, 0.0f);
                        if (mask_out) mask_out[idx] = prune ? 1 : 0;
                    }
                }
            }
        };
        // the next WS_PRE loads in flight while the current vectors are decided and stored
        const uint32_t wg_first = lb * WM_THREADS;              // (uniform: the loop's trip count)
        for (uint32_t base = 0; wg_first + base < nvec32; base += WS_PRE * stride32) {
            u32x4 nxt[WS_PRE];
            const bool more = wg_first + base + WS_PRE * stride32 < nvec32;
            if (more) {
#pragma unroll
                for (int j = 0; j < WS_PRE; ++j) {
                    const uint32_t v = vfirst32 + base + (WS_PRE + j) * stride32;
                    nxt[j] = ld16(w, (int64_t)(v < nvec32 ? v : nvec32 - 1u));
                }
            }
#pragma unroll
            for (int j = 0; j < WS_PRE; ++j) {
                const int64_t c0 = (int64_t)cv * N;
                cv += step32;
                if (cv >= vpr32) cv -= vpr32;
                const uint32_t v = vfirst32 + base + j * stride32;
                if (v < nvec32) apply_vec(cur[j], (int64_t)v, c0);
            }
            if (more) {
#pragma unroll
                for (int j = 0; j < WS_PRE; ++j) cur[j] = nxt[j];
            }
        }
    }
    // ---- the workgroup's list joins the matrix's -------------------------------------------------
    K7_STAMP(2, 2);
    __syncthreads();
    {
        const uint32_t nl = ln < 2048u ? ln : 2048u;
        if (threadIdx.x == 0) lbase = nl ? atomicAdd(&bs->list_count, nl) : 0u;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nl; i += WM_THREADS) {
            const uint32_t slot = lbase + i;
            if (slot < (uint32_t)WS_CAP) {
                __hip_atomic_store(&bs->list_idx[slot], hsel[2 * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&bs->list_bits[slot], hsel[2 * i + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    // ---- the last workgroup of this matrix settles the threshold's bin ----------------------
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    K7_STAMP(2, 3);
    if (threadIdx.x == 0)
        is_last = (__hip_atomic_fetch_add(&bs->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                   == nb - 1u) ? 1u : 0u;
    __syncthreads();
    K7_EXIT(2);
    if (!is_last) return;
#ifdef ECO_K7_CLOCK
    if (threadIdx.x == 0) atomicMin(&eco_k7_clk[3][14], wall_clock64());
#endif
    const uint32_t n = __hip_atomic_load(&bs->list_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n != binc) {       // every element of the bin is appended exactly once: cannot happen; counted, never silent
        if (threadIdx.x == 0) { bs->fallback = 3u; atomicAdd(&eco_k7_fallbacks[3], 1u); }
        return;
    }
    // the list's metrics relative to the bin's low edge: all below binw <= 2^shift, so the rr-th
    // smallest takes ceil(shift / 12) digit passes (one for the usual bin of <= 4096 bit patterns)
    for (uint32_t i = threadIdx.x; i < n; i += WM_THREADS)
        arr[i] = __hip_atomic_load(&bs->list_bits[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - binlo;
    uint32_t prefix = 0, remaining = rr;
    const int passes = shift <= 12 ? 1 : (shift <= 24 ? 2 : 3);
#pragma unroll 1
    for (int pass = 3 - passes; pass < 3; ++pass) {
        const int sh = pass == 0 ? 24 : (pass == 1 ? 12 : 0);
        const uint32_t himask = pass == 0 ? 0u : (pass == 1 ? 0xff000000u : 0xfffff000u);
        const uint32_t dmask = pass == 0 ? 255u : 4095u;
        for (int i = threadIdx.x; i < 4096; i += WM_THREADS) hsel[i] = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += WM_THREADS) {
            const uint32_t b = arr[i];
            if ((b & himask) == prefix) atomicAdd(&hsel[(b >> sh) & dmask], 1u);
        }
        __syncthreads();
        const u32x4 c4 = *(const u32x4*)(hsel + 4 * threadIdx.x);
        const uint32_t csum = c4[0] + c4[1] + c4[2] + c4[3];
        uint32_t tot;
        const uint32_t inc = block_scan_wm(csum, wave4, tot);
        uint32_t exc = inc - csum;
        if (exc < remaining && remaining <= inc) {
            int d = 0;
            while (remaining > exc + c4[d]) { exc += c4[d]; ++d; }
            out2[0] = 4 * threadIdx.x + (uint32_t)d;
            out2[1] = remaining - exc;
        }
        __syncthreads();
        prefix |= out2[0] << sh;
        remaining = out2[1];
        __syncthreads();
    }
    const uint32_t thres = prefix;           // sorted[k], relative to the bin's low edge
    for (uint32_t i = threadIdx.x; i < n; i += WM_THREADS) {
        const uint32_t idx = __hip_atomic_load(&bs->list_idx[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool prune = arr[i] <= thres;
        if (prune) Vec<DT>::store1(w, (int64_t)idx, 0.0f);
        if (mask_out) mask_out[idx] = prune ? 1 : 0;
    }
    K7_EXIT(3);
}

// the counting pass's slots, one per workgroup of a launch (matrix-mode groups run one after the
// other on the stream and share them)
#define WS_PART_BYTES ((size_t)(WM_TOTAL_WGS + WMAX) * WS_PART_WORDS * sizeof(uint32_t))
static inline size_t sq_bytes(int64_t cols) { return (((size_t)cols * sizeof(float) + 255) / 256) * 256; }

extern "C" size_t ecoflap_wanda_workspace_bytes(int64_t rows, int64_t cols) {
    (void)rows;
    if (cols <= 0) return 0;
    // sqrt table (padded to 256 B) + matrix-mode selection states
    return sq_bytes(cols) + sizeof(MatrixSelState) + sizeof(BracketState) + WS_PART_BYTES;
}

extern "C" size_t ecoflap_wanda_block_workspace_bytes(const ecoflap_wanda_item* items, int n_items) {
    size_t total = 0;
    if (!items) return 0;
    for (int i = 0; i < n_items; ++i)
        if (items[i].cols > 0)
            total += sq_bytes(items[i].cols) + sizeof(MatrixSelState) + sizeof(BracketState);
    return total ? total + WS_PART_BYTES : 0;
}

static inline bool item_vector_ok(const ecoflap_wanda_item& it, const float* sq) {
    const int nvec_elems = it.dtype == ECOFLAP_F32 ? 4 : 8;
    // mask rows are written with 4/8-byte stores: cols % 8 keeps them aligned for 16-bit dtypes
    return it.cols % nvec_elems == 0 && aligned16(it.w) && aligned16(sq) &&
           (!it.mask_out || (((uintptr_t)it.mask_out) & 7u) == 0);
}

#define DT_SWITCH(dt, CALL)                                    \
    do {                                                       \
        if ((dt) == ECOFLAP_F32) { CALL(ECOFLAP_F32); }        \
        else if ((dt) == ECOFLAP_F16) { CALL(ECOFLAP_F16); }   \
        else { CALL(ECOFLAP_BF16); }                           \
    } while (0)

// All Linears of one transformer block in one call: one sqrt launch, one selection launch per
// (dtype, register class) of the rows-mode matrices, 3 histogram launches + 1 apply launch for
// ALL matrix-mode matrices of a dtype — instead of 2 (rows) / 6 (matrix) launches per matrix.
// Besides the launch count this is what fills the chip: one T5 matrix has 2048-5120 rows (2-5
// waves per SIMD for the one-wave-per-row kernel), a T5 block has 18-28 thousand.
extern "C" int ecoflap_wanda_prune_block(const ecoflap_wanda_item* items, int n_items,
                                         void* workspace, size_t workspace_bytes, void* stream) {
    if (n_items < 0 || n_items > WMAX) return ECOFLAP_ESIZE;
    if (n_items == 0) return 0;
    if (!items) return ECOFLAP_ENULL;
    for (int i = 0; i < n_items; ++i) {
        const ecoflap_wanda_item& it = items[i];
        if (!dtype_ok(it.dtype)) return ECOFLAP_EDTYPE;
        if (it.mode != ECOFLAP_WANDA_ROWS && it.mode != ECOFLAP_WANDA_MATRIX) return ECOFLAP_EMODE;
        if (it.rows <= 0 || it.cols <= 0 || it.k < 0) return ECOFLAP_ESIZE;
        if (it.mode == ECOFLAP_WANDA_ROWS && it.cols > WANDA_ROWS_MAX_COLS) return ECOFLAP_ESIZE;
        // the reference indexes sorted[k]: k == numel raises IndexError there (W:555)
        if (it.mode == ECOFLAP_WANDA_MATRIX &&
            (it.k >= it.rows * it.cols || it.rows * it.cols >= (int64_t)0xffffffffLL))
            return ECOFLAP_ESIZE;
        if (it.rows > 0x7fffffffLL / 2) return ECOFLAP_ESIZE;
        if (!it.w || !it.scaler_row) return ECOFLAP_ENULL;
    }
    if (!workspace) return ECOFLAP_ENULL;
    if (workspace_bytes < ecoflap_wanda_block_workspace_bytes(items, n_items)) return ECOFLAP_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;

    // workspace: sqrt tables, then the selection states
    float* sq[WMAX];
    MatrixSelState* st[WMAX];
    char* p = (char*)workspace;
    for (int i = 0; i < n_items; ++i) { sq[i] = (float*)p; p += sq_bytes(items[i].cols); }
    for (int i = 0; i < n_items; ++i) {
        st[i] = (MatrixSelState*)p;      // cleared by the sqrt kernel below (matrix-mode items)
        p += sizeof(MatrixSelState);
    }
    BracketState* bst[WMAX];
    for (int i = 0; i < n_items; ++i) { bst[i] = (BracketState*)p; p += sizeof(BracketState); }
    uint32_t* part = (uint32_t*)p;
    // The sampled-bracket path (matrix-mode items big enough to sample one vector per stride; it
    // stages sqrt(scaler_row) itself and clears its own state): 2 reads + 1 write of W instead of
    // 4 + 1, in two launches.  DEFAULT since round 5 — the matrices it cannot settle are finished
    // exactly inside the second launch (`matrix_fallback_body`), no flag crosses to the host and
    // the call stays asynchronous.  ECOFLAP_WANDA_SAMPLED=0 (read at every call) selects the
    // three-histogram path for every matrix.
    const char* sampled_env = getenv("ECOFLAP_WANDA_SAMPLED");
    const bool force_legacy = sampled_env != nullptr && sampled_env[0] == '0';
    bool sampled_item[WMAX];
    for (int i = 0; i < n_items; ++i) {
        const ecoflap_wanda_item& a = items[i];
        const int64_t nv = a.rows * a.cols / (a.dtype == ECOFLAP_F32 ? 4 : 8);
        sampled_item[i] = a.mode == ECOFLAP_WANDA_MATRIX && !force_legacy && item_vector_ok(a, sq[i]) &&
                          nv >= 8 * WS_SAMPLE_VECS && a.cols <= WM_SQ_LDS && a.rows * a.cols <= WS_MAX_NUMEL &&
                          (((uintptr_t)a.scaler_row) & 15u) == 0;
    }
    auto launch_sqrt = [&](const bool* want) -> int {
        SqrtGroup g;
        g.n = n_items;
        g.zero_words = (int)(sizeof(MatrixSelState) / sizeof(uint32_t));
        g.start[0] = 0;
        for (int i = 0; i < n_items; ++i) {
            g.src[i] = items[i].scaler_row; g.dst[i] = sq[i]; g.cols[i] = items[i].cols;
            g.zero[i] = items[i].mode == ECOFLAP_WANDA_MATRIX ? (uint32_t*)st[i] : nullptr;
            g.start[i + 1] = g.start[i] + (want[i] ? (int32_t)((items[i].cols + 255) / 256) : 0);
        }
        if (g.start[n_items] == 0) return 0;
        hipLaunchKernelGGL(sqrt_cols_kernel, dim3((unsigned)g.start[n_items]), dim3(256), 0, s, g);
        ECO_CHECK_LAUNCH();
        return 0;
    };
    {
        bool want[WMAX];
        for (int i = 0; i < n_items; ++i) want[i] = !sampled_item[i];
        const int rc = launch_sqrt(want);
        if (rc) return rc;
    }

    bool done[WMAX];
    for (int i = 0; i < n_items; ++i) done[i] = false;

    // ---- rows mode -------------------------------------------------------------------------
    // Groups of one (dtype, form, register class) each; a wave-form group and a workgroup-form
    // group of the same dtype share one grid when the long rows fit 4 vectors per thread.
    struct Pending { RowsGroup g; int dtype, cls; bool wave, used; };
    static thread_local Pending pend[8];
    int n_pend = 0;
    for (int i = 0; i < n_items; ++i) {
        if (done[i] || items[i].mode != ECOFLAP_WANDA_ROWS) continue;
        const ecoflap_wanda_item& a = items[i];
        const int nve = a.dtype == ECOFLAP_F32 ? 4 : 8;
        if (!item_vector_ok(a, sq[i])) {     // odd widths / unaligned views: the LDS form, per matrix
            const size_t lds = ((size_t)a.cols + 256 + 4 + 4) * sizeof(uint32_t);
#define ROWS_LDS(DT_) hipLaunchKernelGGL((wanda_rows_kernel<DT_>), dim3((unsigned)a.rows), dim3(256), lds, s, a.w, sq[i], a.cols, a.k, a.mask_out)
            DT_SWITCH(a.dtype, ROWS_LDS);
#undef ROWS_LDS
            ECO_CHECK_LAUNCH();
            done[i] = true;
            continue;
        }
        // measured on MI355X: the wave form wins up to 256 vectors per row (2048 bf16 columns)
        const int64_t nvec = a.cols / nve;
        const bool wave = nvec <= ROWS_WAVE_MAX_NVEC;
        const int cls = wave ? rows_wave_class(nvec) : rows_reg_class(nvec);
        if (n_pend == 8) {                   // more classes than slots: flush what is pending
            for (int q = 0; q < n_pend; ++q) {
                Pending& P = pend[q];
#define ROWS_GO(DT_) do { if (P.wave) launch_rows_wave<DT_>(P.g, P.cls, s); else launch_rows_reg<DT_>(P.g, P.cls, s); } while (0)
                DT_SWITCH(P.dtype, ROWS_GO);
#undef ROWS_GO
                ECO_CHECK_LAUNCH();
            }
            n_pend = 0;
        }
        Pending& P = pend[n_pend++];
        P.dtype = a.dtype; P.cls = cls; P.wave = wave; P.used = false;
        RowsGroup& g = P.g;
        g.n = 0;
        g.start[0] = 0;
        for (int j = i; j < n_items; ++j) {
            const ecoflap_wanda_item& b = items[j];
            if (done[j] || b.mode != ECOFLAP_WANDA_ROWS || b.dtype != a.dtype || !item_vector_ok(b, sq[j]))
                continue;
            const int64_t nvb = b.cols / nve;
            const bool wb = nvb <= ROWS_WAVE_MAX_NVEC;
            if (wb != wave || (wb ? rows_wave_class(nvb) : rows_reg_class(nvb)) != cls) continue;
            if ((int64_t)g.start[g.n] + b.rows > 0x3fffffffLL) continue;   // next group
            g.w[g.n] = b.w; g.sq[g.n] = sq[j]; g.cols[g.n] = b.cols; g.k[g.n] = b.k; g.mask[g.n] = b.mask_out;
            g.start[g.n + 1] = g.start[g.n] + (int32_t)b.rows;
            ++g.n;
            done[j] = true;
        }
    }
    for (int q = 0; q < n_pend; ++q) {
        Pending& P = pend[q];
        if (P.used) continue;
        P.used = true;
        Pending* mate = nullptr;             // the other form, same dtype, long rows of <= 4 vectors per thread
        for (int r = q + 1; r < n_pend && !mate; ++r) {
            Pending& Q = pend[r];
            if (Q.used || Q.dtype != P.dtype || Q.wave == P.wave) continue;
            const Pending& longer = P.wave ? Q : P;
            if (longer.cls <= 4) mate = &Q;
        }
        if (mate) {
            mate->used = true;
            const Pending& W = P.wave ? P : *mate;
            const Pending& L = P.wave ? *mate : P;
#define FUSED_GO(DT_) launch_rows_fused<DT_>(W.g, W.cls, L.g, L.cls, s)
            DT_SWITCH(P.dtype, FUSED_GO);
#undef FUSED_GO
        } else {
#define ROWS_GO(DT_) do { if (P.wave) launch_rows_wave<DT_>(P.g, P.cls, s); else launch_rows_reg<DT_>(P.g, P.cls, s); } while (0)
            DT_SWITCH(P.dtype, ROWS_GO);
#undef ROWS_GO
        }
        ECO_CHECK_LAUNCH();
    }

    // ---- matrix mode -----------------------------------------------------------------------
    for (int i = 0; i < n_items; ++i) {
        if (done[i]) continue;
        const ecoflap_wanda_item& a = items[i];
        const bool vec = item_vector_ok(a, sq[i]);
        MatGroup g;
        g.n = 0;
        g.start[0] = 0;
        // the group's workgroups are shared out in proportion to the matrices' sizes (equal work
        // per workgroup: a ViT-g block's matrices differ 4x), WM_TOTAL_WGS over the launch
        int members[WMAX];
        int64_t total_elems = 0;
        for (int j = i; j < n_items; ++j) {
            const ecoflap_wanda_item& b = items[j];
            if (done[j] || b.dtype != a.dtype || item_vector_ok(b, sq[j]) != vec ||
                sampled_item[j] != sampled_item[i]) continue;
            members[g.n++] = j;
            total_elems += b.rows * b.cols;
            done[j] = true;
        }
        const int budget = WM_TOTAL_WGS;
        for (int q = 0; q < g.n; ++q) {
            const ecoflap_wanda_item& b = items[members[q]];
            const int64_t n = b.rows * b.cols;
            // >= 4 vectors per thread and pass
            int64_t cap = vec ? (n / (b.dtype == ECOFLAP_F32 ? 4 : 8) + 4 * WM_THREADS - 1) / (4 * WM_THREADS)
                              : (b.rows + 3) / 4;
            int64_t nb = (int64_t)((double)budget * (double)n / (double)total_elems + 0.5);
            if (nb > cap) nb = cap;
            if (nb < 1) nb = 1;
            g.w[q] = b.w; g.sq[q] = sq[members[q]]; g.rows[q] = b.rows; g.cols[q] = b.cols;
            g.rank0[q] = (uint32_t)(b.k + 1);     // sorted[k], 0-indexed -> (k+1)-th smallest
            g.st[q] = st[members[q]]; g.mask[q] = b.mask_out;
            g.start[q + 1] = g.start[q] + (int32_t)nb;
        }
        const dim3 grid((unsigned)g.start[g.n]), blk(WM_THREADS);
#define MATRIX_LEGACY(DT_, G_, GRID_)                                                               \
    do {                                                                                            \
        if (vec) {                                                                                  \
            hipLaunchKernelGGL((wanda_matrix_hist_kernel<DT_, 0, true, WM_SUB0>), GRID_, blk, 0, s, G_); \
            hipLaunchKernelGGL((wanda_matrix_hist_kernel<DT_, 1, true>), GRID_, blk, 0, s, G_);     \
            hipLaunchKernelGGL((wanda_matrix_hist_kernel<DT_, 2, true>), GRID_, blk, 0, s, G_);     \
            hipLaunchKernelGGL((wanda_matrix_apply_kernel<DT_, true>), GRID_, blk, 0, s, G_);       \
        } else {                                                                                    \
            hipLaunchKernelGGL((wanda_matrix_hist_kernel<DT_, 0, false>), GRID_, blk, 0, s, G_);    \
            hipLaunchKernelGGL((wanda_matrix_hist_kernel<DT_, 1, false>), GRID_, blk, 0, s, G_);    \
            hipLaunchKernelGGL((wanda_matrix_hist_kernel<DT_, 2, false>), GRID_, blk, 0, s, G_);    \
            hipLaunchKernelGGL((wanda_matrix_apply_kernel<DT_, false>), GRID_, blk, 0, s, G_);      \
        }                                                                                           \
    } while (0)
        // sampled bracket (2 reads + 1 write) when every matrix of the group is big enough to
        // sample one vector per stride; otherwise / on a flagged matrix the three histograms
        const bool sampled = sampled_item[i];
        if (!sampled) {
#define MATRIX_GO(DT_) MATRIX_LEGACY(DT_, g, grid)
            DT_SWITCH(a.dtype, MATRIX_GO);
#undef MATRIX_GO
            ECO_CHECK_LAUNCH();
            continue;
        }
        BracketGroup bg;
        bg.m = g;
        for (int q = 0; q < g.n; ++q) {
            bg.bs[q] = bst[members[q]];
            bg.sr[q] = items[members[q]].scaler_row;
        }
        if ((size_t)g.start[g.n] > (size_t)(WM_TOTAL_WGS + WMAX)) return ECOFLAP_ESIZE;   // (cannot happen: see nb)
#define MATRIX_GO(DT_)                                                                              \
    do {                                                                                            \
        hipLaunchKernelGGL((wanda_matrix_sbracket_kernel<DT_>), grid, blk, 0, s, bg, part);         \
        hipLaunchKernelGGL((wanda_matrix_apply2_kernel<DT_>), grid, blk, 0, s, bg, (const uint32_t*)part); \
    } while (0)
        DT_SWITCH(a.dtype, MATRIX_GO);
#undef MATRIX_GO
        ECO_CHECK_LAUNCH();
#undef MATRIX_LEGACY
    }
    return 0;
}

static int wanda_single(void* w, const float* scaler_row, int64_t rows, int64_t cols, int dtype,
                        int64_t k, uint8_t* mask_out, int mode, void* workspace,
                        size_t workspace_bytes, void* stream) {
    ecoflap_wanda_item it;
    it.w = w; it.scaler_row = scaler_row; it.rows = rows; it.cols = cols; it.k = k;
    it.mask_out = mask_out; it.dtype = dtype; it.mode = mode;
    return ecoflap_wanda_prune_block(&it, 1, workspace, workspace_bytes, stream);
}

extern "C" int ecoflap_wanda_prune_rows(void* w, const float* scaler_row, int64_t rows,
                                        int64_t cols, int dtype, int64_t k, uint8_t* mask_out,
                                        void* workspace, size_t workspace_bytes, void* stream) {
    return wanda_single(w, scaler_row, rows, cols, dtype, k, mask_out, ECOFLAP_WANDA_ROWS, workspace,
                        workspace_bytes, stream);
}

extern "C" int ecoflap_wanda_prune_matrix(void* w, const float* scaler_row, int64_t rows,
                                          int64_t cols, int dtype, int64_t k, uint8_t* mask_out,
                                          void* workspace, size_t workspace_bytes, void* stream) {
    return wanda_single(w, scaler_row, rows, cols, dtype, k, mask_out, ECOFLAP_WANDA_MATRIX,
                        workspace, workspace_bytes, stream);
}

// =====================================================================================
// K7, structured n:m branch (wanda_pruner.py:265-270 = :546-551): in every group of m consecutive
// columns of a row the n smallest metrics are zeroed (`topk(..., largest=False)`: a NaN counts as
// the largest; equal metrics: the lower column first).  HBM-bound, one pass: 2*s*numel + 4*cols.
// One thread per group: the group's metrics as integer keys (non-negative floats order as their
// bits, NaN -> 0xffffffff), each element's rank by counting.
template <int DT>
__global__ __launch_bounds__(256) void wanda_nm_kernel(void* __restrict__ w, const float* __restrict__ scaler_row,
                                                       int64_t rows, int64_t cols, int n, int m,
                                                       uint8_t* __restrict__ mask_out) {
    const int64_t gpr = (cols + m - 1) / m;                    // groups per row
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= rows * gpr) return;
    const int64_t r = g / gpr, c0 = (g - r * gpr) * m;
    const int len = (int)(cols - c0 < m ? cols - c0 : m);
    uint32_t key[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        key[j] = 0xffffffffu;
        if (j < len) {
            const float v = __builtin_fabsf(Vec<DT>::load1(w, r * cols + c0 + j)) * __builtin_sqrtf(scaler_row[c0 + j]);
            key[j] = v != v ? 0xffffffffu : __float_as_uint(v);
        }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (j >= len) break;
        int rank = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < len && i != j && (key[i] < key[j] || (key[i] == key[j] && i < j))) ++rank;
        const bool z = rank < n;
        if (z) Vec<DT>::store1(w, r * cols + c0 + j, 0.0f);
        if (mask_out) mask_out[r * cols + c0 + j] = z ? 1 : 0;
    }
}

extern "C" int ecoflap_wanda_prune_nm(void* w, const float* scaler_row, int64_t rows, int64_t cols,
                                      int dtype, int n, int m, uint8_t* mask_out, void* stream) {
    if (!dtype_ok(dtype)) return ECOFLAP_EDTYPE;
    if (rows < 0 || cols < 0 || n <= 0 || m < n || m > 16) return ECOFLAP_ESIZE;
    if (cols % m != 0 && cols % m < n) return ECOFLAP_ESIZE;       // (topk raises on the short slice)
    if (rows == 0 || cols == 0) return 0;
    if (!w || !scaler_row) return ECOFLAP_ENULL;
    const int64_t groups = rows * ((cols + m - 1) / m);
    if ((groups + 255) / 256 > 0x7fffffffLL) return ECOFLAP_ESIZE;
    const dim3 grid((unsigned)((groups + 255) / 256));
#define NM_GO(DT_) hipLaunchKernelGGL((wanda_nm_kernel<DT_>), grid, dim3(256), 0, (hipStream_t)stream, w, scaler_row, rows, cols, n, m, mask_out)
    DT_SWITCH(dtype, NM_GO);
#undef NM_GO
    ECO_CHECK_LAUNCH();
    return 0;
}

// =====================================================================================
// K8  grad *= mask
// =====================================================================================
// 16-byte vectors of the gradient, 8 (4) mask bytes per vector; scalar form for unaligned tails
template <int DT>
__global__ __launch_bounds__(256) void mask_mul_kernel(void* g, const uint8_t* __restrict__ keep,
                                                       int64_t n) {
    constexpr int N = Vec<DT>::N;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const bool vec = (((uintptr_t)g) & 15u) == 0 && (((uintptr_t)keep) & (N - 1)) == 0;
    const int64_t nvec = vec ? n / N : 0;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += stride) {
        float f[N];
        Vec<DT>::unpack(ld16(g, v), f);
        uint32_t m0 = *(const uint32_t*)(keep + v * N), m1 = 0;
        if (N == 8) m1 = *(const uint32_t*)(keep + v * N + 4);
#pragma unroll
        for (int e = 0; e < N; ++e) {
            const uint32_t byte = ((e < 4 ? m0 : m1) >> (8 * (e & 3))) & 0xffu;
            f[e] = Vec<DT>::round(f[e] * (byte ? 1.0f : 0.0f));      // grad * mask, as the reference multiplies
        }
        st16(g, v, Vec<DT>::pack(f));
    }
    for (int64_t i = nvec * N + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
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
    int64_t b = (n / 8 + 256 * 2 - 1) / (256 * 2);            // ~2 vectors per thread
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
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
