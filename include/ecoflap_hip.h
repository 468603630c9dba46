/*
 * ecoflap_hip.h — C ABI of the MI355X (gfx950) ECoFLaP scoring + pruning kernels.
 *
 * This is the drop-in boundary of the hot path.  The reference
 * (ylsung/ECoFLaP) is pure Python on torch; it has no FFI of its own, so each
 * entry point below names the reference op chain (file:line, relative to the
 * reference checkout) that a maintainer would replace with a ctypes call — the
 * binding stub is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C types only: raw device pointers (tensor.data_ptr()), element
 *     counts, a dtype code, and the HIP stream handle as void*
 *     (torch.cuda.current_stream().cuda_stream); no torch types.
 *   - every function returns 0 on success, a positive hipError_t value on a
 *     HIP failure, or a negative ECOFLAP_E* code on a bad argument.
 *   - nothing is allocated, freed or synchronised inside a launch function
 *     (graph-capture safe); scratch is a caller-owned workspace whose size
 *     comes from the matching *_workspace_bytes() query.
 *   - no global state, re-entrant, no ownership transfer.
 *   - all float arithmetic is done in fp32 and rounded to the storage dtype
 *     after EVERY reference op (no fma contraction), so results equal the
 *     reference's torch op chain bit for bit when the same z is supplied.
 */
#ifndef ECOFLAP_HIP_H
#define ECOFLAP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* storage dtype codes (parameter / activation dtype) */
#define ECOFLAP_F32  0
#define ECOFLAP_F16  1
#define ECOFLAP_BF16 2

/* argument errors (negative so they never collide with hipError_t) */
#define ECOFLAP_EDTYPE  (-1)
#define ECOFLAP_ENULL   (-2)
#define ECOFLAP_ESIZE   (-3)
#define ECOFLAP_EMODE   (-4)
#define ECOFLAP_EALIGN  (-5)
#define ECOFLAP_EWORKSPACE (-6)

/* Rounds of the in-register Philox4x32 generator (Salmon et al., SC'11).  7 is the
 * smallest count the authors report as Crush-resistant (passes TestU01 BigCrush); 10 is the
 * conventional safety margin.  The layer-batched K1 kernel is VALU-bound on the generator for
 * bf16, so the build ships 7; set 10 here and rebuild for the
 * conventional stream.  Parity with the reference never depends on this stream: it is taken
 * with z supplied (z != NULL). */
#ifndef ECOFLAP_PHILOX_ROUNDS
#define ECOFLAP_PHILOX_ROUNDS 7
#endif

/* reduce modes of ecoflap_absprod_reduce* */
#define ECOFLAP_RED_ABSW_ABSG 0 /* sum |w|*|g|   GradMagAbs     layer_single_base_pruner.py:455,467 */
#define ECOFLAP_RED_SQW_SQG   1 /* sum w^2*g^2   GradMagSquare  :453,465 */
#define ECOFLAP_RED_ABSG      2 /* sum |g|       GradOnly       :455,469 (w ignored) */
#define ECOFLAP_RED_ABSW      3 /* sum |w|       MEZO-GradMagAbs    :556 (g ignored) */
#define ECOFLAP_RED_SQW       4 /* sum w^2       MEZO-GradMagSquare :559 (g ignored) */

const char* ecoflap_version(void);
const char* ecoflap_error_string(int code);

/* ---------------------------------------------------------------------------
 * K1  zeroth-order perturbation
 * replaces LayerSparsity.zo_perturb_parameters
 *   LAVIS/lavis/compression/pruners/layer_single_base_pruner.py:473-486
 *     torch.manual_seed(seed); z = torch.normal(0,1,size,dtype=param.dtype)
 *     param.data = param.data + scaling_factor * z * zo_eps
 * Arithmetic per element (rd = round-to-nearest-even to `dtype`):
 *     t = rd(z * scaling_factor);  u = rd(t * zo_eps);  w = rd(w + u)
 * z: if `z` != NULL it is a device array of n elements of `dtype` (parity
 * mode: the caller supplies the reference's own torch.normal draw); if NULL,
 * z is generated in registers: Philox4x32-R (R = ECOFLAP_PHILOX_ROUNDS), key = seed, Box-Muller,
 * rounded to `dtype` — the stream ecoflap_zo_fill_normal writes and
 * oracle/ecoflap_oracle.c:oracle_normal_stream restates (fp32 storage: one Philox call per 4
 * elements, 32-bit radius and angle words; fp16 / bf16: three calls per 16 elements, 32-bit
 * radius word + 16-bit angle per Box-Muller pair; element layout in csrc/zo_perturb.hip).
 * w must be 16-byte aligned (torch allocations are).
 * ------------------------------------------------------------------------- */
int ecoflap_zo_perturb(void* w, int64_t n, int dtype,
                       float scaling_factor, float zo_eps,
                       uint64_t seed, const void* z, void* stream);

/* The reference calls K1 three times per (layer, batch, noise) with the same
 * seed and scaling factors +1, -2, +1 (layer_single_base_pruner.py:530-539).
 * All three results are elementwise functions of (w, z); this entry point
 * computes them in ONE pass over w — same three roundings each, so
 * w_plus / w_minus / w_restored are bit-identical to three ecoflap_zo_perturb
 * calls — reading w once and generating z once (4*s instead of 6*s bytes per
 * element).  Outputs may alias w_in (in place) but not each other.
 * w_plus and w_minus may BOTH be NULL: drift-only form that stores just
 * w_restored (2*s bytes per element) — used by data-parallel ranks that do not
 * evaluate a batch but must carry the reference's rounding drift so that every
 * replica ends with the single-process weights. */
int ecoflap_zo_perturb_triple(const void* w_in, void* w_plus, void* w_minus,
                              void* w_restored, int64_t n, int dtype,
                              float zo_eps, uint64_t seed, const void* z,
                              void* stream);

/* Layer-batched form: n_units consecutive triples of the SAME matrix in one pass
 * (bit-identical to n_units ecoflap_zo_perturb_triple calls chained through
 * w_restored).  w is read once and rewritten in place with the final drifted
 * weights; unit u's theta+ / theta- go to w_plus[u] / w_minus[u] (host arrays
 * of device pointers; a NULL pair = carry the drift only, for units another
 * rank evaluates).  seeds: host array of n_units seeds; z: NULL (in-register
 * Philox) or host array of n_units device pointers (parity mode).
 * n_units <= ECOFLAP_MAX_UNITS; callers chunk longer schedules.
 * Algorithmic bytes: (2*owned_units + 2) * s per element. */
#define ECOFLAP_MAX_UNITS 32
int ecoflap_zo_perturb_units(void* w, int64_t n, int dtype, float zo_eps,
                             int n_units, const uint64_t* seeds,
                             void* const* w_plus, void* const* w_minus,
                             const void* const* z, void* stream);

/* Same launch, instrumented: start_event / stop_event (hipEvent_t created by the caller with
 * timing enabled) receive the kernel's own begin / end timestamps (hipExtLaunchKernelGGL), so
 * hipEventElapsedTime(start, stop) is the kernel duration rocprofv3 reports — no marker
 * latency, no host gap.  bench.py's roofline leg uses it; the product loop uses the plain form. */
int ecoflap_zo_perturb_units_timed(void* w, int64_t n, int dtype, float zo_eps,
                                   int n_units, const uint64_t* seeds,
                                   void* const* w_plus, void* const* w_minus,
                                   const void* const* z, void* stream,
                                   void* start_event, void* stop_event);

/* Block-batched form: the layer-batched pass for SEVERAL matrices (the layers of one transformer
 * block, scored back to back) in one launch, in-register z only.  Per layer the drifted weights
 * go to w_final (NOT over w_in: the block's other layers are still evaluated with the original
 * weights; the caller copies w_final in when the layer's turn is over).  table: DEVICE int64
 * [n_layers][5 + 3*ECOFLAP_MAX_UNITS], row = {w_in, w_final, numel, n_units, first_super_row,
 * seeds[MAX_UNITS], w_plus[MAX_UNITS], w_minus[MAX_UNITS]} (super-row = 128 16-byte vectors:
 * rows_l = ceil(numel_l / (128 * 16 / s)); first_super_row = sum of the earlier layers' rows);
 * total_rows = the sum over all layers.  One dtype per launch.  start_event / stop_event: both
 * NULL, or a hipEvent_t pair that receives the kernel's begin / end (as the _timed form).
 * Bit-identical to one ecoflap_zo_perturb_units call per layer. */
int ecoflap_zo_perturb_layers(const int64_t* table, int n_layers, int64_t total_rows,
                              int dtype, float zo_eps, void* stream,
                              void* start_event, void* stop_event);

/* The same with z SUPPLIED per unit (parity mode: the caller draws each unit's z as the
 * reference does, torch.manual_seed(seed) + torch.normal on the device,
 * layer_single_base_pruner.py:482-485, for every layer of the block up front — each draw
 * re-seeds, so the order of the draws does not matter).  Rows are [5 + 4*ECOFLAP_MAX_UNITS] wide:
 * the row above followed by z[MAX_UNITS] (device pointers to tensors of the layer's numel and
 * dtype, 16-byte aligned, non-NULL for every unit, owned or not: the drift needs them too);
 * seeds are ignored.  Bit-identical to one ecoflap_zo_perturb_units call with z per layer. */
int ecoflap_zo_perturb_layers_z(const int64_t* table, int n_layers, int64_t total_rows,
                                int dtype, float zo_eps, void* stream,
                                void* start_event, void* stop_event);

/* ---- the reference's own draw, in registers ------------------------------------------------
 * layer_single_base_pruner.py:482-485 draws z with torch.manual_seed(seed) followed by ONE
 * torch.normal(0, 1, size, device = param's device, dtype = param's dtype).  On a GPU that is
 * ATen's distribution_elementwise_grid_stride_kernel<float, 4> over rocRAND's Philox4x32-10
 * (torch/include/ATen/native/cuda/DistributionTemplates.h:50-100): `threads` = 256 * grid
 * threads, thread idx takes rocrand_normal4 number j of Philox(counter {j, 0, idx, 0}, key =
 * seed) and writes it to elements idx + threads * (4j + 0..3).  The three entry points below
 * regenerate exactly that stream inside K1 — the same integer stream, the same Box-Muller
 * instruction sequence as the kernel in libtorch_hip.so (csrc/zo_perturb.hip restates it from
 * the disassembly), the same single rounding to `dtype` — so z costs no HBM traffic and no
 * library launch, and the outputs equal the `z` != NULL forms fed torch's own tensor bit for bit
 * (tests/test_gpu_parity.py; probed at start-up by pruners/layer_sparsity.py, which falls back to
 * the materialised draw when a torch / rocRAND upgrade changes the stream).
 * threads: what ATen's calc_execution_policy gives for n on the device at hand:
 * ecoflap_torch_normal_threads(n, multiProcessorCount, maxThreadsPerMultiProcessor).
 * n < 2^31 (ECOFLAP_ESIZE otherwise): a larger tensor torch draws in several launches
 * (TensorIterator::with_32bit_indexing), each at its own Philox offset; the caller takes the
 * materialised draw for it (pruners/layer_sparsity.py does). */
int64_t ecoflap_torch_normal_threads(int64_t n, int multiprocessors, int max_threads_per_mp);

/* z_out <- the tensor torch.manual_seed(seed); torch.normal(0, 1, [n], dtype) returns. */
int ecoflap_zo_fill_normal_torch(void* z_out, int64_t n, int dtype, uint64_t seed,
                                 int64_t threads, void* stream);

/* ecoflap_zo_perturb with z = that draw (one reference K1 call, in place). */
int ecoflap_zo_perturb_torch(void* w, int64_t n, int dtype, float scaling_factor, float zo_eps,
                             uint64_t seed, int64_t threads, void* stream);

/* Work items of one layer in ecoflap_zo_perturb_layers_torch's table: a tensor of n elements
 * drawn by `threads` threads is R = ceil(n / (4*threads)) rounds; a full round is
 * wpr = ceil(threads / (16/s) / 64) items (one wave each: the four 16-byte vectors per lane that
 * one Philox call per element column feeds), of the last round only the ceil(v / 64) items that
 * hold one of the v vectors of its first row (the rest are empty; torch's own kernel draws and
 * discards there), at least 1 (item 0 also carries a ragged tail).  0 for invalid arguments. */
int64_t ecoflap_torch_layer_items(int64_t n, int64_t threads, int dtype);

/* ecoflap_zo_perturb_layers with z = that draw per unit.  table: DEVICE int64
 * [n_layers][6 + 3*ECOFLAP_MAX_UNITS], row = {w_in, w_final, numel, n_units, first_item, threads,
 * seeds[MAX_UNITS], w_plus[MAX_UNITS], w_minus[MAX_UNITS]}; items_l =
 * ecoflap_torch_layer_items(numel, threads, dtype);
 * first_item = sum of the earlier layers' items, total_items = the sum over all.
 * w_final may equal w_in and a unit's w_plus may alias w_in (a lane reads its elements before it
 * writes them), so this one entry point also serves the units form (one layer, in place) and the
 * triple form (one unit).  Algorithmic bytes: (2*owned_units + 2) * s per element: no z bytes. */
int ecoflap_zo_perturb_layers_torch(const int64_t* table, int n_layers, int64_t total_items,
                                    int dtype, float zo_eps, void* stream,
                                    void* start_event, void* stop_event);

/* Self-check of the Box-Muller radius those kernels compute.  rocRAND's radius of a 32-bit word
 * x is s(x) = sqrtf(-2 logf((x + 1) * 2^-32)) through ocml's logf and correctly rounded sqrtf;
 * the kernels reach the same fp32 value with a shorter instruction sequence that is exact on
 * this function's domain, not in general (csrc/zo_perturb.hip: torch_radius vs
 * torch_radius_reference).  The domain is enumerable: this entry point evaluates both for the
 * words [first_word, first_word + n_words) on the device and ADDS the number of words whose
 * results differ in any bit to *differ (device memory, zeroed by the caller).  The whole range
 * (first_word = 0, n_words = 2^32) runs in well under a second; tests/test_torch_stream.py
 * asserts 0. */
int ecoflap_zo_torch_radius_sweep(uint64_t first_word, uint64_t n_words,
                                  unsigned long long* differ, void* stream);

/* An empty kernel through the same instrumented launch: the floor of that event pair. */
int ecoflap_null_launch_timed(void* stream, void* start_event, void* stop_event);

/* Materialise the in-register z stream of K1 for (seed, n, dtype). */
int ecoflap_zo_fill_normal(void* z_out, int64_t n, int dtype, uint64_t seed,
                           void* stream);

/* Raw Philox4x32-R words (counter = i/4, lane = i%4) — integer, bit-exact
 * against oracle/ecoflap_oracle.c:oracle_philox4x32. */
int ecoflap_philox_u32(uint32_t* out, int64_t n, uint64_t seed, void* stream);

/* ---------------------------------------------------------------------------
 * K3+K4  fused |grad| (x) |W| per-layer reduction
 * replaces the accumulate + product + .sum() chain
 *   layer_single_base_pruner.py:446-471 (first order), :551-559 (MEZO-GradMag*)
 *   and the per-layer `.sum()` at :370
 * out_accum[0] += sum_e f(w_e, g_e)   (double, device memory; the caller
 * zeroes it once per layer and divides by the batch count afterwards).
 * Deterministic: fixed per-block partials, fixed-order final sum.
 * ------------------------------------------------------------------------- */
size_t ecoflap_absprod_reduce_workspace_bytes(int64_t n);
int ecoflap_absprod_reduce(const void* w, const void* g, int64_t n,
                           int dtype_w, int dtype_g, int mode,
                           double* out_accum, void* workspace,
                           size_t workspace_bytes, void* stream);

/* Multi-tensor form: one launch over every prunable matrix of the model.
 * table: device array of n_layers rows {w_ptr, g_ptr, numel} (int64 each).
 * out_accum: device double[n_layers], accumulated into. */
size_t ecoflap_absprod_reduce_multi_workspace_bytes(int n_layers);
int ecoflap_absprod_reduce_multi(const int64_t* table, int n_layers,
                                 int64_t max_numel, int dtype_w, int dtype_g,
                                 int mode, double* out_accum, void* workspace,
                                 size_t workspace_bytes, void* stream);

/* The same over matrices of DIFFERENT dtypes in one launch (BLIP-2: fp16 ViT-g, bf16 FlanT5, fp32
 * Q-Former — the single-dtype form needs one launch per class).  Rows of FOUR words
 * {w_ptr, g_ptr, numel, dtype_w | dtype_g << 8}; `table_host` is the same table in host memory (the
 * dtype words are validated there).  Workspace: ecoflap_absprod_reduce_multi_workspace_bytes. */
int ecoflap_absprod_reduce_mixed(const int64_t* table, const int64_t* table_host, int n_layers,
                                 int mode, double* out_accum, void* workspace,
                                 size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * K6  Wanda calibration statistic (running mean of per-input-channel sum x^2)
 * replaces WrappedGPT.add_batch   LAVIS/lavis/compression/pruners/wanda_pruner.py:71-84
 *     scaler_row *= n/(n+b); n += b; scaler_row += norm(x, 2, dim=tokens)**2 / n
 * x: [tokens, cols] row-major of `dtype`; scaler_row: float[cols].
 * ONE launch per call: the last row chunk of every column block to finish (ticket counters at
 * the head of the workspace) sums the block's partials in fixed order and applies the update.
 * The workspace must be ZEROED ONCE by its owner (hipMemset after allocation); the tickets
 * reset themselves, so it can be reused by later calls — on the same stream, one call at a
 * time — without another memset.
 * ------------------------------------------------------------------------- */
size_t ecoflap_colsqnorm_workspace_bytes(int64_t tokens, int64_t cols);
int ecoflap_colsqnorm_accum(float* scaler_row, const void* x, int64_t tokens,
                            int64_t cols, int dtype, int64_t nsamples_before,
                            int64_t batch, void* workspace,
                            size_t workspace_bytes, void* stream);
/* Same update with `n` (WrappedGPT.nsamples, :58/:81) held in device memory: reads
 * *nsamples_dev as nsamples_before and adds `batch` to it afterwards.  No launch argument
 * changes between calls, so a block forward with its hooks can be captured once as a HIP
 * graph and replayed per calibration sample. */
int ecoflap_colsqnorm_accum_dev(float* scaler_row, const void* x, int64_t tokens,
                                int64_t cols, int dtype, int64_t* nsamples_dev,
                                int64_t batch, void* workspace,
                                size_t workspace_bytes, void* stream);

/* All hooked inputs of ONE transformer block for one calibration sample in ONE launch
 * (the reference's per-Linear forward hooks, wanda_pruner.py:240-252 / :521-533, each call
 * add_batch on their own input): up to ECOFLAP_COLSQ_MAX_ITEMS inputs of one dtype, the
 * same arithmetic per input as ecoflap_colsqnorm_accum[_dev].
 *   nsamples_dev != NULL: the device form (count read from / bumped in device memory);
 *   raw != 0: scaler_row[c] = norm(x_c, 2)**2 of THIS input alone — no running mean, the
 *             counts are neither read nor bumped (data-parallel stage 2: per-batch rows are
 *             exchanged and replayed in global batch order by ecoflap_colsq_replay).
 * One workspace for the call (zeroed once by its owner, like the one-input form). */
#define ECOFLAP_COLSQ_MAX_ITEMS 16
typedef struct {
    float* scaler_row;          /* float[cols] */
    const void* x;              /* [tokens, cols] row-major */
    int64_t tokens, cols;
    int64_t nsamples_before;    /* host form (ignored when nsamples_dev != NULL) */
    int64_t* nsamples_dev;      /* device form, or NULL */
    int64_t batch;              /* leading dimension of the hook's input (W:72-75) */
    int raw;
} ecoflap_colsq_item;
size_t ecoflap_colsqnorm_multi_workspace_bytes(const ecoflap_colsq_item* items, int n);
int ecoflap_colsqnorm_accum_multi(const ecoflap_colsq_item* items, int n, int dtype,
                                  void* workspace, size_t workspace_bytes, void* stream);
/* The running mean of wanda_pruner.py:80-84 replayed over per-batch statistics:
 *   for j in 0..n_batches-1:  row = row * (n/(n+b_j)) + sq[j][c] / (n+b_j);  n += b_j
 * with exactly the roundings of the fused update above, so batches reduced on different
 * ranks give the one-process scaler_row bit for bit.  sq: float[n_batches][ld] (device),
 * batches_dev: int64[n_batches] (device). */
int ecoflap_colsq_replay(float* scaler_row, const float* sq, const int64_t* batches_dev,
                         int n_batches, int64_t cols, int64_t ld, int64_t nsamples_before,
                         void* stream);

/* ---------------------------------------------------------------------------
 * K7  Wanda metric + selection + zeroing
 * metric = |W| * sqrt(scaler_row)  (fp32)          wanda_pruner.py:260, :541
 *  - rows mode (T5/BERT):  per row, stable ascending sort, zero the first k
 *    columns (ties: lower column index first)       wanda_pruner.py:272-279
 *  - matrix mode (ViT): thres = sorted(flatten)[k]; zero metric <= thres
 *                                                   wanda_pruner.py:555-558
 * mask_out (optional): uint8[rows*cols], 1 where zeroed.
 * ------------------------------------------------------------------------- */
/* Matrix mode, asynchronous and graph-capturable.  Matrices of 16384 .. 48 Mi 16-byte-aligned vectors'
 * worth of elements take the sampled-bracket selection (2 reads + 1 write of W in two launches: a
 * sample taken by every workgroup of the counting pass brackets the threshold, the pass counts
 * below / histograms inside the bracket exactly, the apply pass settles all but the threshold bin's
 * ~1000 elements, which its last workgroup sorts; every count exact; a matrix the passes cannot
 * settle — bracket miss, massive ties, non-finite threshold — is finished exactly by one workgroup
 * inside the apply pass, see ecoflap_wanda_fallback_counts).  Others, and every matrix when
 * ECOFLAP_WANDA_SAMPLED=0 is in the environment (read at every call): three histogram passes
 * (11 + 11 + 10 bits) for the k-th order statistic, then `metric <= thres`.  Same result, bit for
 * bit, either way. */
size_t ecoflap_wanda_workspace_bytes(int64_t rows, int64_t cols);
int ecoflap_wanda_prune_rows(void* w, const float* scaler_row, int64_t rows,
                             int64_t cols, int dtype, int64_t k,
                             uint8_t* mask_out, void* workspace,
                             size_t workspace_bytes, void* stream);
int ecoflap_wanda_prune_matrix(void* w, const float* scaler_row, int64_t rows,
                               int64_t cols, int dtype, int64_t k,
                               uint8_t* mask_out, void* workspace,
                               size_t workspace_bytes, void* stream);

/* Structured n:m branch of both Wanda pruners (wanda_pruner.py:265-270, :546-551; dead in the
 * reference's shipped configs, prune_n = 0): in every group of m consecutive columns of a row the n
 * smallest |W| * sqrt(scaler_row) are zeroed (`torch.topk(..., largest=False)`: NaN counts as the
 * largest; equal metrics: the lower column first).  0 < n <= m <= 16; a ragged last group
 * (cols % m != 0) is selected among its own elements and must hold at least n (the reference's topk
 * raises otherwise: ECOFLAP_ESIZE).  No workspace, asynchronous. */
int ecoflap_wanda_prune_nm(void* w, const float* scaler_row, int64_t rows, int64_t cols, int dtype,
                           int n, int m, uint8_t* mask_out, void* stream);

/* Block-level form: every Linear of one transformer block in ONE call (the reference prunes
 * them one after another inside its per-block loop, wanda_pruner.py:253-283 / :534-562; the
 * selections are independent).  Same results as n_items single calls, bit for bit; the launches
 * are shared: one sqrt launch (it also clears the matrix-mode selection state: the workspace
 * need not be zeroed), ONE selection grid for the rows-mode items of a dtype when their rows are
 * at most 8192 (fp16 / bf16) or 4096 (fp32) columns wide (rows of up to 256 16-byte vectors one
 * wave each, longer rows one workgroup each), three histogram launches + one apply launch for
 * all matrix-mode items of a dtype. */
#define ECOFLAP_WANDA_MAX_ITEMS 16
#define ECOFLAP_WANDA_ROWS   0
#define ECOFLAP_WANDA_MATRIX 1
typedef struct ecoflap_wanda_item {
    void* w;                   /* [rows, cols] of `dtype`, pruned in place */
    const float* scaler_row;   /* float[cols] */
    int64_t rows, cols;
    int64_t k;                 /* rows mode: columns zeroed per row; matrix mode: index into the sort */
    uint8_t* mask_out;         /* optional uint8[rows*cols], 1 where zeroed */
    int dtype;
    int mode;                  /* ECOFLAP_WANDA_ROWS / ECOFLAP_WANDA_MATRIX */
} ecoflap_wanda_item;
size_t ecoflap_wanda_block_workspace_bytes(const ecoflap_wanda_item* items, int n_items);
int ecoflap_wanda_prune_block(const ecoflap_wanda_item* items, int n_items,
                              void* workspace, size_t workspace_bytes, void* stream);
/* Diagnostics of the matrix-mode selection: how many matrices since the last reset were finished by
 * the exact on-device fallback instead of the sampled two-pass selection — out4[1]: the sample's
 * bracket missed the threshold or its bin reaches non-finite bit patterns, out4[2]: the
 * threshold's bin was too crowded (massive ties).  Results are identical either way; the fallback
 * streams the matrix from one workgroup (~1 ms for 8 M elements).  Synchronises the device. */
int ecoflap_wanda_fallback_counts(unsigned int* out4, int reset);

/* ---------------------------------------------------------------------------
 * SparseGPT block step
 * replaces the inner block of SparseGPT.fasterprune
 *   LAVIS/lavis/compression/pruners/sparsegpt_pruner.py:172-216
 * for columns [i1, i1+count), count <= 128, of the fp32 working copy W[rows, ldw]:
 *   threshold of W1**2/diag(Hinv1)**2 at rank k = int(rows*count*sparsity) (:186-188),
 *   the sequential OBS sweep (:192-210), W[:, i1:i1+count] = Q1 (:212) and Err1 into err_out
 *   (the caller applies W[:, i2:] -= Err1 @ Hinv[i1:i2, i2:], :216, with a library GEMM).
 * Hinv: float[cols, ldh], the upper Cholesky factor the reference calls Hinv (:162).
 * mask_in (optional) replaces the threshold by a given uint8[rows*count] mask.
 * ------------------------------------------------------------------------- */
size_t ecoflap_sparsegpt_workspace_bytes(void);
int ecoflap_sparsegpt_block(float* W, int64_t rows, int64_t ldw, const float* Hinv,
                            int64_t ldh, int64_t i1, int count, int64_t k,
                            const uint8_t* mask_in, float* err_out, uint8_t* mask_out,
                            void* workspace, size_t workspace_bytes, void* stream);

/* The same block step under n:m (sparsegpt_pruner.py:190, :196-198; `prune_n` is 0 in every shipped
 * config): no threshold — at every block column i with i % m == 0 the n smallest
 * W1[:, i:i+m]**2 / diag(Hinv1)[i:i+m]**2 of each row, taken on the sweep's CURRENT values, join
 * the mask (`torch.topk(..., largest=False)`: NaN counts as the largest; equal values: the lower
 * column first).  0 < n <= m <= 16; no workspace. */
int ecoflap_sparsegpt_block_nm(float* W, int64_t rows, int64_t ldw, const float* Hinv, int64_t ldh,
                               int64_t i1, int count, int n, int m, float* err_out,
                               uint8_t* mask_out, void* stream);

/* SparseGPT Hessian accumulation, the path's one GEMM-shaped
 * contraction, on the matrix cores:
 * replaces SparseGPT.add_batch   LAVIS/lavis/compression/pruners/sparsegpt_pruner.py:71-82
 *     H *= n/(n+b); n += b; inp = sqrt(2/n) * x.float().t(); H += inp @ inp.t()
 * as  H <- (n/(n+b)) * H + (2/(n+b)) * X^T X  with X = [tokens, cols] of fp16 / bf16 (the
 * Linear's input as the forward produced it under autocast; an fp32 X returns ECOFLAP_EDTYPE:
 * the caller keeps the library GEMM for it), fp32 accumulation in v_mfma_f32_32x32x16_*,
 * upper-triangle tiles computed once and mirrored.  H: float[cols, cols], both triangles kept.
 * Agreement with the reference's fp32 expression: 1e-5 relative (products are exact, the sums
 * re-associate). */
size_t ecoflap_hessian_workspace_bytes(int64_t tokens, int64_t cols);
int ecoflap_hessian_accum(float* H, const void* x, int64_t tokens, int64_t cols,
                          int dtype, int64_t nsamples_before, int64_t batch,
                          void* workspace, size_t workspace_bytes, void* stream);

/* SparseGPT's factorisations:
 * replaces torch.linalg.cholesky(H) / torch.linalg.cholesky(Hinv, upper=True)
 *     LAVIS/lavis/compression/pruners/sparsegpt_pruner.py:113-123, :146-155
 * a <- the Cholesky factor of the symmetric fp32 matrix a[n, lda] (row-major, IN PLACE: copy
 * first to keep the input, as the damped retry loop of the caller must; the LOWER triangle of the
 * input is read).  upper = 0: L with a = L L^T, strict upper triangle zeroed; upper = 1: U = L^T
 * with a = U^T U, strict lower triangle zeroed (torch.linalg.cholesky's two forms).
 * *info (device int): 0, or LAPACK's potrf convention — the 1-based index of the first pivot
 * that was not positive (or not a number): the matrix is not numerically positive definite and
 * the contents of a are undefined.  workspace: ecoflap_cholesky_workspace_bytes() (16 KB) of
 * caller-owned device scratch PER CALL IN FLIGHT (the factored diagonal block waits there for the
 * end of its panel launch).  No handle, nothing else shared between calls: factorisations may be
 * in flight side by side on several streams, from several host threads.  Blocked right-looking, 64 columns per step,
 * trailing updates on v_mfma_f32_32x32x2_f32, every sum in a fixed order (bit-repeatable).
 * Agreement with rocSOLVER's factor: a few 1e-7 relative (fp32 factorisations re-associate). */
size_t ecoflap_cholesky_workspace_bytes(void);
int ecoflap_cholesky_f32(float* a, int64_t n, int64_t lda, int upper, int* info, void* workspace,
                         size_t workspace_bytes, void* stream);

/* replaces torch.cholesky_inverse(L)   sparsegpt_pruner.py:134
 * out[n, ldo] <- (L L^T)^-1, both triangles, from the LOWER Cholesky factor l[n, ldl] (fp32,
 * row-major; only its lower triangle is read; out may not alias l).  X = L^-1 by the 64 x 64
 * diagonal blocks (one wave each, in registers) and doubling below them (X21 = -X22 L21 X11, tile
 * GEMMs on v_mfma_f32_32x32x2_f32), then out = X^T X.  workspace: caller-owned scratch of
 * ecoflap_cholesky_inverse_workspace_bytes(n) bytes (1.5 n^2 floats), contents undefined
 * afterwards.  Fixed summation order; nothing shared between calls. */
size_t ecoflap_cholesky_inverse_workspace_bytes(int64_t n);
int ecoflap_cholesky_inverse_f32(const float* l, int64_t n, int64_t ldl, float* out, int64_t ldo,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * "Real-*" global iterative pruning
 * replaces, in layer_single_base_pruner.py:156-245 / :446-471, the per-element accumulator
 * `acc += |g|` (:455), the score (:461-469) times the previous mask (:221-224), the global
 * threshold = k-th smallest of ALL scores (torch.topk over the concatenation, :170-175), the
 * new mask and `W *= mask` (:180, :229-231), and the per-parameter zero fraction (:236-237).
 * Tables are device int64 arrays; dtype codes are per row (mixed fp16 / bf16 models).
 *   grad_accum rows:  {acc_ptr(float), g_ptr, numel, dtype_g}
 *   threshold rows:   {w_ptr, acc_ptr(float), mask_ptr(uint8, 1 = kept), numel, dtype_w}
 *   count_zeros rows: {w_ptr, numel, dtype_w}
 * mode: 0 |W|*|acc/n|, 1 W^2*(acc/n), 2 |acc/n|, 3 the signed weight itself (acc_ptr unused;
 * BLIPT5GlobalMagPruner, global_pruner.py:116-142 + :251 — the same get_mask over
 * `v.data.float()`, no abs, as shipped);  k = num_to_zero_out (1-indexed rank).
 * One call = one threshold over the rows given: all layers (global), one sub-model's layers
 * (global_pruner.py:183-192) or a single layer (get_layerwise_mask, :144-157).
 * ------------------------------------------------------------------------- */
int ecoflap_grad_accum_multi(const int64_t* table, int n_layers, void* stream);
size_t ecoflap_global_prune_workspace_bytes(void);
int ecoflap_global_threshold_prune(const int64_t* table, int n_layers, int mode,
                                   float n_batches, int64_t k, int64_t total_numel,
                                   void* workspace, size_t workspace_bytes, void* stream);
/* The same with get_mask's protection step (:160-167): per layer, scores >= its num_to_set-th
 * largest (num_to_set = int(numel * (1 - max_sparsity_per_layer))) are raised to finfo.max before
 * the global threshold is taken.  protect_ranks: device int64[n_layers], the 1-indexed rank from
 * the smallest of that per-layer threshold (numel - num_to_set + 1; 0 = nothing protected). */
size_t ecoflap_global_prune_protected_workspace_bytes(int n_layers);
int ecoflap_global_threshold_prune_protected(const int64_t* table, int n_layers, int mode,
                                             float n_batches, int64_t k, int64_t total_numel,
                                             const int64_t* protect_ranks,
                                             void* workspace, size_t workspace_bytes, void* stream);
int ecoflap_count_zeros_multi(const int64_t* table, int n_layers, int64_t* out_counts,
                              void* stream);

/* ---------------------------------------------------------------------------
 * K8  mask apply in masked fine-tuning:  grad *= mask
 * replaces UPop/ecoflap_compression_vqa.py:124-129
 * keep_mask: uint8[n], 1 = keep (multiply by 1), 0 = pruned (multiply by 0).
 * ------------------------------------------------------------------------- */
int ecoflap_mask_mul(void* g, const uint8_t* keep_mask, int64_t n, int dtype,
                     void* stream);

/* ---------------------------------------------------------------------------
 * K5  sparsity allocator (host, no GPU): keep-counts per group
 * replaces LayerSparsity.compute_the_sparsity_per_group
 *   layer_single_base_pruner.py:247-314, replaying its mixed
 *   float32 / int32 / int64 tensor arithmetic op for op.
 * group_scores: float[n_groups]; group_num_params: int64[n_groups].
 * out_sparsity: float[n_groups] (fp32-valued, as `.item()` of a float tensor)
 * out_keep (optional): double[n_groups] — the final keep vector.
 * ------------------------------------------------------------------------- */
int ecoflap_allocate_sparsity(const float* group_scores,
                              const int64_t* group_num_params, int n_groups,
                              int64_t total_parameters_to_keep,
                              double max_sparsity_per_layer,
                              float* out_sparsity, double* out_keep);

#ifdef __cplusplus
}
#endif
#endif /* ECOFLAP_HIP_H */
