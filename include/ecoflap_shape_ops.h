/*
 * ecoflap_shape_ops.h — fused forward ops of the synthetic shape modules (plumbing).
 *
 * NOT part of the pruner drop-in boundary (that is ecoflap_hip.h).  The scoring loop is
 * forward-bound, and a third of a BLIP-2-shaped forward's GPU time was eight-kernel
 * RMSNorm chains and other element-wise glue; these entry points fuse them for the
 * build's own shape modules (ecoflap_amd/shapes/).  Same conventions as ecoflap_hip.h.
 */
#ifndef ECOFLAP_SHAPE_OPS_H
#define ECOFLAP_SHAPE_OPS_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* T5LayerNorm (LAVIS/lavis/models/blip2_models/modeling_t5.py:239-262):
 *   var = mean(float(x)^2); h = bf16/f16(x * rsqrt(var + eps)); y = w * h
 * x, y: [rows, d] of `dtype` (F16/BF16), w: [d] of `dtype`; d % 8 == 0. */
int ecoflap_t5_rmsnorm(const void* x, const void* w, void* y, int64_t rows, int64_t d,
                       float eps, int dtype, void* stream);
/* The same preceded by the residual add that produces its input (T5 block: h = x + sublayer(x),
 * then the NEXT sublayer's norm of h): sum_out = dtype(x + residual), y = rmsnorm(sum_out) * w —
 * the bits of the separate add and norm, one pass instead of two. */
int ecoflap_t5_add_rmsnorm(const void* x, const void* residual, const void* w, void* sum_out, void* y,
                           int64_t rows, int64_t d, float eps, int dtype, void* stream);

/* gated GELU of T5DenseGatedActDense (modeling_t5.py:296-330): y = gelu(a) * b
 * (erf GELU in fp32, rounded to dtype, then the product rounded to dtype). n % 8 == 0. */
int ecoflap_gelu_mul(const void* a, const void* b, void* y, int64_t n, int dtype, void* stream);

/* nn.LayerNorm on a 16-bit activation under autocast (EVA ViT blocks,
 * LAVIS/lavis/models/eva_vit.py:160-184): fp32 statistics and fp32 affine parameters, 16-bit
 * in and out; with `residual` the block's residual add is fused in front:
 *   s = dtype(x + residual) -> sum_out;  y = dtype((float(s) - mean) * rstd * w + b)
 * x, residual, sum_out, y: [rows, d] of `dtype` (F16/BF16); w, b: [d] float; d % 8 == 0.
 * residual == NULL: plain LayerNorm of x (sum_out unused). */
int ecoflap_add_layernorm(const void* x, const void* residual, const float* w, const float* b,
                          void* sum_out, void* y, int64_t rows, int64_t d, float eps, int dtype,
                          void* stream);

/* The consumers of a Linear output that arrives WITHOUT its bias (the pinned GEMM solutions of
 * libecoflap_gemm.so have no bias epilogue on gfx950: every bias-epilogue solution of the library
 * there is a Stream-K kernel).  `bias` is the Linear's own bias, [d] of `dtype`; the first step is
 * always dtype(out + bias), the value the Linear would have returned from a rounded accumulator:
 *   ecoflap_add_bias_layernorm: ecoflap_add_layernorm with residual' = dtype(residual + bias)
 *   ecoflap_bias_gelu:          y = dtype(gelu(dtype(a + bias)))        (erf GELU, EVA Mlp)
 *   ecoflap_bias_add_residual:  y = dtype(x + dtype(m + bias))          (EVA Block's MLP residual) */
int ecoflap_add_bias_layernorm(const void* x, const void* residual, const void* residual_bias,
                               const float* w, const float* b, void* sum_out, void* y,
                               int64_t rows, int64_t d, float eps, int dtype, void* stream);
int ecoflap_bias_gelu(const void* a, const void* bias, void* y, int64_t rows, int64_t d, int dtype,
                      void* stream);
int ecoflap_bias_add_residual(const void* x, const void* m, const void* bias, void* y, int64_t rows,
                              int64_t d, int dtype, void* stream);

/* EVA attention bias (eva_vit.py:123-128): qkv += cat(q_bias, zeros, v_bias).to(dtype), in
 * place; qkv: [rows, 3*dim] of `dtype` (F16/BF16), q_bias / v_bias: [dim] float; dim % 8 == 0.
 * (The pinned-forward qkv Linear of shapes/eva_vit.py adds this bias in its GEMM's epilogue
 * instead; this pass serves the other cases.) */
int ecoflap_qkv_bias_add(void* qkv, const float* q_bias, const float* v_bias, int64_t rows,
                         int64_t dim, int dtype, void* stream);

/* Multi-head self-attention of the EVA ViT blocks (LAVIS/lavis/models/eva_vit.py:119-141 without
 * the relative position bias BLIP-2's tower does not use): out = softmax(q k^T * scale) v per
 * (image, head).  qkv: [batch, tokens, 3, heads, head_dim] of F16 as the qkv Linear wrote it;
 * out: [batch, tokens, heads * head_dim].  tokens <= 288, head_dim <= 96 and a multiple of 8.
 * fp32 scores / softmax, probabilities rounded to f16 for the second product. */
int ecoflap_vit_attention(const void* qkv, void* out, int64_t batch, int64_t tokens,
                          int64_t heads, int64_t head_dim, float scale, int dtype, void* stream);

/* Up to ECOFLAP_COPY_MAX_ITEMS device-to-device copies (non-overlapping, any alignment) in
 * one launch: the loop's state hand-overs between static graph buffers. */
#define ECOFLAP_COPY_MAX_ITEMS 32
typedef struct {
    void* dst;
    const void* src;
    int64_t bytes;
} ecoflap_copy_item;
int ecoflap_multi_copy(const ecoflap_copy_item* items, int n, void* stream);
/* Bitwise comparison of up to ECOFLAP_COPY_MAX_ITEMS pairs (dst vs src of each item) in one
 * launch: *mismatch_flag |= 1 when any byte differs (the caller zeroes it). */
int ecoflap_multi_compare(const ecoflap_copy_item* items, int n, int* mismatch_flag, void* stream);

/* y[M,N] = x[M,K] W[N,K]^T (+ bias[N]) in fp32 on v_mfma_f32_32x32x2_f32 (csrc/gemm_f32.hip): the
 * forward's fp32 Linears (BLIP-2's Q-Former), for which hipBLASLt has only Stream-K solutions on
 * gfx950.  Every output element is one fp32 accumulation chain in a fixed k order, the bias added
 * last: a row's result does not depend on the rows that travel with it, nor on M.  N % 128 == 0, K % 32 == 0 (else
 * ECOFLAP_ESIZE: the caller keeps the framework's GEMM); x, w 16-byte aligned. */
int ecoflap_linear_f32(const float* x, const float* w, const float* bias, float* y, int64_t M,
                       int64_t N, int64_t K, void* stream);

/* ---- libecoflap_gemm.so (csrc/gemm_pinned.hip) ------------------------------------------------
 * y[M,N] = x[M,K] W[N,K]^T (+ bias[N]) for F16 / BF16 (fp32 accumulation) through hipBLASLt with
 * the SOLUTION PINNED per (N, K, dtype, bias): chosen once, among the library's own candidates
 * for the weight shape, by name (no Stream-K, no K split over workgroups) and by measurement
 * (same bits on a repeated call; the probe problem alone == its first and its last slot of a
 * 16-slot problem), then used for every M.  One macro tile and one K order for every row count:
 * a row's result does not depend on how many rows travel with it (batch invariance by
 * construction) nor on an environment variable of the library.
 *
 * ecoflap_linear_pinned_plan: choose (or look up) the plan; allocates and synchronises, so call
 *   it OUTSIDE stream capture, before the first ecoflap_linear_pinned of that weight shape.
 *   m_probe = a row count the loop uses (the check runs at m_probe and 16 * m_probe).
 *   -> 0 and the solution's index / name (the lowest index among the survivors: no timing enters
 *   the choice), its time on the 16-slot problem (best_us)
 *   and, for the record, the time of the library's own first choice on the same data
 *   (default_us), or ECOFLAP_ESIZE when no candidate survived.
 *   Bias: gfx950's bias-epilogue solutions are all Stream-K kernels, so the solution is a
 *   no-bias one; ecoflap_linear_pinned with a bias prefills the output with the bias rows and runs
 *   beta = 1 (one extra write + read of the output); the shape modules instead leave the bias to
 *   the op that consumes the output (ecoflap_add_bias_layernorm / _bias_gelu / _bias_add_residual).
 * ecoflap_linear_pinned: the product; launches only (safe under capture once planned and once
 *   called with this M).  bias_dtype: dtype code of `bias` (the Linear's own dtype, or F32).
 *   workspace: >= 64 MiB recommended (the pinned solutions take none or little).
 *   ECOFLAP_EMODE: no plan for this weight shape. */
int ecoflap_linear_pinned_plan(int64_t m_probe, int64_t N, int64_t K, int dtype, int has_bias,
                               int bias_dtype, int* solution_index, int* tried, int* passed,
                               float* best_us, float* default_us, char* name_out, int name_len);
int ecoflap_linear_pinned(const void* x, const void* w, const void* bias, void* y, int64_t M,
                          int64_t N, int64_t K, int dtype, int bias_dtype, void* workspace,
                          size_t workspace_bytes, void* stream);
/* Measurement only (tools/tune_gemm.py): every solution of the library that supports the problem
 * at m_probe and 16 * m_probe rows (heuristic list + all algorithms, no name filter; bias in the
 * GEMM's epilogue when has_bias), each timed on both, the `top` fastest by t(16 m) + t(m) checked
 * bit for bit for repeatability (flag 1) and batch invariance (flag 2); flag 4 = the heuristic's
 * first choice (always listed).  Rows fastest first; names packed at name_len bytes each. */
int ecoflap_linear_tune(int64_t m_probe, int64_t N, int64_t K, int dtype, int has_bias, int bias_dtype,
                        int top, int* index_out, float* us_big_out, float* us_small_out, int* flags_out,
                        char* names_out, int name_len, int* n_out, int* n_candidates);
/* "<hipblasLtGetVersion>-<git revision>" of the hipBLASLt serving this process: a solution index
 * means something only together with it (recorded by the run summaries). */
int ecoflap_linear_library_version(char* out, int len);

#ifdef __cplusplus
}
#endif
#endif
