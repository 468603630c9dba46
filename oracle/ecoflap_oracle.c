/*
 * ecoflap_oracle.c — CPU restatement of the ECoFLaP hot-path arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * Parity status: PINNED.  Every function here is checked against golden
 * vectors produced by importing the reference's own Python pruners in the
 * build container (tests/golden/make_golden.py -> tests/golden/g*.npz;
 * tests/test_oracle_golden.py).  The reference has no tests of its own for
 * this path.
 *
 * Citations are relative to the reference checkout (ylsung/ECoFLaP):
 *   P  = LAVIS/lavis/compression/pruners/layer_single_base_pruner.py
 *   W  = LAVIS/lavis/compression/pruners/wanda_pruner.py
 *
 * torch semantics restated: every elementwise torch op on an fp16/bf16 tensor
 * computes in fp32 and rounds the result to the storage dtype (round to
 * nearest even); ops are separate kernels, so nothing is fused (build with
 * -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define DT_F32 0
#define DT_F16 1
#define DT_BF16 2

/* ------------------------------------------------------------------ scalar conversions */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static inline float bf16_to_f(uint16_t h) { return u2f((uint32_t)h << 16); }
static inline uint16_t f_to_bf16(float f) {
    uint32_t u = f2u(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u); /* quiet NaN */
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static inline float f16_to_f(uint16_t h) {
    uint32_t s = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    if (e == 0) {
        if (m == 0) return u2f(s);
        float v = (float)m * 5.9604644775390625e-08f; /* 2^-24 */
        return (s ? -v : v);
    }
    if (e == 31) return u2f(s | 0x7f800000u | (m << 13));
    return u2f(s | ((e + 112u) << 23) | (m << 13));
}
static inline uint16_t f_to_f16(float f) {
    uint32_t u = f2u(f), s = (u >> 16) & 0x8000u, a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint16_t)(s | 0x7e00u);
    if (a >= 0x477ff000u) return (uint16_t)(s | 0x7c00u);      /* >= 65520 -> inf */
    if (a < 0x33000001u) return (uint16_t)s;                    /* <= 2^-25 -> 0 (ties to even) */
    if (a < 0x38800000u) {                                     /* subnormal half */
        /* value = a_f * 2^24 rounded to nearest even integer */
        float scaled = u2f(a) * 16777216.0f;                   /* exact: power of two */
        float r = nearbyintf(scaled);
        return (uint16_t)(s | (uint32_t)r);
    }
    uint32_t m = a & 0x7fffffu, e = (a >> 23) - 112u;
    uint32_t h = (e << 10) | (m >> 13), rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return (uint16_t)(s | h);
}

static inline float load_dt(const void* p, int64_t i, int dt) {
    if (dt == DT_F32) return ((const float*)p)[i];
    if (dt == DT_F16) return f16_to_f(((const uint16_t*)p)[i]);
    return bf16_to_f(((const uint16_t*)p)[i]);
}
static inline void store_dt(void* p, int64_t i, int dt, float v) {
    if (dt == DT_F32) ((float*)p)[i] = v;
    else if (dt == DT_F16) ((uint16_t*)p)[i] = f_to_f16(v);
    else ((uint16_t*)p)[i] = f_to_bf16(v);
}
static inline float round_dt(float v, int dt) {
    if (dt == DT_F32) return v;
    if (dt == DT_F16) return f16_to_f(f_to_f16(v));
    return bf16_to_f(f_to_bf16(v));
}

/* exported for tests of the conversions themselves */
void oracle_round_array(float* x, int64_t n, int dt) {
    for (int64_t i = 0; i < n; ++i) x[i] = round_dt(x[i], dt);
}

/* ------------------------------------------------------------------ Philox4x32-R
 * Published algorithm (Salmon et al., "Parallel random numbers: as easy as
 * 1, 2, 3", SC'11; Random123 philox.h).  The reference itself draws z with
 * torch.normal (P:485), whose stream is device specific; the build's
 * in-register generator is defined here instead and parity with the reference
 * is taken with z supplied explicitly. */
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u
void oracle_philox4x32(uint32_t out[4], const uint32_t ctr[4], const uint32_t key[2], int rounds) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < rounds; ++r) {
        uint64_t p0 = (uint64_t)PHILOX_M0 * c0, p1 = (uint64_t)PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0; k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
/* the build's stream: counter = (i/4, 0, 0), key = seed, `rounds` rounds */
void oracle_philox_u32(uint32_t* out, int64_t n, uint64_t seed, int rounds) {
    uint32_t r[4];
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (int64_t i = 0; i < n; ++i) {
        if ((i & 3) == 0) {
            const uint64_t q = (uint64_t)(i >> 2);
            const uint32_t ctr[4] = {(uint32_t)q, (uint32_t)(q >> 32), 0u, 0u};
            oracle_philox4x32(r, ctr, key, rounds);
        }
        out[i] = r[i & 3];
    }
}

/* ------------------------------------------------------------------ in-register normal streams
 * The build's own definition of z ~ N(0,1) "in param.dtype" (the reference draws it with
 * torch.normal, P:482-485, whose stream is device specific; parity with the reference is taken
 * with z supplied).  Restated step by step from ecoflap_amd/csrc/zo_perturb.hip:
 *   radius:  u1 = fma((float)a, 2^-32, 2^-33)  (fp32, exact steps)
 *            r  = sqrt_f32( -2 ln2 *_f32 log2_f32(u1) )
 *   angle:   N32: u2 = (float)b *_f32 2^-32 turns;  N16: u2 = h * 2^-16 turns (exact)
 *   z = r *_f32 cos(2 pi u2),  r *_f32 sin(2 pi u2);  then ONE rounding to the storage dtype.
 * log2 / sqrt / sin / cos are evaluated in double and rounded once to fp32 (correctly rounded
 * for all practical purposes); the GPU's v_log / v_sqrt / v_sin / v_cos units are accurate to
 * a few fp32 ulps, not correctly rounded, so the GPU test compares z32 within a stated
 * tolerance and the dtype-rounded stream exactly except where z32 sits on a rounding boundary.
 *   N32 (fp32 storage): element e <- Philox call e/4; words (w0,w1) -> e%4 = 0 (cos), 1 (sin);
 *        (w2,w3) -> 2, 3.
 *   N16 (fp16 / bf16):  vector v = e/8; block g = 64*(v/128) + v%64 = vectors (128*(v/128) +
 *        v%64, that + 64); calls 3g..3g+2 -> W[0..11]; pair j: radius W[j], angle halfword j of
 *        W[8..11] (low half first); position p = 8*((v/64)%2) + e%8 = 4q + t takes pair
 *        2q + (t&1), cos for t < 2, sin for t >= 2. */
static const double TWO_PI = 6.283185307179586476925286766559;
static float stream_radius(uint32_t a) {
    const float u1 = fmaf((float)a, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
    const float lg = (float)log2((double)u1);
    const float t = -1.3862943611198906f * lg;
    return (float)sqrt((double)t);
}
static void stream_block_words(uint32_t* w, int64_t first_call, int n_calls, uint64_t seed,
                               int rounds) {
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (int c = 0; c < n_calls; ++c) {
        const uint64_t q = (uint64_t)(first_call + c);
        const uint32_t ctr[4] = {(uint32_t)q, (uint32_t)(q >> 32), 0u, 0u};
        oracle_philox4x32(w + 4 * c, ctr, key, rounds);
    }
}
/* z32_out (optional): the fp32 value before the storage rounding; z_out (optional): `dt` */
void oracle_normal_stream(float* z32_out, void* z_out, int64_t n, int dt, uint64_t seed,
                          int rounds) {
    if (dt == DT_F32) {
        for (int64_t c = 0; 4 * c < n; ++c) {
            uint32_t w[4];
            float z[4];
            stream_block_words(w, c, 1, seed, rounds);
            for (int j = 0; j < 2; ++j) {
                const float u2 = (float)w[2 * j + 1] * 2.3283064365386963e-10f;
                const float r = stream_radius(w[2 * j]);
                z[2 * j + 0] = r * (float)cos(TWO_PI * (double)u2);
                z[2 * j + 1] = r * (float)sin(TWO_PI * (double)u2);
            }
            for (int i = 0; i < 4 && 4 * c + i < n; ++i) {
                if (z32_out) z32_out[4 * c + i] = z[i];
                if (z_out) ((float*)z_out)[4 * c + i] = z[i];
            }
        }
        return;
    }
    const int64_t nvec = (n + 7) / 8;                 /* a ragged tail still belongs to its vector */
    const int64_t nblocks = ((nvec + 127) / 128) * 64;
    for (int64_t g = 0; g < nblocks; ++g) {
        uint32_t w[12];
        float z[16];
        const int64_t v0 = (g / 64) * 128 + (g % 64);
        if (v0 >= nvec) continue;
        stream_block_words(w, 3 * g, 3, seed, rounds);
        for (int q = 0; q < 4; ++q) {
            const uint32_t ang = w[8 + q];
            const float ua = (float)(ang & 0xffffu) * 1.52587890625e-05f;
            const float ub = (float)(ang >> 16) * 1.52587890625e-05f;
            const float ra = stream_radius(w[2 * q]), rb = stream_radius(w[2 * q + 1]);
            z[4 * q + 0] = ra * (float)cos(TWO_PI * (double)ua);
            z[4 * q + 1] = rb * (float)cos(TWO_PI * (double)ub);
            z[4 * q + 2] = ra * (float)sin(TWO_PI * (double)ua);
            z[4 * q + 3] = rb * (float)sin(TWO_PI * (double)ub);
        }
        for (int p = 0; p < 16; ++p) {
            const int64_t e = (v0 + (p >= 8 ? 64 : 0)) * 8 + (p & 7);
            if (e >= n) continue;
            if (z32_out) z32_out[e] = z[p];
            if (z_out) store_dt(z_out, e, dt, z[p]);
        }
    }
}

/* ------------------------------------------------------------------ K1 (P:473-486)
 * param.data = param.data + scaling_factor * z * zo_eps, three roundings. */
static inline float k1_step(float w, float z, float sf, float eps, int dt) {
    float t = round_dt(z * sf, dt);    /* scaling_factor * z   */
    float u = round_dt(t * eps, dt);   /* (...) * zo_eps       */
    return round_dt(w + u, dt);        /* param.data + (...)   */
}
void oracle_zo_perturb(void* w, int64_t n, int dt, float sf, float eps, const void* z) {
    for (int64_t i = 0; i < n; ++i)
        store_dt(w, i, dt, k1_step(load_dt(w, i, dt), load_dt(z, i, dt), sf, eps, dt));
}
/* P:530-539: +1, -2, +1 with the same z; all three states from one read. */
void oracle_zo_perturb_triple(const void* w_in, void* w_plus, void* w_minus, void* w_rest,
                              int64_t n, int dt, float eps, const void* z) {
    for (int64_t i = 0; i < n; ++i) {
        float zz = load_dt(z, i, dt);
        float a = k1_step(load_dt(w_in, i, dt), zz, 1.0f, eps, dt);
        float b = k1_step(a, zz, -2.0f, eps, dt);
        float c = k1_step(b, zz, 1.0f, eps, dt);
        store_dt(w_plus, i, dt, a);
        store_dt(w_minus, i, dt, b);
        store_dt(w_rest, i, dt, c);
    }
}

/* ------------------------------------------------------------------ K3+K4 (P:446-471, :551-559, :370)
 * modes: 0 sum|w||g|  1 sum w^2 g^2  2 sum|g|  3 sum|w|  4 sum w^2.
 * Element terms are fp32 products as torch computes them (W.float(), g.float());
 * the running sum is double (the reference's fp32 .sum() is matched to 1e-5 rel). */
double oracle_absprod_reduce(const void* w, const void* g, int64_t n, int dtw, int dtg, int mode) {
    double acc = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        float a = (mode == 2) ? 0.f : load_dt(w, i, dtw);
        float b = (mode >= 3) ? 0.f : load_dt(g, i, dtg);
        float t;
        switch (mode) {
            case 0: t = fabsf(a) * fabsf(b); break;
            case 1: t = (a * a) * (b * b); break;
            case 2: t = fabsf(b); break;
            case 3: t = fabsf(a); break;
            default: t = a * a; break;
        }
        acc += (double)t;
    }
    return acc;
}

/* ------------------------------------------------------------------ K6 (W:71-84)
 * scaler_row *= n/(n+b); n += b; scaler_row += norm(x,2,dim=tokens)**2 / n
 *
 * The sum of squares is torch's, bit for bit, as its CPU kernel forms it for this call
 * (`torch.norm(inp, p=2, dim=1)` on the TRANSPOSED [cols, tokens] fp32 view: an outer reduction,
 * one fp32 accumulator per column, tokens in order, acc = fma(x, x, acc) — ATen's NormTwoOps
 * `acc + data * data` compiled with contraction; measured in this container against torch
 * 2.10 for fp32 / fp16 / bf16 inputs at 16 ... 2056 tokens x 768 ... 6144 columns: 0 differing
 * columns, where a double-precision sum rounded once differs in most columns at long token
 * counts).  At true row lengths the Wanda selection sits on near-ties of |W| * sqrt(scaler_row)
 * (bf16 weights take few distinct magnitudes), so an ulp in this statistic moves masks:
 * tests/test_true_width.py.  A GPU reduction necessarily adds in another order; the HIP
 * kernel is held to a few ulps of this value (tests/test_gpu_parity.py). */
static inline float colsq_sum(const void* x, int64_t tokens, int64_t cols, int64_t c, int dt) {
    float s = 0.0f;
    for (int64_t t = 0; t < tokens; ++t) {
        float v = load_dt(x, t * cols + c, dt);
        s = fmaf(v, v, s);
    }
    return s;
}

void oracle_colsqnorm_accum(float* scaler_row, const void* x, int64_t tokens, int64_t cols,
                            int dt, int64_t n_before, int64_t batch) {
    float decay = (float)((double)n_before / (double)(n_before + batch));
    float n_new = (float)(n_before + batch);
    for (int64_t c = 0; c < cols; ++c) {
        float nrm = sqrtf(colsq_sum(x, tokens, cols, c, dt));
        float sq = nrm * nrm;
        float r = scaler_row[c] * decay;
        scaler_row[c] = r + sq / n_new;
    }
}

/* One input's own statistic, torch.norm(x, p=2, dim=tokens) ** 2 (W:84), and the running mean
 * of W:80-84 replayed over such per-batch rows in order — the decomposition the data-parallel
 * stage 2 exchanges; composing the two equals oracle_colsqnorm_accum call by call. */
void oracle_colsq_raw(float* out_row, const void* x, int64_t tokens, int64_t cols, int dt) {
    for (int64_t c = 0; c < cols; ++c) {
        float nrm = sqrtf(colsq_sum(x, tokens, cols, c, dt));
        out_row[c] = nrm * nrm;
    }
}
void oracle_colsq_replay(float* scaler_row, const float* sq, const int64_t* batches, int64_t J,
                         int64_t cols, int64_t ld, int64_t n_before) {
    for (int64_t c = 0; c < cols; ++c) {
        float row = scaler_row[c];
        int64_t n0 = n_before;
        for (int64_t j = 0; j < J; ++j) {
            float decay = (float)((double)n0 / (double)(n0 + batches[j]));
            float n_new = (float)(n0 + batches[j]);
            float r = row * decay;
            row = r + sq[j * ld + c] / n_new;
            n0 += batches[j];
        }
        scaler_row[c] = row;
    }
}

/* ------------------------------------------------------------------ K7 (W:260-279, W:541-558) */
static inline float wanda_metric(const void* w, int64_t i, int dt, float sq) {
    return fabsf(load_dt(w, i, dt)) * sq;   /* abs(W) * sqrt(scaler_row) in fp32 */
}
/* torch.sort(stable=True) ascending: NaN last, ties (incl. -0.0 == +0.0) by original index.
 * An order-preserving 32-bit key of the metric and a STABLE least-significant-digit radix sort
 * give exactly that order (the comparator-based qsort this replaces took most of a CPU parity
 * run at true row lengths); `cmp_f` below is the same order as a comparator. */
static inline uint32_t sort_key(float m) {
    uint32_t u;
    if (isnan(m)) return 0xffffffffu;
    if (m == 0.0f) m = 0.0f;                       /* -0.0 and +0.0 compare equal */
    memcpy(&u, &m, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
/* stable LSD radix sort of (key, idx) pairs by key; result in key/idx (tmp arrays: same length) */
static void radix_sort_pairs(uint32_t* key, int32_t* idx, uint32_t* tkey, int32_t* tidx, int64_t n) {
    for (int pass = 0; pass < 4; ++pass) {
        int64_t count[257] = {0};
        const int sh = 8 * pass;
        for (int64_t i = 0; i < n; ++i) ++count[((key[i] >> sh) & 0xff) + 1];
        for (int b = 0; b < 256; ++b) count[b + 1] += count[b];
        for (int64_t i = 0; i < n; ++i) {
            int64_t d = count[(key[i] >> sh) & 0xff]++;
            tkey[d] = key[i];
            if (idx) tidx[d] = idx[i];
        }
        uint32_t* sk = key; key = tkey; tkey = sk;
        if (idx) { int32_t* si = idx; idx = tidx; tidx = si; }
    }   /* four passes: the result is back in the caller's arrays */
}
static int cmp_f(const void* a, const void* b) {
    float x = *(const float*)a, y = *(const float*)b;
    int xn = isnan(x), yn = isnan(y);
    if (xn != yn) return xn - yn;
    return (x > y) - (x < y);
}
/* rows mode: zero the first k columns of each row in stable ascending metric order */
void oracle_wanda_prune_rows(void* w, const float* scaler_row, int64_t rows, int64_t cols,
                             int dt, int64_t k, uint8_t* mask) {
    uint32_t* key = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)cols * 2);
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)cols * 2);
    float* sq = (float*)malloc(sizeof(float) * (size_t)cols);
    for (int64_t c = 0; c < cols; ++c) sq[c] = sqrtf(scaler_row[c]);
    if (mask) memset(mask, 0, (size_t)(rows * cols));
    for (int64_t r = 0; r < rows; ++r) {
        for (int64_t c = 0; c < cols; ++c) {
            key[c] = sort_key(wanda_metric(w, r * cols + c, dt, sq[c]));
            idx[c] = (int32_t)c;
        }
        radix_sort_pairs(key, idx, key + cols, idx + cols, cols);
        for (int64_t j = 0; j < k && j < cols; ++j) {
            store_dt(w, r * cols + idx[j], dt, 0.0f);
            if (mask) mask[r * cols + idx[j]] = 1;
        }
    }
    free(key); free(idx); free(sq);
}
/* matrix mode: thres = sort(flatten)[k]; zero every metric <= thres */
void oracle_wanda_prune_matrix(void* w, const float* scaler_row, int64_t rows, int64_t cols,
                               int dt, int64_t k, uint8_t* mask) {
    int64_t n = rows * cols;
    float* m = (float*)malloc(sizeof(float) * (size_t)n);
    uint32_t* s = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n * 2);
    float* sq = (float*)malloc(sizeof(float) * (size_t)cols);
    for (int64_t c = 0; c < cols; ++c) sq[c] = sqrtf(scaler_row[c]);
    for (int64_t i = 0; i < n; ++i) {
        m[i] = wanda_metric(w, i, dt, sq[i % cols]);
        s[i] = sort_key(m[i]);
    }
    radix_sort_pairs(s, NULL, s + n, NULL, n);
    /* the k-th smallest VALUE (any element carrying that key: equal keys are equal values, or
     * both NaN — then `<=` is false for every element, as in torch) */
    float thres = 0.0f;
    int found = 0;
    for (int64_t i = 0; i < n && !found; ++i)
        if (sort_key(m[i]) == s[k]) { thres = m[i]; found = 1; }
    for (int64_t i = 0; i < n; ++i) {
        int z = (m[i] <= thres);
        if (z) store_dt(w, i, dt, 0.0f);
        if (mask) mask[i] = (uint8_t)z;
    }
    free(m); free(s); free(sq);
}

/* ------------------------------------------------------------------ K7, structured n:m branch
 * LAVIS/lavis/compression/pruners/wanda_pruner.py:265-270 (= :546-551): for every group of m
 * consecutive columns, `torch.topk(W_metric[:, ii:ii+m].float(), n, dim=1, largest=False)` picks
 * the n smallest metrics of each row's group (a NaN counts as the largest), those weights are
 * zeroed.  Equal metrics: the lower column first (torch leaves the order among ties to the
 * implementation; the parity fixtures hold no ties inside a group, the tie rule is this build's).
 * A ragged last group (cols % m != 0) is selected among its own elements, as the slice is; the
 * caller rejects one shorter than n (topk raises there). */
void oracle_wanda_prune_nm(void* w, const float* scaler_row, int64_t rows, int64_t cols, int dt,
                           int n, int m, uint8_t* mask) {
    float* sq = (float*)malloc(sizeof(float) * (size_t)cols);
    for (int64_t c = 0; c < cols; ++c) sq[c] = sqrtf(scaler_row[c]);
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t c0 = 0; c0 < cols; c0 += m) {
            int len = (int)((cols - c0) < m ? (cols - c0) : m);
            float met[64];
            for (int j = 0; j < len; ++j) met[j] = wanda_metric(w, r * cols + c0 + j, dt, sq[c0 + j]);
            for (int j = 0; j < len; ++j) {
                int rank = 0;
                for (int i = 0; i < len; ++i) {
                    if (i == j) continue;
                    int less = cmp_f(&met[i], &met[j]);
                    if (less < 0 || (less == 0 && i < j)) ++rank;
                }
                int z = rank < n;
                if (z) store_dt(w, r * cols + c0 + j, dt, 0.0f);
                if (mask) mask[r * cols + c0 + j] = (uint8_t)z;
            }
        }
    free(sq);
}

/* ------------------------------------------------------------------ K8 (UPop/ecoflap_compression_vqa.py:124-129) */
void oracle_mask_mul(void* g, const uint8_t* keep, int64_t n, int dt) {
    for (int64_t i = 0; i < n; ++i)
        store_dt(g, i, dt, load_dt(g, i, dt) * (keep[i] ? 1.0f : 0.0f));
}

/* ------------------------------------------------------------------ SparseGPT block step
 * LAVIS/lavis/compression/pruners/sparsegpt_pruner.py:172-216 for columns [i1, i1+count) of the
 * fp32 working copy W[rows, ldw]; Hinv[cols, ldh] is the upper Cholesky factor (:162).
 * tmp = W1**2 / diag(Hinv1)**2; thresh = sorted(tmp.flatten())[k]; mask1 = tmp <= thresh;
 * sequential column sweep with the reference's op order (product, then subtract).  The
 * trailing GEMM (:216) stays with the caller. */
void oracle_sparsegpt_block(float* W, int64_t rows, int64_t ldw, const float* Hinv, int64_t ldh,
                            int64_t i1, int count, int64_t k, float* err_out, uint8_t* mask_out) {
    int64_t n = rows * count;
    float* tmp = (float*)malloc(sizeof(float) * (size_t)n);
    float* srt = (float*)malloc(sizeof(float) * (size_t)n);
    for (int64_t r = 0; r < rows; ++r)
        for (int c = 0; c < count; ++c) {
            float w = W[r * ldw + i1 + c], d = Hinv[(i1 + c) * ldh + i1 + c];
            float a = w * w, b = d * d;
            tmp[r * count + c] = a / b;
        }
    memcpy(srt, tmp, sizeof(float) * (size_t)n);
    qsort(srt, (size_t)n, sizeof(float), cmp_f);
    float thresh = srt[k];
    float* w1 = (float*)malloc(sizeof(float) * (size_t)count);
    for (int64_t r = 0; r < rows; ++r) {
        for (int c = 0; c < count; ++c) w1[c] = W[r * ldw + i1 + c];
        for (int i = 0; i < count; ++i) {
            int masked = tmp[r * count + i] <= thresh;
            float wi = w1[i], d = Hinv[(i1 + i) * ldh + i1 + i];
            float q = masked ? 0.0f : wi;
            float err = (wi - q) / d;
            for (int j = i; j < count; ++j) {
                float p = err * Hinv[(i1 + i) * ldh + i1 + j];
                w1[j] = w1[j] - p;
            }
            W[r * ldw + i1 + i] = q;
            err_out[r * count + i] = err;
            if (mask_out) mask_out[r * count + i] = (uint8_t)masked;
        }
    }
    free(tmp); free(srt); free(w1);
}

/* The same block step under n:m (sparsegpt_pruner.py:190, :196-198): no threshold; at every block
 * column i with i % m == 0 the n smallest `W1[:, i:i+m]**2 / diag(Hinv1)[i:i+m]**2` of each row —
 * on the CURRENT values of the sweep — join the mask (`torch.topk(..., largest=False)`: NaN counts
 * as the largest; equal values: the lower column first, this build's rule).  A group cut short by
 * the block's end is selected among its own columns; the caller rejects one shorter than n. */
void oracle_sparsegpt_block_nm(float* W, int64_t rows, int64_t ldw, const float* Hinv, int64_t ldh,
                               int64_t i1, int count, int n, int m, float* err_out, uint8_t* mask_out) {
    float* w1 = (float*)malloc(sizeof(float) * (size_t)count);
    uint8_t* mk = (uint8_t*)malloc((size_t)count);
    for (int64_t r = 0; r < rows; ++r) {
        for (int c = 0; c < count; ++c) { w1[c] = W[r * ldw + i1 + c]; mk[c] = 0; }
        for (int i = 0; i < count; ++i) {
            if (i % m == 0) {
                int len = count - i < m ? count - i : m;
                float t[64];
                for (int p = 0; p < len; ++p) {
                    float w = w1[i + p], d = Hinv[(i1 + i + p) * ldh + i1 + i + p];
                    float a = w * w, b = d * d;
                    t[p] = a / b;
                }
                for (int p = 0; p < len; ++p) {
                    int rank = 0;
                    for (int q = 0; q < len; ++q) {
                        if (q == p) continue;
                        int less = cmp_f(&t[q], &t[p]);
                        if (less < 0 || (less == 0 && q < p)) ++rank;
                    }
                    if (rank < n) mk[i + p] = 1;
                }
            }
            int masked = mk[i];
            float wi = w1[i], d = Hinv[(i1 + i) * ldh + i1 + i];
            float q = masked ? 0.0f : wi;
            float err = (wi - q) / d;
            for (int j = i; j < count; ++j) {
                float p = err * Hinv[(i1 + i) * ldh + i1 + j];
                w1[j] = w1[j] - p;
            }
            W[r * ldw + i1 + i] = q;
            err_out[r * count + i] = err;
            if (mask_out) mask_out[r * count + i] = (uint8_t)masked;
        }
    }
    free(w1); free(mk);
}
