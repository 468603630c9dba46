// allocator.cpp — K5: per-group sparsity allocation (host code of the C ABI, no GPU).
//
// Replaces LayerSparsity.compute_the_sparsity_per_group
//   (LAVIS/lavis/compression/pruners/layer_single_base_pruner.py:247-314).
// The reference runs this on torch CPU tensors whose dtype drifts from int64 to
// float32 after the first `keep + add` (:262); counts above 2^24 (a ViT-g block
// has 25 231 360 parameters) are therefore rounded, and every `sum()` is torch's
// 8-lane cascade.  To give bit-identical keep-counts this file replays that
// arithmetic in scalar float32: same promotions, same summation order
// (torch_sum_f32 below), same stable descending argsort, and the shipped
// "remove excess" branch that adds instead of subtracting (:301).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <vector>

#include "../../include/ecoflap_hip.h"

namespace {

int ceil_log2(int64_t x) {
    if (x <= 2) return 1;
    int n = 0;
    uint64_t v = (uint64_t)(x - 1);
    while (v) { ++n; v >>= 1; }
    return n;
}

// ATen SumKernel.cpp multi_row_sum with LANES-wide rows: `size` rows, ILP rows interleaved.
template <int LANES>
struct VecAcc {
    float v[LANES];
    VecAcc() { for (int i = 0; i < LANES; ++i) v[i] = 0.f; }
    void add(const float* p) { for (int i = 0; i < LANES; ++i) v[i] = v[i] + p[i]; }
    void add(const VecAcc& o) { for (int i = 0; i < LANES; ++i) v[i] = v[i] + o.v[i]; }
    void zero() { for (int i = 0; i < LANES; ++i) v[i] = 0.f; }
};

template <int LANES>
VecAcc<LANES> row_sum(const float* data, int64_t size /* in LANES-wide vectors */) {
    constexpr int ILP = 4, LEVELS = 4;
    const int64_t size_ilp = size / ILP;
    const int power = std::max(4, ceil_log2(size_ilp) / LEVELS);
    const int64_t step = (int64_t)1 << power, mask0 = step - 1;
    VecAcc<LANES> acc[LEVELS][ILP];
    int64_t i = 0;
    for (; i + step <= size_ilp;) {
        for (int64_t j = 0; j < step; ++j, ++i)
            for (int k = 0; k < ILP; ++k) acc[0][k].add(data + (i * ILP + k) * LANES);
        for (int j = 1; j < LEVELS; ++j) {
            for (int k = 0; k < ILP; ++k) { acc[j][k].add(acc[j - 1][k]); acc[j - 1][k].zero(); }
            if ((i & (mask0 << (j * power))) != 0) break;
        }
    }
    for (; i < size_ilp; ++i)
        for (int k = 0; k < ILP; ++k) acc[0][k].add(data + (i * ILP + k) * LANES);
    for (int j = 1; j < LEVELS; ++j)
        for (int k = 0; k < ILP; ++k) acc[0][k].add(acc[j][k]);
    for (int64_t r = size_ilp * ILP; r < size; ++r) acc[0][0].add(data + r * LANES);
    for (int k = 1; k < ILP; ++k) acc[0][0].add(acc[0][k]);
    return acc[0][0];
}

// torch.sum of a contiguous float32 CPU tensor (8 lanes under every x86 dispatch level)
float torch_sum_f32(const std::vector<float>& x) {
    const int64_t n = (int64_t)x.size();
    constexpr int L = 8;
    if (n >= L) {
        const int64_t nvec = n / L;
        VecAcc<L> vacc = row_sum<L>(x.data(), nvec);
        float fin = 0.f;
        for (int64_t k = nvec * L; k < n; ++k) fin = fin + x[k];
        for (int k = 0; k < L; ++k) fin = fin + vacc.v[k];
        return fin;
    }
    return row_sum<1>(x.data(), n).v[0];
}

}  // namespace

extern "C" int ecoflap_allocate_sparsity(const float* group_scores,
                                         const int64_t* group_num_params, int n_groups,
                                         int64_t total_parameters_to_keep,
                                         double max_sparsity_per_layer, float* out_sparsity,
                                         double* out_keep) {
    if (n_groups < 0) return ECOFLAP_ESIZE;
    if (n_groups == 0) return 0;
    if (!group_scores || !group_num_params || !out_sparsity) return ECOFLAP_ENULL;
    const int G = n_groups;
    const int64_t total = total_parameters_to_keep;
    const float total_f = (float)total;
    const float one_minus = (float)(1.0 - max_sparsity_per_layer);
    std::vector<float> scores(group_scores, group_scores + G), num_f(G), keep(G), add(G);
    std::vector<int64_t> keep_i(G);
    int64_t isum = 0;
    for (int g = 0; g < G; ++g) {
        num_f[g] = (float)group_num_params[g];
        keep_i[g] = (int64_t)(int32_t)std::ceil(num_f[g] * one_minus);   // :253
        isum += keep_i[g];
    }
    bool is_float = false;
    const int64_t guard_max = 1000000;
    int64_t guard = 0;
    auto below_total = [&]() {
        return is_float ? (torch_sum_f32(keep) < total_f) : (isum < total);
    };
    while (below_total()) {                                               // :255
        if (++guard > guard_max) return ECOFLAP_ESIZE;  // the reference would spin forever
        const float total_ratio = torch_sum_f32(scores);                  // :256
        const float rest = is_float ? (total_f - torch_sum_f32(keep)) : (float)(total - isum);
        for (int g = 0; g < G; ++g) {
            const float share = scores[g] / total_ratio;
            add[g] = std::ceil(share * rest);                             // :260
            const float base = is_float ? keep[g] : (float)keep_i[g];
            keep[g] = base + add[g];                                      // :262
        }
        is_float = true;
        for (int g = 0; g < G; ++g) {
            if (keep[g] >= num_f[g]) scores[g] = 0.f;                     // :264
            keep[g] = (keep[g] != keep[g]) ? keep[g] : std::min(keep[g], num_f[g]);   // :266
        }
        if (torch_sum_f32(add) == 0.f) {                                  // :269
            const float cur = torch_sum_f32(keep);
            if (cur < total_f) {
                float need = total_f - cur;
                while (need > 0.f) {
                    if (++guard > guard_max) return ECOFLAP_ESIZE;
                    for (int g = 0; g < G; ++g) {
                        if (!(scores[g] > 0.f)) continue;
                        const float room = num_f[g] - keep[g];
                        const float can = (room < need) ? room : need;
                        keep[g] = keep[g] + can;
                        need = need - can;
                        if (need == 0.f) break;
                    }
                }
            }
        }
        if (torch_sum_f32(keep) > total_f) {                              // :288
            float excess = torch_sum_f32(keep) - total_f;
            while (excess > 0.f) {
                if (++guard > guard_max) return ECOFLAP_ESIZE;
                std::vector<int> order(G);
                std::iota(order.begin(), order.end(), 0);
                std::stable_sort(order.begin(), order.end(),
                                 [&](int a, int b) { return keep[a] > keep[b]; });
                for (int idx : order) {
                    const float floor_f = (float)(int32_t)(num_f[idx] * one_minus);   // :299
                    const float room = keep[idx] - floor_f;
                    const float can = (room < excess) ? room : excess;
                    keep[idx] = keep[idx] + can;                          // adds, as shipped (:301)
                    excess = excess - can;
                    if (excess == 0.f) break;
                }
            }
        }
    }
    for (int g = 0; g < G; ++g) {
        const float kf = is_float ? keep[g] : (float)keep_i[g];
        float sp = 1.0f - kf / num_f[g];                                  // :312
        if (sp == sp) sp = std::min(std::max(sp, 0.0f), 1.0f);
        out_sparsity[g] = sp;
        if (out_keep) out_keep[g] = (double)kf;
    }
    return 0;
}
