// k1_bound.hip — what bounds the layer-batched K1 kernel: the product kernel next to a
// VALU-only twin (same generator + update, stores never taken) and a memory-only twin (same
// loads / stores, trivial arithmetic), on BLIP-2's matrix shapes, 16 units, cold HBM.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 k1_bound.hip -o k1_bound
#include "../../ecoflap_amd/csrc/zo_perturb.hip"
#include <stdio.h>
#include <vector>

template <int DT, int MODE>   // 1: VALU only, 2: memory only
__global__ __launch_bounds__(ECO_K1_THREADS) void twin_kernel(void* __restrict__ w, int64_t n, float eps,
                                                   int n_units, const UnitTable tab, int never) {
    constexpr int N = Vec<DT>::N;
    const int64_t nvec = n / N;
    ECO_FOR_SUPER_ROWS(R, nvec) {
        const LaneVecs L = lane_vecs(R, nvec);
        u32x4 s0 = ld16_if(L.ok0, w, L.v0), s1 = ld16_if(L.ok1, w, L.v1);
        for (int u = 0; u < n_units; ++u) {
            u32x4 p0, m0, p1, m1;
            if (MODE == 1) {
                float z[2 * N];
                gen_z_lane<DT>(L.v0, (uint32_t)tab.seed[u], (uint32_t)(tab.seed[u] >> 32), z);
                unit_update<DT, true>(s0, z, eps, p0, m0);
                unit_update<DT, true>(s1, z + N, eps, p1, m1);
            } else {
                const uint32_t k = (uint32_t)tab.seed[u];
                p0 = s0 + k; m0 = s0 ^ k; p1 = s1 + k; m1 = s1 ^ k;
                s0 = s0 + 1u; s1 = s1 + 1u;
            }
            void* dp = tab.plus[u];
            const bool store = (MODE == 2) ? (dp != nullptr) : (never != 0 && p0[0] == 0x12345u);
            if (store) {
                void* dm = tab.minus[u];
                if (L.ok0) { st16_nt(dp, L.v0, p0); st16_nt(dm, L.v0, m0); }
                if (L.ok1) { st16_nt(dp, L.v1, p1); st16_nt(dm, L.v1, m1); }
            }
        }
        if (MODE == 2 || never) {
            if (L.ok0) st16(w, L.v0, s0);
            if (L.ok1) st16(w, L.v1, s1);
        } else if ((s0[0] ^ s1[3]) == 0x7654321u) {
            st16(w, L.v0, s0);
        }
    }
}

struct Shape { const char* name; int64_t n; int dt; };

// ~0.4 ms of back-to-back MFMAs on every SIMD: the state the chip is in when K1 runs inside
// the scoring loop (between GEMM-bound forwards), where DVFS has lowered the clock
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(256) void heater(float* out, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (i + 1)); }
    f32x16 acc0 = {0}, acc1 = {0};
    for (int it = 0; it < iters; ++it) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0);
    }
    if (acc0[0] + acc1[5] == 123.f) out[threadIdx.x] = acc0[1];
}
static int g_heat = 0;
static float* g_heat_out = nullptr;

template <int DT> float run(int mode, void* w, int64_t n, const UnitTable& tab, int U) {
    const unsigned g = grid_exact(n / Vec<DT>::N);
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    if (g_heat) hipLaunchKernelGGL(heater, dim3(256 * 8), dim3(256), 0, 0, g_heat_out, g_heat);
    hipEventRecord(s);
    if (mode == 0) hipLaunchKernelGGL((zo_perturb_units_kernel<DT, false>), dim3(g), dim3(ECO_K1_THREADS), 0, 0, w, n, 1e-3f, U, tab);
    if (mode == 1) hipLaunchKernelGGL((twin_kernel<DT, 1>), dim3(g), dim3(ECO_K1_THREADS), 0, 0, w, n, 1e-3f, U, tab, 0);
    if (mode == 2) hipLaunchKernelGGL((twin_kernel<DT, 2>), dim3(g), dim3(ECO_K1_THREADS), 0, 0, w, n, 1e-3f, U, tab, 0);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    hipEventDestroy(s); hipEventDestroy(e);
    return ms * 1e3f;
}

int main(int argc, char** argv) {
    const int U = argc > 1 ? atoi(argv[1]) : 16;
    g_heat = argc > 2 ? atoi(argv[2]) : 0;          // MFMA iterations of the pre-heater (0 = cool chip)
    hipMalloc(&g_heat_out, 4096);
    printf("units %d, heater iterations %d\n", U, g_heat);
    const Shape shapes[] = {{"t5_wi_wo bf16", 5120 * 2048, ECOFLAP_BF16}, {"t5_qkvo bf16", 2048 * 2048, ECOFLAP_BF16},
                            {"vit_fc f16", 6144 * 1408, ECOFLAP_F16}, {"vit_qkv f16", 4224 * 1408, ECOFLAP_F16},
                            {"vit_proj f16", 1408 * 1408, ECOFLAP_F16},
                            {"6144x1408 as bf16", 6144 * 1408, ECOFLAP_BF16}, {"5120x2048 as f16", 5120 * 2048, ECOFLAP_F16}};
    const char* modes[] = {"product", "VALU only", "memory only"};
    for (const Shape& sh : shapes) {
        const size_t bytes = (size_t)sh.n * 2;
        const int sets = (int)(1300000000ull / ((2 * U + 1) * bytes)) + 2;   // > 1 GiB: cold HBM
        std::vector<char*> bufs(sets);
        for (int k = 0; k < sets; ++k) {
            hipMalloc(&bufs[k], (2 * U + 1) * bytes);
            hipMemset(bufs[k], 0x3c, (2 * U + 1) * bytes);      // 0x3c3c: a normal f16 / bf16 value
        }
        hipDeviceSynchronize();
        for (int mode = 0; mode < 3; ++mode) {
            double tot = 0; int cnt = 0;
            for (int it = 0; it < 3 * sets; ++it) {
                char* b = bufs[it % sets];
                UnitTable tab;
                for (int u = 0; u < ECOFLAP_MAX_UNITS; ++u) { tab.seed[u] = 0; tab.plus[u] = tab.minus[u] = nullptr; tab.z[u] = nullptr; }
                for (int u = 0; u < U; ++u) {
                    tab.seed[u] = 1000003ull * it + u;
                    tab.plus[u] = b + (size_t)(1 + 2 * u) * bytes;
                    tab.minus[u] = b + (size_t)(2 + 2 * u) * bytes;
                }
                float us = sh.dt == ECOFLAP_BF16 ? run<ECOFLAP_BF16>(mode, b, sh.n, tab, U)
                                                 : run<ECOFLAP_F16>(mode, b, sh.n, tab, U);
                if (it >= sets) { tot += us; ++cnt; }
            }
            const double us = tot / cnt, gb = (2.0 * U + 2) * bytes / 1e9;
            printf("%-14s %-12s %8.2f us   (%6.0f GB/s algorithmic, %4.1f %% of 8 TB/s)\n", sh.name, modes[mode], us,
                   gb / us * 1e6, gb / us * 1e6 / 80);
        }
        for (char* b : bufs) hipFree(b);
    }
    return 0;
}
