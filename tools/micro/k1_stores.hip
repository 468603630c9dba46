// k1_stores.hip — store-pattern study for the layer-batched K1 kernel (memory-only twins):
// which mapping of waves to 1-KiB rows, which store flavour and how many rows per lane get
// closest to the HBM write ceiling when one pass writes 2*U+1 streams.
#include "../../ecoflap_amd/csrc/zo_perturb.hip"
#include <stdio.h>
#include <vector>

// VPL rows of 64 vectors per lane; XCD: remap workgroups so each XCD walks a contiguous region
template <int VPL, bool NT, bool XCD, int WG, int LAYOUT = 0>
__global__ __launch_bounds__(WG) void store_twin(void* __restrict__ w, int64_t nvec, int n_units,
                                                 const UnitTable tab) {
    constexpr int WAVES = WG / 64;
    int64_t b = blockIdx.x;
    if (XCD) {
        const int64_t nb8 = gridDim.x / 8;
        b = (b % 8) * nb8 + b / 8;
    }
    const int64_t R = b * WAVES + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    u32x4 s[VPL];
    int64_t v[VPL];
    bool ok[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        v[i] = R * (64 * VPL) + i * 64 + lane;
        ok[i] = v[i] < nvec;
        s[i] = ld16_if(ok[i], w, v[i]);
    }
    for (int u = 0; u < n_units; ++u) {
        const uint32_t k = (uint32_t)tab.seed[u];
        void* dp = tab.plus[u];
        void* dm = tab.minus[u];
        if (LAYOUT == 1) {
            // all 2U streams of a super-row contiguous: [R][2u + sign][VPL*64 vectors]
            char* base = (char*)tab.plus[0];
            dp = base + ((size_t)R * 2 * n_units + 2 * u) * (VPL * 1024) - (size_t)R * (VPL * 1024);
            dm = (char*)dp + VPL * 1024;
        }
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const u32x4 p = s[i] + k, m = s[i] ^ k;
            s[i] = s[i] + 1u;
            if (ok[i]) {
                if (NT) { st16_nt(dp, v[i], p); st16_nt(dm, v[i], m); }
                else { st16(dp, v[i], p); st16(dm, v[i], m); }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i)
        if (ok[i]) st16(w, v[i], s[i]);
}

template <int VPL, bool NT, bool XCD, int WG, int LAYOUT = 0>
float run(void* w, int64_t n, const UnitTable& tab, int U) {
    const int64_t nvec = n / 8;
    const int64_t rows = (nvec + 64 * VPL - 1) / (64 * VPL);
    int64_t blocks = (rows + WG / 64 - 1) / (WG / 64);
    if (XCD) blocks = (blocks + 7) / 8 * 8;
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    hipEventRecord(s);
    hipLaunchKernelGGL((store_twin<VPL, NT, XCD, WG, LAYOUT>), dim3((unsigned)blocks), dim3(WG), 0, 0, w, nvec, U, tab);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    hipEventDestroy(s); hipEventDestroy(e);
    return ms * 1e3f;
}

typedef float (*runner)(void*, int64_t, const UnitTable&, int);
struct Variant { const char* name; runner fn; };

int main(int argc, char** argv) {
    const int U = argc > 1 ? atoi(argv[1]) : 16;
    const size_t pad = argc > 2 ? (size_t)atoll(argv[2]) : 0;      // bytes added to every stream's stride
    printf("units %d, stream stride padding %zu B\n", U, pad);
    const Variant variants[] = {
        {"vpl2 nt wg256 xcd        ", run<2, true, true, 256>},
        {"vpl2 nt wg64  xcd        ", run<2, true, true, 64>},
        {"vpl2 nt wg256     chunked", run<2, true, false, 256, 1>},
        {"vpl2 nt wg256 xcd chunked", run<2, true, true, 256, 1>},
        {"vpl2 nt wg64      chunked", run<2, true, false, 64, 1>},
        {"vpl2 nt wg64  xcd chunked", run<2, true, true, 64, 1>},
        {"vpl2 pl wg64  xcd chunked", run<2, false, true, 64, 1>},
        {"vpl1 nt wg64  xcd chunked", run<1, true, true, 64, 1>},
    };
    struct Shape { const char* name; int64_t n; };
    const Shape shapes[] = {{"5120x2048", 5120 * 2048}, {"2048x2048", 2048 * 2048}, {"6144x1408", 6144 * 1408},
                            {"1408x1408", 1408 * 1408}};
    for (const Shape& sh : shapes) {
        const size_t bytes = (size_t)sh.n * 2;
        const int sets = (int)(1300000000ull / ((2 * U + 1) * bytes)) + 2;
        std::vector<char*> bufs(sets);
        for (int k = 0; k < sets; ++k) {
            hipMalloc(&bufs[k], (2 * U + 1) * (bytes + pad));
            hipMemset(bufs[k], 0x3c, (2 * U + 1) * (bytes + pad));
        }
        hipDeviceSynchronize();
        for (const Variant& var : variants) {
            double tot = 0; int cnt = 0;
            for (int it = 0; it < 3 * sets; ++it) {
                char* b = bufs[it % sets];
                UnitTable tab;
                for (int u = 0; u < ECOFLAP_MAX_UNITS; ++u) { tab.seed[u] = 0; tab.plus[u] = tab.minus[u] = nullptr; tab.z[u] = nullptr; }
                for (int u = 0; u < U; ++u) {
                    tab.seed[u] = 1000003ull * it + u;
                    tab.plus[u] = b + (size_t)(1 + 2 * u) * (bytes + pad);
                    tab.minus[u] = b + (size_t)(2 + 2 * u) * (bytes + pad);
                }
                const float us = var.fn(b, sh.n, tab, U);
                if (it >= sets) { tot += us; ++cnt; }
            }
            const double us = tot / cnt, gb = (2.0 * U + 2) * bytes / 1e9;
            printf("%-10s %s %8.2f us  %6.0f GB/s  %4.1f %%\n", sh.name, var.name, us, gb / us * 1e6, gb / us * 1e6 / 80);
        }
        for (char* b : bufs) hipFree(b);
    }
    return 0;
}
