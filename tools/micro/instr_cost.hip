// instr_cost.hip — VALU issue cost of the instructions K1's generator and update are made of,
// measured on gfx950 at full occupancy (8 waves/SIMD): 8 independent chains per lane, 4096
// iterations, time relative to v_xor_b32.  hipcc --offload-arch=gfx950 -O3 instr_cost.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define ITERS 2048
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
    uint32_t a[8];
    uint64_t w[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = (threadIdx.x * 2654435761u + seed * (i + 1)) | 0x3f000001u;
        a[i] &= 0x3fffffffu;            // a positive normal float (0.5..2) when read as f32
        w[i] = ((uint64_t)a[i] << 32) | (a[i] ^ 0x55u);
    }
    const uint32_t m0 = 0xD2511F53u, c16 = 16;
    for (int it = 0; it < ITERS; ++it) {
#define X(i)                                                                                        \
    if (OP == 0) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m0));                     \
    if (OP == 1) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(m0), "v"(c16)); \
    if (OP == 2) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w[i]) : "v"((uint32_t)w[i]), "v"(m0) : "vcc"); \
    if (OP == 3) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m0));                  \
    if (OP == 4) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m0));                  \
    if (OP == 5) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(m0));                 \
    if (OP == 6) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));                                   \
    if (OP == 7) asm volatile("v_sin_f32 %0, %0" : "+v"(a[i]));                                   \
    if (OP == 8) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));                                  \
    if (OP == 9) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m0));             \
    if (OP == 10) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(w[i]));                           \
    if (OP == 11) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(w[i]));                           \
    if (OP == 12) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(w[i]));                       \
    if (OP == 13) asm volatile("v_add_f32 %0, %0, %0" : "+v"(a[i]));                              \
    if (OP == 14) asm volatile("v_alignbit_b32 %0, %0, %1, 9" : "+v"(a[i]) : "v"(m0));            \
    if (OP == 15) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]));                              \
    if (OP == 16) asm volatile("v_pk_add_f16 %0, %0, %0" : "+v"(a[i]));                           \
    if (OP == 17) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(a[i]) : "v"(c16)); \
    if (OP == 18) asm volatile("v_dot2_f32_bf16 %0, %0, %1, %0" : "+v"(a[i]) : "v"(m0));          \
    if (OP == 19) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));                          \
    if (OP == 20) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m0));             \
    if (OP == 21) asm volatile("v_cos_f32 %0, %0" : "+v"(a[i]));                                  \
    if (OP == 22) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(m0));            \
    if (OP == 23) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m0), "v"(c16));     \
    if (OP == 24) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(m0));             \
    if (OP == 25) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));                                  \
    if (OP == 26) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));                                  \
    if (OP == 27) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(m0));            \
    if (OP == 28) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"((uint32_t)(w[i] >> 32)), "v"(m0) : "vcc"); \
    if (OP == 29) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m0), "v"(c16));
        REP8(X)
#undef X
    }
    uint32_t r = 0;
    for (int i = 0; i < 8; ++i) r ^= a[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

static double base_ps = 0;
template <int OP> void run(const char* name) {
    uint32_t* out;
    const int blocks = 256 * 8 * 4;      // 32 workgroups per CU: several rounds of full occupancy
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1u);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(s);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 2u + rep);
        hipEventRecord(e);
        hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        if (ms < best) best = ms;
    }
    const double wave_instr = (double)blocks * 4 * ITERS * 8;       // wave-level instructions
    const double per_simd = wave_instr / (256.0 * 4);               // per SIMD
    const double ns_per = best * 1e6 / per_simd;
    if (OP == 0) base_ps = ns_per;
    printf("%-22s %8.3f ms  %6.3f ns per wave-instr per SIMD  = %5.2f x v_xor\n", name, best, ns_per,
           ns_per / base_ps);
    hipFree(out);
}

int main() {
    run<0>("v_xor_b32"); run<1>("v_bitop3_b32"); run<2>("v_mad_u64_u32 (c=0)"); run<28>("v_mad_u64_u32 (acc)");
    run<3>("v_mul_lo_u32"); run<4>("v_mul_hi_u32"); run<5>("v_mul_u32_u24"); run<24>("v_mul_hi_u32_u24");
    run<22>("v_mad_u32_u24");
    run<6>("v_log_f32"); run<7>("v_sin_f32"); run<21>("v_cos_f32"); run<8>("v_sqrt_f32"); run<25>("v_rsq_f32");
    run<26>("v_exp_f32");
    run<9>("v_cvt_pk_bf16_f32"); run<20>("v_cvt_pk_f16_f32"); run<10>("v_pk_mul_f32"); run<11>("v_pk_add_f32");
    run<12>("v_pk_fma_f32"); run<13>("v_add_f32"); run<19>("v_fma_f32"); run<14>("v_alignbit_b32");
    run<15>("v_cvt_f32_u32"); run<16>("v_pk_add_f16"); run<17>("v_lshlrev_b32_sdwa"); run<18>("v_dot2_f32_bf16");
    run<23>("v_perm_b32"); run<27>("v_lshl_add_u32"); run<29>("v_xad_u32");
    return 0;
}
