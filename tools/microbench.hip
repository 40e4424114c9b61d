// microbench.hip -- measured VALU issue cost on gfx950 of the integer instructions the NTT is
// built from.  Every instruction is emitted through inline asm (the compiler cannot fold the
// loop), 16 independent chains per lane, 8 waves per SIMD.  Reports cycles per wave-instruction
// per SIMD (at the clock measured with s_memtime-free wall time / assumed 2.4 GHz).
// Build: hipcc --offload-arch=gfx950 -O3 microbench.hip -o microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITERS 2048
#define UNROLL 16

#define ASM1(str) asm volatile(str : "+v"(a[i]) : "v"(b), "v"(c))

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed)
{
    uint32_t a[UNROLL], a2[UNROLL], b = seed + threadIdx.x, c = seed * 3 + 1;
    double d[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; i++) { a[i] = seed + i + threadIdx.x; a2[i] = a[i] * 7u; d[i] = 1.0 + i; }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            if (OP == 0) ASM1("v_add_u32 %0, %0, %1");
            else if (OP == 1) ASM1("v_mul_lo_u32 %0, %0, %1");
            else if (OP == 2) ASM1("v_mul_hi_u32 %0, %0, %1");
            else if (OP == 3) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*(uint64_t *)&d[i]) : "v"(b), "v"(c) : "vcc");
            else if (OP == 4) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(a[i]), "+v"(a2[i]) : "v"(b), "v"(c) : "vcc");
            else if (OP == 5) ASM1("v_alignbit_b32 %0, %0, %1, 7");
            else if (OP == 6) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
            else if (OP == 7) ASM1("v_lshlrev_b32 %0, 5, %0");
            else if (OP == 8) ASM1("v_lshl_or_b32 %0, %0, 3, %1");
            else if (OP == 9) ASM1("v_mul_u32_u24 %0, %0, %1");
            else if (OP == 10) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d[i]));
            else if (OP == 11) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(*(uint64_t *)&d[i]));
            else if (OP == 12) ASM1("v_add3_u32 %0, %0, %1, %2");
            else if (OP == 13) ASM1("v_and_or_b32 %0, %0, %1, %2");
            else if (OP == 14) ASM1("v_xor_b32 %0, %0, %1");
            else if (OP == 15) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : );
            else if (OP == 16) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(*(uint64_t *)&d[i]), "v"(*(uint64_t *)&d[(i + 1) % UNROLL]) : "vcc");
            else if (OP == 17) asm volatile("v_sub_co_u32 %0, vcc, %0, %2\n\tv_subb_co_u32 %1, vcc, %1, %3, vcc" : "+v"(a[i]), "+v"(a2[i]) : "v"(b), "v"(c) : "vcc");
            else if (OP == 18) ASM1("v_lshrrev_b32 %0, 5, %0");
            else if (OP == 19) ASM1("v_perm_b32 %0, %0, %1, %2");
            else if (OP == 20) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(*(uint64_t *)&d[i]) : "v"(*(uint64_t *)&d[(i + 1) % UNROLL]));
            else if (OP == 21) ASM1("v_mad_u32_u24 %0, %0, %1, %2");
            else if (OP == 22) ASM1("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
            else if (OP == 23) ASM1("v_add_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
            else if (OP == 24) ASM1("v_bfe_u32 %0, %0, 3, 10");
            else if (OP == 25) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
            else if (OP == 26) ASM1("v_sub_u32 %0, %0, %1");
            else if (OP == 27) ASM1("v_mul_hi_u32_u24 %0, %0, %1");
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; i++) r ^= a[i] ^ a2[i] ^ (uint32_t)__double_as_longlong(d[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

static double g_ghz = 2.4;

template <int OP>
void run(const char *name, int instr_per_op, uint32_t *d_out, int blocks)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    double wave_ops = (double)blocks * 4 * ITERS * UNROLL * instr_per_op;
    double per_simd_per_s = wave_ops / (ms * 1e-3) / 1024.0;
    printf("%-34s %8.3f ms   %6.2f cycles per wave-instruction per SIMD (@%.2f GHz)\n", name, ms,
           g_ghz * 1e9 / per_simd_per_s, g_ghz);
}

int main()
{
    int blocks = 256 * 8;   // 8 blocks of 4 waves per CU = 8 waves per SIMD
    uint32_t *d_out;
    hipMalloc(&d_out, (size_t)blocks * 256 * 4);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("device: %s, %d CUs, clockRate %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    run<0>("v_add_u32", 1, d_out, blocks);
    run<26>("v_sub_u32", 1, d_out, blocks);
    run<14>("v_xor_b32", 1, d_out, blocks);
    run<7>("v_lshlrev_b32", 1, d_out, blocks);
    run<18>("v_lshrrev_b32", 1, d_out, blocks);
    run<24>("v_bfe_u32", 1, d_out, blocks);
    run<8>("v_lshl_or_b32", 1, d_out, blocks);
    run<12>("v_add3_u32", 1, d_out, blocks);
    run<13>("v_and_or_b32", 1, d_out, blocks);
    run<5>("v_alignbit_b32", 1, d_out, blocks);
    run<19>("v_perm_b32", 1, d_out, blocks);
    run<15>("v_cndmask_b32 (vcc)", 1, d_out, blocks);
    run<25>("v_cmp_lt_u32", 1, d_out, blocks);
    run<6>("v_cmp_lt_u32 + v_cndmask", 2, d_out, blocks);
    run<16>("v_cmp_lt_u64", 1, d_out, blocks);
    run<4>("v_add_co + v_addc_co", 2, d_out, blocks);
    run<17>("v_sub_co + v_subb_co", 2, d_out, blocks);
    run<20>("v_lshl_add_u64", 1, d_out, blocks);
    run<11>("v_lshlrev_b64", 1, d_out, blocks);
    run<1>("v_mul_lo_u32", 1, d_out, blocks);
    run<2>("v_mul_hi_u32", 1, d_out, blocks);
    run<3>("v_mad_u64_u32", 1, d_out, blocks);
    run<9>("v_mul_u32_u24", 1, d_out, blocks);
    run<27>("v_mul_hi_u32_u24", 1, d_out, blocks);
    run<21>("v_mad_u32_u24", 1, d_out, blocks);
    run<10>("v_fma_f64", 1, d_out, blocks);
    run<22>("v_mov_b32_dpp quad_perm", 1, d_out, blocks);
    run<23>("v_add_u32_dpp quad_perm", 1, d_out, blocks);
    hipFree(d_out);
    return 0;
}
