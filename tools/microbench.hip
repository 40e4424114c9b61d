// microbench.hip -- measured VALU issue rates on gfx950 for the integer ops the NTT is built from.
// Each kernel runs a dependent-free unrolled stream of one instruction kind; we report
// wave-instructions per cycle per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 microbench.hip -o microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITERS 4096
#define UNROLL 16

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed)
{
    uint32_t a[UNROLL], b = seed + threadIdx.x, c = seed * 3 + 1;
    uint64_t w[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; i++) { a[i] = seed + i + threadIdx.x; w[i] = ((uint64_t)a[i] << 32) | (a[i] * 7u); }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            if (OP == 0) a[i] = a[i] + b;                                   // v_add_u32
            else if (OP == 1) a[i] = a[i] * b;                              // v_mul_lo_u32
            else if (OP == 2) a[i] = __umulhi(a[i], b);                     // v_mul_hi_u32
            else if (OP == 3) w[i] = (uint64_t)(uint32_t)w[i] * b + w[i];   // v_mad_u64_u32
            else if (OP == 4) w[i] = w[i] + (((uint64_t)b << 32) | c);      // 64-bit add (2 ops)
            else if (OP == 5) a[i] = __builtin_amdgcn_alignbit(a[i], b, 7); // v_alignbit_b32
            else if (OP == 6) a[i] = (a[i] < b) ? c : a[i];                 // cmp + cndmask
            else if (OP == 7) w[i] = w[i] * (((uint64_t)b << 32) | c);      // 64x64 -> 64 low mul
            else if (OP == 8) w[i] = __umul64hi(w[i], (((uint64_t)b << 32) | c)); // 64x64 high
            else if (OP == 9) a[i] = __umul24(a[i], b);     // v_mul_u32_u24
            else if (OP == 10) { double d = __longlong_as_double(w[i]); d = __builtin_fma(d, 1.0000001, 0.5); w[i] = __double_as_longlong(d); } // v_fma_f64
            else if (OP == 11) w[i] = w[i] << (b & 31);                     // v_lshlrev_b64
            else if (OP == 12) a[i] = a[i] + b + c;                         // v_add3_u32
            else if (OP == 13) a[i] = (a[i] & b) | c;                       // v_and_or_b32
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; i++) r ^= a[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
void run(const char *name, uint32_t *d_out, int blocks)
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
    // waves = blocks * 4; ops per wave = ITERS * UNROLL
    double wave_ops = (double)blocks * 4 * ITERS * UNROLL;
    double per_s = wave_ops / (ms * 1e-3);
    // 1024 SIMDs: ops per SIMD per second; at f GHz cycles/op = f / rate
    printf("%-28s %8.3f ms  %8.2f G wave-ops/s  => %.2f cycles/wave-op/SIMD @2.4GHz (occupancy 8 waves/SIMD)\n", name, ms,
           per_s / 1e9, 2.4e9 / (per_s / 1024.0));
}

int main()
{
    int blocks = 256 * 8;   // 8 blocks of 4 waves per CU = 8 waves per SIMD
    uint32_t *d_out;
    hipMalloc(&d_out, (size_t)blocks * 256 * 4);
    run<0>("v_add_u32", d_out, blocks);
    run<1>("v_mul_lo_u32", d_out, blocks);
    run<2>("v_mul_hi_u32", d_out, blocks);
    run<3>("v_mad_u64_u32", d_out, blocks);
    run<4>("add_u64 (2 instr)", d_out, blocks);
    run<5>("v_alignbit_b32", d_out, blocks);
    run<6>("cmp+cndmask (2 instr)", d_out, blocks);
    run<7>("mul64 lo (compiler seq)", d_out, blocks);
    run<8>("umul64hi (compiler seq)", d_out, blocks);
    run<9>("v_mul_u32_u24", d_out, blocks);
    run<10>("v_fma_f64", d_out, blocks);
    run<11>("v_lshlrev_b64", d_out, blocks);
    run<12>("v_add3_u32", d_out, blocks);
    run<13>("v_and_or_b32", d_out, blocks);
    hipFree(d_out);
    return 0;
}
