// microbench_l4.hip -- (1) whole transforms of the blind-rotation loop in the two arithmetic forms
// (64-bit canonical, ntt1024.h, vs redundant 24-bit limbs, ntt1024_l4.h), timed at the occupancy of
// the bootstrap kernel (2 waves per SIMD, one 512-thread work-group per CU, LDS exchanges included);
// (2) issue cost of further single VALU instructions at the same occupancy.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../nufhe_amd/csrc microbench_l4.hip -o microbench_l4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "ff.h"
#include "ntt1024.h"
#include "ntt1024_l4.h"
#include "ntt_tables.h"

#ifndef REPS
#define REPS 64
#endif
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

__device__ __forceinline__ void load_tw(const u64 *g)
{
    u64 *t = (u64 *)smem;
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) t[i] = g[i];
    __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void k_transform(u64 *io, const i32 *d, const u64 *tw, long long *cycles)
{
    load_tw(tw);
    u64 *xbuf = (u64 *)smem + 2048 + (threadIdx.x >> 6) * NTT_XBUF_ELEMS;
    const u64 *twf = (const u64 *)smem, *twi = (const u64 *)smem + 1024;
    const NttLane L = ntt_lane_init(threadIdx.x & 63);
    const long base = (long)blockIdx.x * 512 * 16 + threadIdx.x;
    i32 dg[16];
    u64 x[16];
    u32 c[16];
    for (int r = 0; r < 16; r++) dg[r] = d[base + 512 * r];
    for (int r = 0; r < 16; r++) x[r] = io[base + 512 * r];
    const long long w0 = wall_clock64();
    const long long t0 = clock64();
    for (int it = 0; it < REPS; it++) {
        if (MODE == 0) {            // forward, 64-bit form
            ntt_forward_small(x, dg, xbuf, twf, L);
            for (int r = 0; r < 16; r++) dg[r] = (i32)((u32)x[r] & 1023u) - 512;
        } else if (MODE == 1) {     // forward, limb form
            ntt_forward_small_l4(x, dg, xbuf, twf, L);   // (timing only: the table layout does not matter)
            for (int r = 0; r < 16; r++) dg[r] = (i32)((u32)x[r] & 1023u) - 512;
        } else if (MODE == 2) {     // inverse, 64-bit form (+ conversion to int32)
            ntt_inverse_t<true>(x, xbuf, twi, L);
            for (int r = 0; r < 16; r++) x[r] = (u64)(u32)ff_to_i32(x[r]) * 0x9E3779B97F4A7C15ULL >> 1;
        } else {                    // inverse, limb form
            ntt_inverse_l4_i32(c, x, xbuf, twi, L);
            for (int r = 0; r < 16; r++) x[r] = (u64)c[r] * 0x9E3779B97F4A7C15ULL >> 1;
        }
    }
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    for (int r = 0; r < 16; r++) io[base + 512 * r] = x[r] + (u64)dg[r];
    if ((threadIdx.x & 63) == 0) {
        cycles[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
        cycles[2048 + blockIdx.x * 8 + (threadIdx.x >> 6)] = w1 - w0;     // constant-rate counter (hipDeviceAttributeWallClockRate)
    }
}

template <int MODE>
static void run_transform(const char *name, u64 *d_io, i32 *d_d, u64 *d_tw, long long *d_cyc)
{
    const int blocks = 256;
    const size_t lds = 2048 * 8 + 8 * NTT_XBUF_ELEMS * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_transform<MODE>, dim3(blocks), dim3(512), lds, 0, d_io, d_d, d_tw, d_cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_transform<MODE>, dim3(blocks), dim3(512), lds, 0, d_io, d_d, d_tw, d_cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> cyc(2 * blocks * 8);
    hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0, wall = 0;
    for (int i = 0; i < blocks * 8; i++) { avg += (double)cyc[i]; wall += (double)cyc[2048 + i]; }
    avg /= blocks * 8; wall /= blocks * 8;
    int wall_khz = 0;
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    // a wave shares its SIMD with one other wave: SIMD cycles per transform = wave-resident cycles / 2
    printf("%-34s %8.3f ms   %9.0f clock64 ticks per transform per wave, %7.1f us wall per transform; "
           "clock64 rate %.3f GHz (wall_clock64 at %d kHz)\n", name, ms,
           avg / REPS, ms * 1e3 / REPS, avg / (wall / (wall_khz * 1e3)) * 1e-9, wall_khz);
}

// ------------------------------------------------------------------------------------------
#define ITERS 2048
#define UNROLL 16
#define ASM1(str) asm volatile(str : "+v"(a[i]) : "v"(b), "v"(c), "s"(sc))

template <int OP>
__global__ __launch_bounds__(512, 2) void k_op(uint32_t *out, uint32_t seed, long long *cycles)
{
    uint32_t a[UNROLL], b = seed + threadIdx.x, c = (seed * 3 + 1) & 15, sc = seed * 5 + 7;
#pragma unroll
    for (int i = 0; i < UNROLL; i++) a[i] = seed + i * 977 + threadIdx.x;
    asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(b), "v"(a[0]) : "vcc");
    const long long t0 = clock64();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            if (OP == 0) ASM1("v_add_u32 %0, %0, %1");
            else if (OP == 1) ASM1("v_and_b32 %0, %0, %1");
            else if (OP == 2) ASM1("v_or_b32 %0, %0, %1");
            else if (OP == 3) ASM1("v_ashrrev_i32 %0, 3, %0");
            else if (OP == 4) ASM1("v_mov_b32 %0, %1");
            else if (OP == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
            else if (OP == 6) ASM1("v_perm_b32 %0, %0, %1, %3");
            else if (OP == 7) ASM1("v_lshl_add_u32 %0, %0, 3, %1");
            else if (OP == 8) ASM1("v_add3_u32 %0, %0, %1, %2");
            else if (OP == 9) ASM1("v_mad_i32_i24 %0, %0, %1, %2");
            else if (OP == 10) ASM1("v_bfm_b32 %0, %0, %1");
            else if (OP == 11) ASM1("v_sub_u32 %0, %3, %0");
            else if (OP == 12) ASM1("v_not_b32 %0, %0");
            else if (OP == 13) ASM1("v_lshlrev_b32 %0, %2, %0");
            else if (OP == 14) ASM1("v_lshrrev_b32 %0, %2, %0");
            else if (OP == 15) ASM1("v_subrev_u32 %0, %0, %1");
            else if (OP == 16) ASM1("v_mul_i32_i24 %0, %0, %1");
            else if (OP == 17) ASM1("v_and_or_b32 %0, %0, %1, %2");
            else if (OP == 18) ASM1("v_add_u32 %0, 0x12345, %0");
            else if (OP == 19) ASM1("v_max_i32 %0, %0, %1");
            else if (OP == 20) ASM1("v_bfe_i32 %0, %0, 3, 10");
            else if (OP == 21) ASM1("v_sub_u32 %0, %0, %1");
            else if (OP == 22) ASM1("v_xor_b32 %0, %0, %1");
            else if (OP == 23) ASM1("v_pk_add_u16 %0, %0, %1");
            else if (OP == 24) ASM1("v_pk_sub_i16 %0, %0, %1");
            else if (OP == 25) ASM1("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD src0_sel:DWORD src1_sel:WORD_0");
            else if (OP == 26) ASM1("v_add_u32 %0, %3, %0");
            else if (OP == 27) ASM1("v_and_b32 %0, 0xffffff, %0");
            else if (OP == 28) ASM1("v_lshlrev_b32 %0, 12, %0");
            else if (OP == 29) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*(uint64_t *)&a[i & ~1]) : "v"(b), "v"(c) : "vcc");
            else if (OP == 30) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(*(uint64_t *)&a[i & ~1]) : "v"(*(uint64_t *)&a[(i + 2) & 14]));
            else if (OP == 31) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            else if (OP == 32) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(*(double *)&a[i & ~1]));
        }
    }
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; i++) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
static void run_op(const char *name, uint32_t *d_out, long long *d_cyc)
{
    const int blocks = 256;     // one 8-wave work-group per CU = 2 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_op<OP>, dim3(blocks), dim3(512), 0, 0, d_out, 12345u, d_cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_op<OP>, dim3(blocks), dim3(512), 0, 0, d_out, 12345u, d_cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> cyc(blocks * 8);
    hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (long long v : cyc) avg += (double)v;
    avg /= cyc.size();
    // a wave shares its SIMD with one other wave running the same stream: while one wave is resident for
    // `avg` ticks the SIMD issues 2 x ITERS x UNROLL instructions
    printf("%-34s %8.3f ms   %6.2f clock64 ticks per wave-instruction per SIMD (2 waves/SIMD)\n", name, ms,
           avg / ((double)ITERS * UNROLL) / 2.0);
}

int main()
{
    std::vector<u64> tw(2048);
    ntt_make_tables(tw.data(), tw.data() + 1024);
    const long n = 256L * 512 * 16;
    std::vector<u64> io(n);
    std::vector<i32> dg(n);
    uint64_t s = 88172645463325252ULL;
    for (long i = 0; i < n; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        io[i] = s % FF_P;
        dg[i] = (i32)(s >> 40 & 1023) - 512;
    }
    u64 *d_io, *d_tw;
    i32 *d_d;
    long long *d_cyc;
    hipMalloc(&d_io, n * 8); hipMalloc(&d_tw, 2048 * 8); hipMalloc(&d_d, n * 4); hipMalloc(&d_cyc, 2 * 256 * 8 * 8);
    hipMemcpy(d_io, io.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_d, dg.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_tw, tw.data(), 2048 * 8, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k_transform<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void *)k_transform<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void *)k_transform<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void *)k_transform<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    printf("transforms: %d back-to-back per wave, 256 work-groups x 8 waves (2 waves per SIMD)\n", REPS);
    run_transform<0>("forward small, 64-bit form", d_io, d_d, d_tw, d_cyc);
    run_transform<1>("forward small, limb form", d_io, d_d, d_tw, d_cyc);
    run_transform<2>("inverse, 64-bit form", d_io, d_d, d_tw, d_cyc);
    run_transform<3>("inverse, limb form", d_io, d_d, d_tw, d_cyc);
    run_transform<0>("forward small, 64-bit form (again)", d_io, d_d, d_tw, d_cyc);
    run_transform<1>("forward small, limb form (again)", d_io, d_d, d_tw, d_cyc);

    uint32_t *d_out;
    hipMalloc(&d_out, 256 * 512 * 4);
    run_op<0>("v_add_u32", d_out, d_cyc);
    run_op<21>("v_sub_u32", d_out, d_cyc);
    run_op<15>("v_subrev_u32", d_out, d_cyc);
    run_op<11>("v_sub_u32 (sgpr src0)", d_out, d_cyc);
    run_op<18>("v_add_u32 (literal)", d_out, d_cyc);
    run_op<1>("v_and_b32", d_out, d_cyc);
    run_op<2>("v_or_b32", d_out, d_cyc);
    run_op<22>("v_xor_b32", d_out, d_cyc);
    run_op<12>("v_not_b32", d_out, d_cyc);
    run_op<4>("v_mov_b32", d_out, d_cyc);
    run_op<3>("v_ashrrev_i32 (const)", d_out, d_cyc);
    run_op<14>("v_lshrrev_b32 (vgpr amount)", d_out, d_cyc);
    run_op<13>("v_lshlrev_b32 (vgpr amount)", d_out, d_cyc);
    run_op<5>("v_cndmask_b32 (vcc, set once)", d_out, d_cyc);
    run_op<19>("v_max_i32", d_out, d_cyc);
    run_op<6>("v_perm_b32", d_out, d_cyc);
    run_op<7>("v_lshl_add_u32", d_out, d_cyc);
    run_op<8>("v_add3_u32", d_out, d_cyc);
    run_op<17>("v_and_or_b32", d_out, d_cyc);
    run_op<9>("v_mad_i32_i24", d_out, d_cyc);
    run_op<16>("v_mul_i32_i24", d_out, d_cyc);
    run_op<10>("v_bfm_b32", d_out, d_cyc);
    run_op<20>("v_bfe_i32", d_out, d_cyc);
    run_op<23>("v_pk_add_u16", d_out, d_cyc);
    run_op<24>("v_pk_sub_i16", d_out, d_cyc);
    run_op<25>("v_add_u32_sdwa", d_out, d_cyc);
    run_op<26>("v_add_u32 (sgpr src0)", d_out, d_cyc);
    run_op<27>("v_and_b32 (literal)", d_out, d_cyc);
    run_op<28>("v_lshlrev_b32 (const)", d_out, d_cyc);
    run_op<29>("v_mad_u64_u32", d_out, d_cyc);
    run_op<30>("v_lshl_add_u64", d_out, d_cyc);
    run_op<31>("v_addc_co_u32", d_out, d_cyc);
    run_op<32>("v_fma_f64", d_out, d_cyc);
    return 0;
}
