// microbench_ff.hip -- throughput of the GF(P) primitives in two formulations on gfx950:
//   C:   what the compiler makes of ff.h (selects: v_cmp + v_cndmask + 64-bit add)
//   ASM: EXEC-predicated correction (s_and_saveexec / one VALU add / restore): fewer VALU issues,
//        the mask bookkeeping moves to the scalar unit
// 2 waves per SIMD (256 blocks x 512 threads), CHAINS independent dependency chains per lane.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../nufhe_amd/csrc microbench_ff.hip -o microbench_ff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "ff.h"

#define ITERS 4096
#define CHAINS 8

__device__ __forceinline__ u64 sub_asm(u64 a, u64 b)
{
    u32 d0, d1;
    u64 sv;
    asm("v_sub_co_u32 %0, vcc, %3, %5\n\t"
        "v_subb_co_u32 %1, vcc, %4, %6, vcc\n\t"
        "s_and_saveexec_b64 %2, vcc\n\t"
        "v_add_co_u32 %0, vcc, 1, %0\n\t"
        "v_addc_co_u32 %1, vcc, -1, %1, vcc\n\t"
        "s_mov_b64 exec, %2"
        : "=&v"(d0), "=&v"(d1), "=&s"(sv)
        : "v"((u32)a), "v"((u32)(a >> 32)), "v"((u32)b), "v"((u32)(b >> 32))
        : "vcc", "scc");
    return ((u64)d1 << 32) | d0;
}

__device__ __forceinline__ u64 add_asm(u64 a, u64 b)
{
    u64 s, sv;
    asm("v_lshl_add_u64 %0, %2, 0, %3\n\t"
        "v_cmp_lt_u64 vcc, %0, %2\n\t"
        "v_cmp_le_u64 %1, %4, %0\n\t"
        "s_or_b64 vcc, vcc, %1\n\t"
        "s_and_saveexec_b64 %1, vcc\n\t"
        "v_lshl_add_u64 %0, %0, 0, %5\n\t"
        "s_mov_b64 exec, %1"
        : "=&v"(s), "=&s"(sv)
        : "v"(a), "v"(b), "s"((u64)FF_P), "s"((u64)FF_EPS)
        : "vcc", "scc");
    return s;
}

__device__ __forceinline__ u64 reduce96_asm(u64 lo, u32 h0)
{
    u64 s, sv;
    asm("v_mad_u64_u32 %0, vcc, %3, -1, %2\n\t"
        "v_cmp_le_u64 %1, %4, %0\n\t"
        "s_or_b64 vcc, vcc, %1\n\t"
        "s_and_saveexec_b64 %1, vcc\n\t"
        "v_lshl_add_u64 %0, %0, 0, %5\n\t"
        "s_mov_b64 exec, %1"
        : "=&v"(s), "=&s"(sv)
        : "v"(lo), "v"(h0), "s"((u64)FF_P), "s"((u64)FF_EPS)
        : "vcc", "scc");
    return s;
}

// fused butterfly: s = a + b, d = a - b (both canonical), the two correction chains interleaved
__device__ __forceinline__ void bfly_asm(u64 a, u64 b, u64 &s_out, u64 &d_out)
{
    u32 d0, d1;
    u64 s, m1, sv;
    asm("v_sub_co_u32 %[d0], vcc, %[a0], %[b0]\n\t"
        "v_lshl_add_u64 %[s], %[a], 0, %[b]\n\t"
        "v_subb_co_u32 %[d1], vcc, %[a1], %[b1], vcc\n\t"
        "v_cmp_lt_u64 %[m1], %[s], %[a]\n\t"
        "s_and_saveexec_b64 %[sv], vcc\n\t"
        "v_add_co_u32 %[d0], vcc, 1, %[d0]\n\t"
        "v_addc_co_u32 %[d1], vcc, -1, %[d1], vcc\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "v_cmp_le_u64 vcc, %[P], %[s]\n\t"
        "s_or_b64 vcc, vcc, %[m1]\n\t"
        "s_and_saveexec_b64 %[sv], vcc\n\t"
        "v_lshl_add_u64 %[s], %[s], 0, %[eps]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [d0] "=&v"(d0), [d1] "=&v"(d1), [s] "=&v"(s), [m1] "=&s"(m1), [sv] "=&s"(sv)
        : [a] "v"(a), [b] "v"(b), [a0] "v"((u32)a), [a1] "v"((u32)(a >> 32)), [b0] "v"((u32)b),
          [b1] "v"((u32)(b >> 32)), [P] "s"((u64)FF_P), [eps] "s"((u64)FF_EPS)
        : "vcc", "scc");
    s_out = s;
    d_out = ((u64)d1 << 32) | d0;
}

template <int OP>
__global__ __launch_bounds__(512, 2) void k(u64 *out, u64 seed)
{
    u64 x[CHAINS], y[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; i++) {
        x[i] = (seed * (threadIdx.x + 1) + i * 0x9E3779B97F4A7C15ULL) % FF_P;
        y[i] = (seed * 31 + threadIdx.x * 0x123456789ULL + i) % FF_P;
    }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) {
            if (OP == 0) x[i] = ff_sub(x[i], y[i]);
            else if (OP == 1) x[i] = sub_asm(x[i], y[i]);
            else if (OP == 2) x[i] = ff_add(x[i], y[i]);
            else if (OP == 3) x[i] = add_asm(x[i], y[i]);
            else if (OP == 4) x[i] = ff_reduce96(x[i], (u32)y[i]);
            else if (OP == 5) x[i] = reduce96_asm(x[i], (u32)y[i]);
            else if (OP == 6) { u64 t = ff_add(x[i], y[i]); y[i] = ff_sub(x[i], y[i]); x[i] = t; }       // butterfly
            else if (OP == 7) { u64 t = add_asm(x[i], y[i]); y[i] = sub_asm(x[i], y[i]); x[i] = t; }
            else if (OP == 8) x[i] = ff_mul(x[i], y[i]);
            else if (OP == 9) { u64 t, d; bfly_asm(x[i], y[i], t, d); x[i] = t; y[i] = d; }
        }
    }
    u64 r = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; i++) r ^= x[i] ^ y[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
void run(const char *name, u64 *d_out, u64 *h)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(512), 0, 0, d_out, 12345ull);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(512), 0, 0, d_out, 12345ull);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, d_out, 512 * sizeof(u64), hipMemcpyDeviceToHost);
    u64 chk = 0;
    for (int i = 0; i < 512; i++) chk ^= h[i] * (i + 1);
    double wave_ops = (double)blocks * 8 * ITERS * CHAINS;
    double per_simd_per_s = wave_ops / (ms * 1e-3) / 1024.0;
    printf("%-28s %8.3f ms  %7.2f cycles per op per SIMD (@2.4 GHz)  checksum %016llx\n", name, ms,
           2.4e9 / per_simd_per_s, (unsigned long long)chk);
}

int main()
{
    u64 *d_out, h[512];
    hipMalloc(&d_out, 256 * 512 * sizeof(u64));
    run<0>("ff_sub  (C)", d_out, h);
    run<1>("ff_sub  (asm, predicated)", d_out, h);
    run<2>("ff_add  (C)", d_out, h);
    run<3>("ff_add  (asm, predicated)", d_out, h);
    run<4>("reduce96 (C)", d_out, h);
    run<5>("reduce96 (asm, predicated)", d_out, h);
    run<6>("butterfly add+sub (C)", d_out, h);
    run<7>("butterfly add+sub (asm)", d_out, h);
    run<8>("ff_mul (C)", d_out, h);
    run<9>("butterfly fused asm", d_out, h);
    return 0;
}
