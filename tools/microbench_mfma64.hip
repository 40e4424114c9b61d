// microbench_mfma64.hip -- does the fp64 MATRIX pipe of gfx950 run beside the fp64 VECTOR pipe?
//
// Paper gate of VERDICT r4 item 2 (pass 2 of the FFT-512 as a 16 x 16 real DFT-8 matrix product on
// v_mfma_f64_16x16x4_f64): the idea only pays if the matrix instructions (a) issue at the rate the 78.6 TFLOP/s
// matrix peak implies (64 cycles per SIMD each) and (b) OVERLAP with the v_fma_f64 / v_add_f64 stream of the same
// wave or of the partner wave on the same SIMD -- the dense product costs ~5 x the flops of the radix-8 butterflies
// it replaces, so it must be free time.  Measured here, per SIMD, with every CU busy (the clock under load counts):
//   m   : MFMA only (4 independent accumulators)
//   v   : v_fma_f64 only (8 independent chains)
//   i   : v_add_u32 only
//   mvK : one wave, K v_fma_f64 after every MFMA (K = 4, 8, 12, 16)
//   miK : one wave, K v_add_u32 after every MFMA
//   m|v : two waves per SIMD, one MFMA-only, its partner v_fma_f64-only (same instruction counts as m and v alone)
//   m|i : two waves per SIMD, one MFMA-only, its partner v_add_u32-only
// Output: shader cycles per MFMA (or per VALU instruction) per wave, the sustained clock, for 1 and 2 waves per SIMD.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 microbench_mfma64.hip -o microbench_mfma64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

#define N_MFMA 2048          /* per wave */

__device__ __forceinline__ uint64_t shader_clock() { return clock64(); }
__device__ __forceinline__ uint64_t real_clock() { return wall_clock64(); }

// role 0: MFMA with K_F64 v_fma_f64 + K_I32 v_add_u32 after each; role 1: VALU only, V_PER per "slot"
template <int K_F64, int K_I32>
__device__ __forceinline__ void mfma_stream(d4 (&acc)[4], double a, double b, double (&w)[8], uint32_t (&u)[8])
{
#pragma unroll 1
    for (int it = 0; it < N_MFMA / 16; it++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            acc[j & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j & 3], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K_F64; k++) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(w[k & 7]));
#pragma unroll
            for (int k = 0; k < K_I32; k++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[k & 7]) : "v"(u[(k + 1) & 7]));
        }
    }
}

template <int F64>
__device__ __forceinline__ void valu_stream(int n, double (&w)[8], uint32_t (&u)[8])
{
#pragma unroll 1
    for (int it = 0; it < n / 64; it++) {
#pragma unroll
        for (int k = 0; k < 64; k++) {
            if (F64) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(w[k & 7]));
            else asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[k & 7]) : "v"(u[(k + 1) & 7]));
        }
    }
}

// MODE: 0 m, 1 v, 2 i, 3 mv4, 4 mv8, 5 mv12, 6 mv16, 7 mi8, 8 mi16, 9 m|v, 10 m|i, 11 m|m (two MFMA waves)
template <int MODE>
__global__ void k_bench(uint64_t *out, double seed, int valu_n)
{
    extern __shared__ char pin[];          // sized by the host so that ONE work-group fits a CU
    const int wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    d4 acc[4];
    for (int i = 0; i < 4; i++) acc[i] = d4{seed, seed, seed, seed};
    double w[8];
    uint32_t u[8];
    for (int i = 0; i < 8; i++) { w[i] = seed * (i + 1) * 1e-3; u[i] = (uint32_t)threadIdx.x * (i + 3); }
    const double a = seed * 1e-9 * (threadIdx.x & 15), b = seed * 1e-9 * (threadIdx.x >> 4);
    __syncthreads();
    const uint64_t t0 = shader_clock(), r0 = real_clock();
    const bool second = waves == 8 && wave >= 4;     // waves w and w + 4 share a SIMD
    if (MODE == 0 || (MODE == 9 && !second) || (MODE == 10 && !second) || MODE == 11) mfma_stream<0, 0>(acc, a, b, w, u);
    else if (MODE == 1 || (MODE == 9 && second)) valu_stream<1>(valu_n, w, u);
    else if (MODE == 2 || (MODE == 10 && second)) valu_stream<0>(valu_n, w, u);
    else if (MODE == 3) mfma_stream<4, 0>(acc, a, b, w, u);
    else if (MODE == 4) mfma_stream<8, 0>(acc, a, b, w, u);
    else if (MODE == 5) mfma_stream<12, 0>(acc, a, b, w, u);
    else if (MODE == 6) mfma_stream<16, 0>(acc, a, b, w, u);
    else if (MODE == 7) mfma_stream<0, 8>(acc, a, b, w, u);
    else if (MODE == 8) mfma_stream<0, 16>(acc, a, b, w, u);
    const uint64_t t1 = shader_clock(), r1 = real_clock();
    double s = 0;
    for (int i = 0; i < 4; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    for (int i = 0; i < 8; i++) s += w[i] + (double)u[i];
    if (s == 1.2345e-300) out[1023] = 1;              // keep everything alive
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) {
        out[2 * wave] = t1 - t0;
        out[2 * wave + 1] = r1 - r0;
    }
}

template <int MODE>
static void run(const char *name, int waves_per_simd, int valu_n, double per_wave_units, const char *unit, uint64_t *d_out,
                int cus)
{
    const int threads = 256 * waves_per_simd;
    const size_t lds = 96 * 1024;                     // > half of 160 KiB: one work-group per CU
    hipFuncSetAttribute((const void *)k_bench<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_bench<MODE>, dim3(cus), dim3(threads), lds, 0, d_out, 1.0, valu_n);     // warm-up
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_bench<MODE>, dim3(cus), dim3(threads), lds, 0, d_out, 1.0, valu_n);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    uint64_t h[16];
    hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / ((double)h[1] * 10.0);
    printf("%-6s %d wave(s)/SIMD: wave0 %8.0f cycles = %6.2f cycles per %s", name, waves_per_simd, (double)h[0],
           (double)h[0] / per_wave_units, unit);
    if (waves_per_simd == 2) printf(" | wave4 %8.0f cycles", (double)h[8]);
    printf(" | clock %.2f GHz, kernel %.3f ms\n", ghz, ms);
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs; %d MFMAs (v_mfma_f64_16x16x4_f64) per wave\n", prop.gcnArchName, cus, N_MFMA);
    uint64_t *d_out;
    hipMalloc(&d_out, 1024 * 8);
    hipMemset(d_out, 0, 1024 * 8);
    const int VN = N_MFMA * 16;      // VALU-only streams: 16 instructions per "MFMA slot"
    for (int wps = 1; wps <= 2; wps++) {
        run<0>("m", wps, 0, N_MFMA, "MFMA", d_out, cus);
        run<1>("v", wps, VN, VN, "v_fma_f64", d_out, cus);
        run<2>("i", wps, VN, VN, "v_add_u32", d_out, cus);
        run<3>("mv4", wps, 0, N_MFMA, "MFMA + 4 v_fma_f64", d_out, cus);
        run<4>("mv8", wps, 0, N_MFMA, "MFMA + 8 v_fma_f64", d_out, cus);
        run<5>("mv12", wps, 0, N_MFMA, "MFMA + 12 v_fma_f64", d_out, cus);
        run<6>("mv16", wps, 0, N_MFMA, "MFMA + 16 v_fma_f64", d_out, cus);
        run<7>("mi8", wps, 0, N_MFMA, "MFMA + 8 v_add_u32", d_out, cus);
        run<8>("mi16", wps, 0, N_MFMA, "MFMA + 16 v_add_u32", d_out, cus);
    }
    run<9>("m|v", 2, VN, N_MFMA, "MFMA (wave0); wave4 = the v_fma_f64 stream of `v`", d_out, cus);
    run<10>("m|i", 2, VN, N_MFMA, "MFMA (wave0); wave4 = the v_add_u32 stream of `i`", d_out, cus);
    run<11>("m|m", 2, 0, N_MFMA, "MFMA, both waves of the SIMD MFMA-only", d_out, cus);
    return 0;
}
