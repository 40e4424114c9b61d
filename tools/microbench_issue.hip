// microbench_issue.hip -- VALU issue cost on gfx950 as a function of OCCUPANCY (1, 2, 4, 8 waves per SIMD), for the
// four issue classes the bootstrap kernel's cycle model uses (plain VOP1/VOP2, VOP2 with SGPR/literal, VOP3, the
// 64-bit multiply-add) plus v_fma_f32 / v_fma_f64 as the anchors MI355X_MICROARCH.md quotes.
//
// Every wave runs ITERS x BODY independent-chain instructions (CHAINS chains per lane, BODY = 256 instructions per
// loop trip so that the 3 scalar loop instructions are < 1.5 % of the stream) and reads s_memtime around the loop.
// cycles per wave-instruction per SIMD = (wave-resident ticks) / (instructions of ONE wave) / (waves per SIMD).
// A second pass uses ONE dependent chain (CHAINS = 1): the back-to-back issue latency of a single wave.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 microbench_issue.hip -o microbench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITERS 512
#define BODY 256

#define A1(str) asm volatile(str : "+v"(a[i % CH]) : "v"(b), "v"(c), "s"(sc))

template <int OP, int CH>
__device__ __forceinline__ void body(uint32_t (&a)[16], uint64_t (&w)[8], uint32_t b, uint32_t c, uint32_t sc)
{
#pragma unroll
    for (int i = 0; i < BODY; i++) {
        if (OP == 0) A1("v_add_u32 %0, %0, %1");
        else if (OP == 1) A1("v_xor_b32 %0, %0, %1");
        else if (OP == 2) A1("v_mov_b32 %0, %1");
        else if (OP == 3) A1("v_lshrrev_b32 %0, 3, %0");
        else if (OP == 4) A1("v_add_u32 %0, %3, %0");                      // SGPR operand
        else if (OP == 5) A1("v_add_u32 %0, 0x12345, %0");                 // literal
        else if (OP == 6) A1("v_lshlrev_b32 %0, 12, %0");
        else if (OP == 7) A1("v_lshl_add_u32 %0, %0, 3, %1");              // VOP3
        else if (OP == 8) A1("v_add3_u32 %0, %0, %1, %2");
        else if (OP == 9) A1("v_perm_b32 %0, %0, %1, %2");
        else if (OP == 10) A1("v_bfe_i32 %0, %0, 3, 10");
        else if (OP == 11) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i % (CH > 8 ? 8 : CH)]) : "v"(b), "v"(c) : "vcc");
        else if (OP == 12) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w[i % (CH > 8 ? 8 : CH)]) : "v"(w[(i + 1) % 8]));
        else if (OP == 13) A1("v_fma_f32 %0, %0, %1, %2");
        else if (OP == 14) A1("v_fmac_f32 %0, %1, %2");
        else if (OP == 15) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(w[i % (CH > 8 ? 8 : CH)]));
        else if (OP == 16) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(w[i % (CH > 8 ? 8 : CH)]));
        else if (OP == 17) A1("v_mul_lo_u32 %0, %0, %1");
        else if (OP == 18) A1("v_mul_u32_u24 %0, %0, %1");
        else if (OP == 19) A1("v_mad_u32_u24 %0, %0, %1, %2");
        else if (OP == 20) A1("v_sub_u32 %0, %0, %1");
        else if (OP == 21) A1("v_and_b32 %0, 0xffffff, %0");
        else if (OP == 22) A1("v_ashrrev_i32 %0, 24, %0");
        else if (OP == 23) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i % CH]) : "v"(b) : "vcc");
        else if (OP == 24) A1("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD src0_sel:DWORD src1_sel:WORD_0");
        else if (OP == 25) A1("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
        else if (OP == 26) A1("v_sad_u32 %0, %0, %1, %2");
        else if (OP == 27) A1("v_mad_i32_i24 %0, %0, %1, %2");
        else if (OP == 28) A1("v_mul_hi_u32 %0, %0, %1");
        else if (OP == 29) {   // the kernel's typical blend: 3 plain : 2 VOP3
            if (i % 5 < 3) A1("v_add_u32 %0, %0, %1"); else A1("v_lshl_add_u32 %0, %0, 3, %1");
        }
        else if (OP == 30) A1("v_pk_add_u16 %0, %0, %1");
        else if (OP == 31) A1("v_dot4_i32_i8 %0, %1, %2, %0");
        else if (OP == 32) A1("v_alignbit_b32 %0, %0, %1, 24");
        else if (OP == 33) A1("v_and_or_b32 %0, %0, %1, %2");
        else if (OP == 34) A1("v_lshl_or_b32 %0, %0, 8, %1");
        else if (OP == 60) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(w[i % 8]) : "v"(a[i % 16]));
        else if (OP == 61) asm volatile("v_add_f64 %0, %0, %0" : "+v"(w[i % 8]));
        else if (OP == 62) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(w[i % 8]));
        else if (OP == 63) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(a[i % 16]) : "v"(w[i % 8]));
        else if (OP == 64) asm volatile("v_rndne_f64 %0, %0" : "+v"(w[i % 8]));
        else if (OP == 65) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(w[i % 8]) : "v"(c));
        else if (OP == 66) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i % 16]) : "v"(b));
        else if (OP == 67) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i % 16]) : "v"(b), "s"(((uint64_t)sc << 32) | sc));
        else if (OP == 68) asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a[i % 16]), "v"(b) : "vcc");
        else if (OP == 69) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(w[i % 8]));
        else if (OP == 70) asm volatile("v_bfe_u32 %0, %0, 3, 10" : "+v"(a[i % 16]));
        else if (OP == 71) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(w[i % 8]) : "v"(a[i % 16]));
        else if (OP == 72) { if (i % 2 == 0) asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a[i % 16]), "v"(b) : "vcc"); else asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i % 16]) : "v"(b)); }
        else if (OP == 73) { if (i % 5 == 0) asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a[i % 16]), "v"(b) : "vcc"); else asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i % 16]) : "v"(b)); }
        else if (OP == 74) { if (i % 5 == 0) asm volatile("v_cmp_lt_u32 s[40:41], %0, %1" :: "v"(a[i % 16]), "v"(b) : "s40", "s41"); else asm volatile("v_cndmask_b32 %0, %0, %1, s[40:41]" : "+v"(a[i % 16]) : "v"(b)); }
        else if (OP == 75) { if (i % 64 == 0) asm volatile("s_mov_b64 vcc, s[40:41]" ::: "vcc"); else asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i % 16]) : "v"(b)); }
        else if (OP == 76) { if (i % 64 == 0) asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a[i % 16]), "v"(b) : "vcc"); else asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i % 16]) : "v"(b)); }
        else if (OP == 77) { if (i % 64 == 0) asm volatile("v_cmp_lt_u32 s[40:41], %0, %1" :: "v"(a[i % 16]), "v"(b) : "s40", "s41"); else asm volatile("v_cndmask_b32 %0, %0, %1, s[40:41]" : "+v"(a[i % 16]) : "v"(b)); }
        else if (OP == 78) asm volatile("v_addc_co_u32 %0, s[40:41], %0, %1, s[40:41]" : "+v"(a[i % 16]) : "v"(b) : "s40", "s41");
        else if (OP == 79) { if (i % 2 == 0) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i % 16]) : "v"(b) : "vcc"); else asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i % 16]) : "v"(b) : "vcc"); }
        else if (OP == 40) { if (i % 16 < 8) A1("v_add_u32 %0, %0, %1"); else A1("v_lshl_add_u32 %0, %0, 3, %1"); }
        else if (OP == 41) { if (i % 64 < 32) A1("v_add_u32 %0, %0, %1"); else A1("v_lshl_add_u32 %0, %0, 3, %1"); }
        else if (OP == 42) { if (i % 256 < 128) A1("v_add_u32 %0, %0, %1"); else A1("v_lshl_add_u32 %0, %0, 3, %1"); }
        else if (OP == 43) { if (i % 2 < 1) A1("v_add_u32 %0, %0, %1"); else A1("v_lshl_add_u32 %0, %0, 3, %1"); }
        else if (OP == 44) { if (i % 4 < 2) A1("v_add_u32 %0, %0, %1"); else A1("v_lshl_add_u32 %0, %0, 3, %1"); }
        else if (OP == 45) { if (i % 4 < 3) A1("v_add_u32 %0, %0, %1"); else A1("v_lshl_add_u32 %0, %0, 3, %1"); }
        else if (OP == 46) { if (i % 8 < 7) A1("v_add_u32 %0, %0, %1"); else A1("v_lshl_add_u32 %0, %0, 3, %1"); }
        else if (OP == 47) { if (i % 4 < 3) A1("v_add_u32 %0, %0, %1"); else asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i % 8]) : "v"(b), "v"(c) : "vcc"); }
        else if (OP == 48) { if (i % 2 < 1) A1("v_add_u32 %0, %0, %1"); else A1("v_and_b32 %0, 0xffffff, %0"); }
        else if (OP == 50) { if (i % 256 == 0) __builtin_amdgcn_s_barrier(); if (i % 256 < 128) A1("v_add_u32 %0, %0, %1"); else A1("v_lshl_add_u32 %0, %0, 3, %1"); }
        else if (OP == 51) { if (i % 128 == 0) __builtin_amdgcn_s_barrier(); if (i % 128 < 64) A1("v_add_u32 %0, %0, %1"); else A1("v_lshl_add_u32 %0, %0, 3, %1"); }
        else if (OP == 52) { if (i % 64 == 0) __builtin_amdgcn_s_barrier(); if (i % 64 < 32) A1("v_add_u32 %0, %0, %1"); else A1("v_lshl_add_u32 %0, %0, 3, %1"); }
        else if (OP == 53) { if (i % 32 == 0) __builtin_amdgcn_s_barrier(); if (i % 32 < 16) A1("v_add_u32 %0, %0, %1"); else A1("v_lshl_add_u32 %0, %0, 3, %1"); }
        else if (OP == 54) { if (i % 256 == 0) __builtin_amdgcn_s_barrier(); if (i % 256 < 192) A1("v_add_u32 %0, %0, %1"); else A1("v_lshl_add_u32 %0, %0, 3, %1"); }
        else if (OP == 55) { if (i % 64 == 0) __builtin_amdgcn_s_barrier(); A1("v_add_u32 %0, %0, %1"); }
        else if (OP == 56) { if (i % 64 == 0) __builtin_amdgcn_s_barrier(); if (i % 64 < 48) A1("v_add_u32 %0, %0, %1"); else A1("v_lshl_add_u32 %0, %0, 3, %1"); }
        else if (OP == 90) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i % 16]), "+v"(a[(i + 8) % 16]));
        else if (OP == 91) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i % 16]), "+v"(a[(i + 8) % 16]));
        else if (OP == 92) { if (i % 8 == 0) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i % 16]), "+v"(a[(i + 8) % 16])); else asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(w[i % 8])); }
        else if (OP == 49) { if (i % 2 < 1) A1("v_add_u32 %0, %0, %1"); else A1("v_add_u32 %0, %3, %0"); }
    }
}

template <int OP, int CH, int THREADS>
__global__ __launch_bounds__(THREADS) void k_op(uint32_t *out, uint32_t seed, long long *cycles)
{
    uint32_t a[16], b = seed + threadIdx.x, c = (seed * 3 + 1) & 15, sc = seed * 5 + 7;
    uint64_t w[8];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = seed + i * 977 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = ((uint64_t)a[2 * i] << 32) | a[2 * i + 1];
    extern __shared__ uint32_t lds_pad[];          // only there to pin the number of work-groups per CU
    if (seed == 0xffffffffu) lds_pad[threadIdx.x] = seed;
    __syncthreads();
    const long long r0 = wall_clock64();
    const long long t0 = clock64();
    for (int it = 0; it < ITERS; it++) body<OP, CH>(a, w, b, c, sc);
    const long long t1 = clock64();
    const long long r1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        cycles[(blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6)) * 2] = t1 - t0;
        cycles[(blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0;
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) r ^= a[i];
#pragma unroll
    for (int i = 0; i < 8; i++) r ^= (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

static int g_cus = 256;
static int g_wall_khz = 100000;
static double g_last_ghz = 0, g_last_ms = 0, g_last_wave_avg = 0;

template <int OP, int CH, int THREADS, int BLOCKS_PER_CU>
static double run_one(uint32_t *d_out, long long *d_cyc, double *wall_cyc = nullptr)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = g_cus * BLOCKS_PER_CU;
    const int waves = blocks * THREADS / 64;
    // dynamic LDS sized so that EXACTLY BLOCKS_PER_CU work-groups fit a CU (160 KiB): without it the dispatcher
    // doubles work-groups up on some CUs and leaves others empty
    const size_t lds = BLOCKS_PER_CU == 1 ? 100 * 1024 : 70 * 1024;
    hipFuncSetAttribute((const void *)k_op<OP, CH, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((k_op<OP, CH, THREADS>), dim3(blocks), dim3(THREADS), lds, 0, d_out, 12345u, d_cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_op<OP, CH, THREADS>), dim3(blocks), dim3(THREADS), lds, 0, d_out, 12345u, d_cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    // the same figure from the launch's wall time at the nominal 2.4 GHz (includes launch overhead and the tail)
    if (wall_cyc) *wall_cyc = ms * 1e-3 * 2.4e9 / ((double)ITERS * BODY) / ((double)THREADS * BLOCKS_PER_CU / 256.0);
    std::vector<long long> cyc(waves * 2);
    hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0, real = 0;
    for (int i = 0; i < waves; i++) { avg += (double)cyc[2 * i]; real += (double)cyc[2 * i + 1]; }
    avg /= waves; real /= waves;
    g_last_ghz = avg / (real / g_wall_khz * 1e-3) * 1e-9;      // s_memtime ticks per second of s_memrealtime
    g_last_ms = ms;
    const double waves_per_simd = (double)THREADS * BLOCKS_PER_CU / 256.0;
    g_last_wave_avg = avg / ((double)ITERS * BODY) / waves_per_simd;   // mean wave life time: UNDER-estimates when the arbiter
                                                                         // serves the oldest wave first and the others finish late
    // what the SIMD sustains: launch duration x the shader clock measured inside it / instructions per SIMD (the launch
    // overhead, ~5 us of >= 240, is included)
    return ms * 1e-3 * g_last_ghz * 1e9 / ((double)ITERS * BODY * waves_per_simd);
}

template <int OP>
static void run_op(const char *name, uint32_t *d_out, long long *d_cyc)
{
    // independent chains (16 per lane; 8 for the 64-bit forms) at 1, 2, 4, 8 waves per SIMD; one dependent chain at 1 and 2
    const double w1 = run_one<OP, 16, 256, 1>(d_out, d_cyc);
    const double w2 = run_one<OP, 16, 512, 1>(d_out, d_cyc);
    const double w3 = run_one<OP, 16, 768, 1>(d_out, d_cyc);
    const double w4 = run_one<OP, 16, 1024, 1>(d_out, d_cyc);
    const double g4 = g_last_ghz, a4 = g_last_wave_avg;
    double wall8 = 0;
    const double w8 = run_one<OP, 16, 1024, 2>(d_out, d_cyc, &wall8);
    const double g8 = g_last_ghz, a8 = g_last_wave_avg;
    const double d1 = run_one<OP, 1, 256, 1>(d_out, d_cyc);
    const double d2 = run_one<OP, 1, 512, 1>(d_out, d_cyc);
    printf("%-28s %6.2f %6.2f %6.2f %6.2f %6.2f   | dependent chain: %6.2f %6.2f | clock %.2f / %.2f GHz, mean wave life / instructions %.2f / %.2f at 4 / 8 waves\n", name, w1, w2, w3, w4, w8, d1, d2, g4, g8, a4, a8);
    fflush(stdout);
}

// ---- two DIFFERENT streams on one SIMD: waves 0-3 of a 512-thread work-group run stream A, waves 4-7 stream B (wave w and
// w+4 share a SIMD; the host checks that through HW_REG_HW_ID).  Both streams loop until BOTH have done their ITERS trips
// (a finished wave keeps running its stream, so the other never runs alone); each wave reports ticks per own instruction
// over its first ITERS trips.
template <int OPA, int OPB, int PRIO>
__global__ __launch_bounds__(512) void k_pair(uint32_t *out, uint32_t seed, long long *cycles)
{
    extern __shared__ uint32_t lds_pad[];
    uint32_t a[16], b = seed + threadIdx.x, c = (seed * 3 + 1) & 15, sc = seed * 5 + 7;
    uint64_t w[8];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = seed + i * 977 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = ((uint64_t)a[2 * i] << 32) | a[2 * i + 1];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    volatile uint32_t *done = lds_pad;       // done[wave] = 1 when the wave has finished its measured trips
    if (threadIdx.x < 8) lds_pad[threadIdx.x] = 0;
    __syncthreads();
    long long t0, t1 = 0;
    const int partner = wave ^ 4;
    if (wave < 4) {
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        t0 = clock64();
        for (int it = 0;; it++) {
            body<OPA, 16>(a, w, b, c, sc);
            if (it == ITERS - 1) { t1 = clock64(); done[wave] = 1; }
            if (it >= ITERS - 1 && done[partner]) break;
        }
    } else {
        t0 = clock64();
        for (int it = 0;; it++) {
            body<OPB, 16>(a, w, b, c, sc);
            if (it == ITERS - 1) { t1 = clock64(); done[wave] = 1; }
            if (it >= ITERS - 1 && done[partner]) break;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        cycles[(blockIdx.x * 8 + wave) * 2] = t1 - t0;
        cycles[(blockIdx.x * 8 + wave) * 2 + 1] = (hwid >> 4) & 3;
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) r ^= a[i];
#pragma unroll
    for (int i = 0; i < 8; i++) r ^= (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OPA, int OPB, int PRIO>
static void run_pair(const char *name, uint32_t *d_out, long long *d_cyc)
{
    const int blocks = g_cus;
    hipFuncSetAttribute((const void *)k_pair<OPA, OPB, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((k_pair<OPA, OPB, PRIO>), dim3(blocks), dim3(512), 100 * 1024, 0, d_out, 12345u, d_cyc);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k_pair<OPA, OPB, PRIO>), dim3(blocks), dim3(512), 100 * 1024, 0, d_out, 12345u, d_cyc);
    hipDeviceSynchronize();
    std::vector<long long> cyc(blocks * 16);
    hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost);
    double ta = 0, tb = 0;
    int mismatched = 0;
    for (int blk = 0; blk < blocks; blk++) {
        int count[4] = {0, 0, 0, 0}, grp[4] = {0, 0, 0, 0};
        for (int wv = 0; wv < 8; wv++) {
            const int simd = (int)cyc[(blk * 8 + wv) * 2 + 1];
            count[simd]++;
            grp[simd] += wv < 4 ? 1 : 0;
            (wv < 4 ? ta : tb) += (double)cyc[(blk * 8 + wv) * 2];
        }
        for (int sd = 0; sd < 4; sd++) mismatched += !(count[sd] == 2 && grp[sd] == 1);
    }
    ta /= blocks * 4.0 * ITERS * BODY; tb /= blocks * 4.0 * ITERS * BODY;
    printf("%-44s A %6.2f  B %6.2f cycles per own instruction -> %5.2f cycles per instruction per SIMD%s\n", name, ta, tb,
           1.0 / (1.0 / ta + 1.0 / tb), mismatched ? "  [SIMD pairing not (w, w+4) everywhere]" : "");
    fflush(stdout);
}

int main(int argc, char **argv)
{
    if (argc > 1 && argv[1][0] == 'p') {         // only the cross-lane swaps of gfx950 (round 4)
        hipDeviceProp_t p;
        hipGetDeviceProperties(&p, 0);
        g_cus = p.multiProcessorCount;
        hipDeviceGetAttribute(&g_wall_khz, hipDeviceAttributeWallClockRate, 0);
        uint32_t *d_out;
        long long *d_cyc;
        hipMalloc(&d_out, (size_t)g_cus * 2 * 1024 * 4);
        hipMalloc(&d_cyc, (size_t)g_cus * 2 * 16 * 8 * 2);
        printf("%-28s %6s %6s %6s %6s %6s\n", "opcode \\ waves per SIMD", "1", "2", "3", "4", "8");
        run_op<15>("v_fma_f64", d_out, d_cyc);
        run_op<90>("v_permlane32_swap_b32", d_out, d_cyc);
        run_op<91>("v_permlane16_swap_b32", d_out, d_cyc);
        run_op<92>("1 permlane32_swap : 7 fma_f64", d_out, d_cyc);
        run_op<25>("v_mov_b32_dpp", d_out, d_cyc);
        return 0;
    }
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    g_cus = p.multiProcessorCount;
    hipDeviceGetAttribute(&g_wall_khz, hipDeviceAttributeWallClockRate, 0);
    uint32_t *d_out;
    long long *d_cyc;
    hipMalloc(&d_out, (size_t)g_cus * 2 * 1024 * 4);
    hipMalloc(&d_cyc, (size_t)g_cus * 2 * 16 * 8 * 2);
    printf("device %s, %d CUs; shader cycles per wave64 instruction per SIMD = launch duration x in-kernel clock (s_memtime / s_memrealtime) / instructions per SIMD\n",
           p.gcnArchName, g_cus);
    printf("%-28s %6s %6s %6s %6s %6s   | %s\n", "opcode \\ waves per SIMD", "1", "2", "3", "4", "8", "one chain, 1 and 2 waves per SIMD");
    run_op<0>("v_add_u32", d_out, d_cyc);
    run_op<20>("v_sub_u32", d_out, d_cyc);
    run_op<1>("v_xor_b32", d_out, d_cyc);
    run_op<2>("v_mov_b32", d_out, d_cyc);
    run_op<3>("v_lshrrev_b32 const", d_out, d_cyc);
    run_op<22>("v_ashrrev_i32 const", d_out, d_cyc);
    run_op<4>("v_add_u32 sgpr", d_out, d_cyc);
    run_op<5>("v_add_u32 literal", d_out, d_cyc);
    run_op<21>("v_and_b32 literal", d_out, d_cyc);
    run_op<6>("v_lshlrev_b32 const", d_out, d_cyc);
    run_op<7>("v_lshl_add_u32", d_out, d_cyc);
    run_op<8>("v_add3_u32", d_out, d_cyc);
    run_op<26>("v_sad_u32", d_out, d_cyc);
    run_op<9>("v_perm_b32", d_out, d_cyc);
    run_op<10>("v_bfe_i32", d_out, d_cyc);
    run_op<32>("v_alignbit_b32", d_out, d_cyc);
    run_op<33>("v_and_or_b32", d_out, d_cyc);
    run_op<34>("v_lshl_or_b32", d_out, d_cyc);
    run_op<24>("v_add_u32_sdwa", d_out, d_cyc);
    run_op<25>("v_mov_b32_dpp", d_out, d_cyc);
    run_op<30>("v_pk_add_u16", d_out, d_cyc);
    run_op<11>("v_mad_u64_u32", d_out, d_cyc);
    run_op<12>("v_lshl_add_u64", d_out, d_cyc);
    run_op<23>("v_addc_co_u32", d_out, d_cyc);
    run_op<17>("v_mul_lo_u32", d_out, d_cyc);
    run_op<28>("v_mul_hi_u32", d_out, d_cyc);
    run_op<18>("v_mul_u32_u24", d_out, d_cyc);
    run_op<19>("v_mad_u32_u24", d_out, d_cyc);
    run_op<27>("v_mad_i32_i24", d_out, d_cyc);
    run_op<31>("v_dot4_i32_i8", d_out, d_cyc);
    run_op<13>("v_fma_f32", d_out, d_cyc);
    run_op<14>("v_fmac_f32", d_out, d_cyc);
    run_op<16>("v_pk_fma_f32", d_out, d_cyc);
    run_op<15>("v_fma_f64", d_out, d_cyc);
    run_op<29>("blend 3 add : 2 lshl_add", d_out, d_cyc);
    run_op<60>("v_cvt_f64_i32", d_out, d_cyc);
    run_op<71>("v_cvt_f64_u32", d_out, d_cyc);
    run_op<63>("v_cvt_i32_f64", d_out, d_cyc);
    run_op<61>("v_add_f64", d_out, d_cyc);
    run_op<62>("v_mul_f64", d_out, d_cyc);
    run_op<64>("v_rndne_f64", d_out, d_cyc);
    run_op<65>("v_ldexp_f64", d_out, d_cyc);
    run_op<66>("v_cndmask_b32 vcc", d_out, d_cyc);
    run_op<67>("v_cndmask_b32 sgpr pair", d_out, d_cyc);
    run_op<68>("v_cmp_lt_u32 -> vcc", d_out, d_cyc);
    run_op<69>("v_lshlrev_b64", d_out, d_cyc);
    run_op<70>("v_bfe_u32", d_out, d_cyc);
    run_op<72>("1 v_cmp vcc : 1 cndmask vcc", d_out, d_cyc);
    run_op<73>("1 v_cmp vcc : 4 cndmask vcc", d_out, d_cyc);
    run_op<74>("1 v_cmp sgpr : 4 cndmask sgpr", d_out, d_cyc);
    run_op<75>("s_mov vcc : 63 cndmask vcc", d_out, d_cyc);
    run_op<76>("1 v_cmp vcc : 63 cndmask vcc", d_out, d_cyc);
    run_op<77>("1 v_cmp sgpr : 63 cndmask sgp", d_out, d_cyc);
    run_op<78>("v_addc_co_u32 sgpr pair", d_out, d_cyc);
    run_op<79>("add_co vcc : addc vcc", d_out, d_cyc);
    printf("\nstreams that alternate classes (same stream on every wave):\n");
    run_op<43>("1 add : 1 lshl_add", d_out, d_cyc);
    run_op<44>("2 add : 2 lshl_add", d_out, d_cyc);
    run_op<40>("8 add : 8 lshl_add", d_out, d_cyc);
    run_op<41>("32 add : 32 lshl_add", d_out, d_cyc);
    run_op<42>("128 add : 128 lshl_add", d_out, d_cyc);
    run_op<45>("3 add : 1 lshl_add", d_out, d_cyc);
    run_op<46>("7 add : 1 lshl_add", d_out, d_cyc);
    run_op<47>("3 add : 1 mad_u64_u32", d_out, d_cyc);
    run_op<48>("1 add : 1 and literal", d_out, d_cyc);
    run_op<49>("1 add : 1 add sgpr", d_out, d_cyc);
    printf("\nthe same with an s_barrier before every run of adds (the two waves of a SIMD start the run together):\n");
    run_op<50>("barrier, 128 add, 128 lshl_add", d_out, d_cyc);
    run_op<51>("barrier, 64 add, 64 lshl_add", d_out, d_cyc);
    run_op<52>("barrier, 32 add, 32 lshl_add", d_out, d_cyc);
    run_op<53>("barrier, 16 add, 16 lshl_add", d_out, d_cyc);
    run_op<54>("barrier, 192 add, 64 lshl_add", d_out, d_cyc);
    run_op<56>("barrier, 48 add, 16 lshl_add", d_out, d_cyc);
    run_op<55>("barrier, 64 add", d_out, d_cyc);
    printf("\ntwo streams per SIMD (A = waves 0-3, B = waves 4-7), wave-resident cycles per own instruction:\n");
    run_pair<0, 0, 0>("add | add", d_out, d_cyc);
    run_pair<7, 7, 0>("lshl_add | lshl_add", d_out, d_cyc);
    run_pair<0, 7, 0>("add | lshl_add", d_out, d_cyc);
    run_pair<0, 11, 0>("add | mad_u64_u32", d_out, d_cyc);
    run_pair<7, 11, 0>("lshl_add | mad_u64_u32", d_out, d_cyc);
    run_pair<11, 11, 0>("mad_u64_u32 | mad_u64_u32", d_out, d_cyc);
    run_pair<29, 29, 0>("blend | blend", d_out, d_cyc);
    run_pair<29, 29, 1>("blend (prio 3) | blend", d_out, d_cyc);
    run_pair<0, 15, 0>("add | fma_f64", d_out, d_cyc);
    run_pair<15, 15, 0>("fma_f64 | fma_f64", d_out, d_cyc);
    run_pair<7, 15, 0>("lshl_add | fma_f64", d_out, d_cyc);
    run_pair<0, 6, 0>("add | lshlrev", d_out, d_cyc);
    run_pair<0, 4, 0>("add | add sgpr", d_out, d_cyc);
    return 0;
}
