// kernels_xfft.hip -- the exact-FFT engine for NTT-parameter keys (blind_rotate_xfft.h): __global__ wrappers, the key
// image builder and the launchers.  A translation unit of its own: its register budget is tuned separately from the
// frozen kernels of kernels.hip.
#include <hip/hip_runtime.h>

#include "blind_rotate_xfft.h"
#include "kernel_parts.h"

// per-wave LDS: the two exchange buffers (accumulator mirror aliased onto B, bara in the row padding of A)
#define WAVE_BRX_LDS_BYTES (2 * WAVE_FXBUF_BYTES)
#define BRX_BASE_PAD 128
static_assert((FTABLE_LDS_BYTES + BR_PACE_BYTES + BRX_BASE_PAD) % 256 == 0 && WAVE_BRX_LDS_BYTES % 256 == 0,
              "per-wave LDS regions of k_bootstrap_xfft are 256-byte aligned");
static constexpr size_t brx_lds_bytes(int waves) { return FTABLE_LDS_BYTES + BR_PACE_BYTES + BRX_BASE_PAD + (size_t)waves * WAVE_BRX_LDS_BYTES; }
static_assert(brx_lds_bytes(BR_WAVES_PER_BLOCK) <= 160 * 1024, "LDS budget of k_bootstrap_xfft");

__device__ __forceinline__ BrXfftLds carve_brx_lds(int wave)
{
    unsigned char *base = g_smem + FTABLE_LDS_BYTES + BR_PACE_BYTES + BRX_BASE_PAD + wave * WAVE_BRX_LDS_BYTES;
    BrXfftLds lds;
    lds.pace.mine = nullptr;
    lds.pace.other = nullptr;
    lds.xbufA = (cplx *)base;
    lds.xbufB = (cplx *)(base + WAVE_FXBUF_BYTES);
    lds.tw1 = (const cplx *)g_smem;
    lds.tw2 = (const cplx *)g_smem + FFT_TW1_ELEMS;
    lds.park = nullptr;
    return lds;
}

__global__ __launch_bounds__(BR_BLOCK_THREADS, 2) void k_bootstrap_xfft(BrLaunch P)
{
    load_ftables((const cplx *)P.tw_a, (const cplx *)P.tw_b);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long gbit = (long)blockIdx.x * (blockDim.x >> 6) + wave;   // waves per group chosen at launch; wave-uniform
    const BrPace pace = carve_pace(g_smem + FTABLE_LDS_BYTES, wave, gbit < P.nbits_total);
    if (gbit >= P.nbits_total) return;
    const int job = gbit >= P.bits_per_job ? 1 : 0;
    const long bit = gbit - (job ? P.bits_per_job : 0);

    BrXfftLds lds = carve_brx_lds(wave);
    lds.pace = pace;
    lds.park = P.park + gbit * 2048;
    const FftLane L = fft_lane_init(lane);
    const u32 barb = brf_prologue(P.job[job].s0, P.job[job].s1, P.job[job].c0, bit, P.n, brx_as_fft_lds(lds), lane);
    WAVE_SYNC();
    u32 acc[2][16];
    ClockProbe probe;
    probe.begin(P);
    brx_blind_rotate(acc, (const cplx *)P.bk, P.n, barb, P.mu, lds, L);
    probe.end(P);
    br_extract<1>(P.out_a + gbit * P.out_a_stride, P.out_b + gbit * P.out_b_stride, acc, lane);
}

// step-by-step driver / nufhe_external_mul / nufhe_blind_rotate on the exact engine (cf. k_blind_rotate_accum_fft)
__global__ __launch_bounds__(BR_BLOCK_THREADS, 2) void k_blind_rotate_accum_xfft(
    i32 *__restrict__ accum, const cplx *__restrict__ bk, const i32 *__restrict__ bara, long bara_stride,
    int row0, int n_rows, int external_mul_only, long batch, const cplx *__restrict__ tw1,
    const cplx *__restrict__ tw2, u32 *__restrict__ park)
{
    load_ftables(tw1, tw2);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long bit = (long)blockIdx.x * BR_WAVES_PER_BLOCK + wave;
    if (bit >= batch) return;
    BrXfftLds lds = carve_brx_lds(wave);
    lds.park = park + bit * 2048;
    const FftLane L = fft_lane_init(lane);
    i32 *my = accum + bit * 2048;
    i32 *mirror = (i32 *)lds.xbufB;
    u32 acc[2][16];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            acc[m][r] = (u32)my[m * 1024 + lane + 64 * r];
            mirror[m * 1024 + lane + 64 * r] = (i32)acc[m][r];
        }
    WAVE_SYNC();
    if (external_mul_only) {
        u32 res[2][16];
        brx_external_product(res, acc, bk + (long)row0 * BKX_ROW_ELEMS, lds, lds.tw2, L);
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][r] = res[m][r];
    } else {
        for (int i = 0; i < n_rows; i++) {
            const u32 a = WAVE_UNIFORM((u32)bara[bit * bara_stride + i]) & 2047u;
            if (a == 0) continue;
#if defined(BR_PROBE) && defined(__HIP_DEVICE_COMPILE__)
            BrProbe probe_ = {};
#endif
            brx_step(acc, a, bk + (long)(row0 + i) * BKX_ROW_ELEMS, lds, lds.tw2, L BR_PROBE_PASS);
        }
    }
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) my[m * 1024 + lane + 64 * r] = (i32)acc[m][r];
}

// ---- tlwe_mask_size = 2 (brxk_*): per wave two exchange buffers + the accumulator int32[3][1024], one wave per SIMD ----
#define BRX2_WAVES 4
#define WAVE_BRX2_LDS_BYTES (2 * WAVE_FXBUF_BYTES + 3 * 1024 * 4)
static constexpr size_t brx2_lds_bytes(int waves) { return FTABLE_LDS_BYTES + BR_PACE_BYTES + BRX_BASE_PAD + (size_t)waves * WAVE_BRX2_LDS_BYTES; }

__device__ __forceinline__ BrFftLdsK carve_brx2_lds(int wave)
{
    unsigned char *base = g_smem + FTABLE_LDS_BYTES + BR_PACE_BYTES + BRX_BASE_PAD + wave * WAVE_BRX2_LDS_BYTES;
    BrFftLdsK lds;
    lds.xbufA = (cplx *)base;
    lds.xbufB = (cplx *)(base + WAVE_FXBUF_BYTES);
    lds.acc = (i32 *)(base + 2 * WAVE_FXBUF_BYTES);
    lds.tw1 = (const cplx *)g_smem;
    lds.tw2 = (const cplx *)g_smem + FFT_TW1_ELEMS;
    return lds;
}

__global__ __launch_bounds__(64 * BRX2_WAVES, 1) void k_bootstrap_xfft_k2(BrLaunch P)
{
    load_ftables((const cplx *)P.tw_a, (const cplx *)P.tw_b);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long gbit = (long)blockIdx.x * (blockDim.x >> 6) + wave;
    if (gbit >= P.nbits_total) return;
    const int job = gbit >= P.bits_per_job ? 1 : 0;
    const long bit = gbit - (job ? P.bits_per_job : 0);
    const BrFftLdsK lds = carve_brx2_lds(wave);
    const FftLane L = fft_lane_init(lane);
    u32 acc[3][16];
    brxk_bootstrap_body<2>(acc, P.job[job].s0, P.job[job].s1, P.job[job].c0, bit, (const cplx *)P.bk, P.n, P.mu, lds, L);
    br_extract<2>(P.out_a + gbit * P.out_a_stride, P.out_b + gbit * P.out_b_stride, acc, lane);
}

__global__ __launch_bounds__(64 * BRX2_WAVES, 1) void k_blind_rotate_accum_xfft_k2(
    i32 *__restrict__ accum, const cplx *__restrict__ bk, const i32 *__restrict__ bara, long bara_stride,
    int row0, int n_rows, int external_mul_only, long batch, const cplx *__restrict__ tw1,
    const cplx *__restrict__ tw2)
{
    load_ftables(tw1, tw2);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long bit = (long)blockIdx.x * BRX2_WAVES + wave;
    if (bit >= batch) return;
    const BrFftLdsK lds = carve_brx2_lds(wave);
    const FftLane L = fft_lane_init(lane);
    i32 *my = accum + bit * 3 * 1024;
#pragma unroll
    for (int m = 0; m < 3; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) lds.acc[m * 1024 + lane + 64 * r] = my[m * 1024 + lane + 64 * r];
    WAVE_SYNC();
    constexpr long kRow = (long)BK_ROW_POLYS(2) * BKX_POLY_ELEMS;
    if (external_mul_only) {
        brxk_external_product<2>(
            [&](int m, u32(&T)[16]) {
#pragma unroll
                for (int r = 0; r < 16; r++) T[r] = (u32)lds.acc[m * 1024 + lane + 64 * r];
            },
            [&](int mo, int r, u32 v) { lds.acc[mo * 1024 + lane + 64 * r] = (i32)v; }, bk + row0 * kRow, lds, L);
    } else {
        for (int i = 0; i < n_rows; i++) {
            const u32 a = WAVE_UNIFORM((u32)bara[bit * bara_stride + i]) & 2047u;
            if (a == 0) continue;
            brxk_step<2>(a, bk + (row0 + i) * kRow, lds, L);
        }
    }
    WAVE_SYNC();
#pragma unroll
    for (int m = 0; m < 3; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) my[m * 1024 + lane + 64 * r] = lds.acc[m * 1024 + lane + 64 * r];
}

// ---- small batches, k = 1: four waves per bit (brxq_*), up to BRXQ_MAX_TEAMS teams per work-group ----
// LDS: tables | arrival words of the team barriers (in the pacing block) | pad | per team: ACC 8 KiB, bara 1 KiB, 4 exchange buffers
#define BRXQ_MAX_TEAMS 2      /* (registers: 245 per wave; a third team would leave 168) */
#define XQUAD_LDS_BYTES (2 * 1024 * 4 + BR_MAX_LWE * 2 + 4 * WAVE_FXBUF_BYTES)
static_assert(XQUAD_LDS_BYTES % 256 == 0, "per-team LDS regions of k_bootstrap_xfft_quad are 256-byte aligned");
// (a lone team has room for a second exchange buffer per wave: no barrier in front of the inverse transform)
static constexpr size_t brxq_lds_bytes(int teams) { return FTABLE_LDS_BYTES + BR_PACE_BYTES + BRX_BASE_PAD + (size_t)teams * XQUAD_LDS_BYTES + (teams == 1 ? 4 * WAVE_FXBUF_BYTES : 0); }
static_assert(brxq_lds_bytes(BRXQ_MAX_TEAMS) <= 160 * 1024, "LDS budget of k_bootstrap_xfft_quad");
static_assert(4 * BRXQ_MAX_TEAMS * 4 <= BR_PACE_BYTES, "arrival words of the team barriers");

// ONE_TEAM: the work-group IS the team (batches up to 1 x CUs bits, the latency case): s_barrier instead of the arrival words
template <bool ONE_TEAM>
__global__ __launch_bounds__(ONE_TEAM ? 256 : 256 * BRXQ_MAX_TEAMS, 1) void k_bootstrap_xfft_quad(BrLaunch P)
{
    load_ftables((const cplx *)P.tw_a, (const cplx *)P.tw_b);      // (zeroes the arrival words, ends with a work-group barrier)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int team = ONE_TEAM ? 0 : wave >> 2, w = wave & 3;
    const long gbit = ONE_TEAM ? (long)blockIdx.x : (long)blockIdx.x * (blockDim.x >> 8) + team;    // teams per group chosen at launch
    if (gbit >= P.nbits_total) return;                                 // (all four waves of the team)
    const int job = gbit >= P.bits_per_job ? 1 : 0;
    const long bit = gbit - (job ? P.bits_per_job : 0);
    unsigned char *base = g_smem + FTABLE_LDS_BYTES + BR_PACE_BYTES + BRX_BASE_PAD + team * XQUAD_LDS_BYTES;
    BrXfftQuadLds lds;
    lds.acc = (i32 *)base;
    lds.bara = (uint16_t *)(base + 2 * 1024 * 4);
    lds.xbuf_team = (const cplx *)(base + 2 * 1024 * 4 + BR_MAX_LWE * 2);
    lds.xbuf = (cplx *)lds.xbuf_team + w * FFT_XBUF_ELEMS;
    lds.xbuf_inv = ONE_TEAM ? lds.xbuf + 4 * FFT_XBUF_ELEMS : lds.xbuf;
    lds.tw1 = (const cplx *)g_smem;
    lds.tw2 = (const cplx *)g_smem + FFT_TW1_ELEMS;
    // team barrier: one arrival counter per wave; a wave publishes its count and waits until all four have reached it
    typedef __attribute__((address_space(3))) volatile u32 lds_vu32;
    lds_vu32 *arrive = (lds_vu32 *)((u32 *)(g_smem + FTABLE_LDS_BYTES) + 4 * team);
    u32 seq = 0;
    auto team_sync = [&] {
        if (ONE_TEAM) {
            __syncthreads();
            return;
        }
        seq++;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        arrive[w] = seq;
        for (;;) {
            // the team's four counters in one 16-byte read, no sleep between polls (4 separate reads + s_sleep 1: +1.5 %)
            typedef u32 brx_u32x4 __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) volatile brx_u32x4 lds_vu32x4;
            const brx_u32x4 v = *(lds_vu32x4 *)arrive;
            const i32 d0 = (i32)(v.x - seq), d1 = (i32)(v.y - seq), d2 = (i32)(v.z - seq), d3 = (i32)(v.w - seq);
            if (__builtin_amdgcn_readfirstlane((d0 | d1 | d2 | d3)) >= 0) break;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    const FftLane L = fft_lane_init(lane);
    i32 *out_a = P.out_a + gbit * P.out_a_stride, *out_b = P.out_b + gbit * P.out_b_stride;
    const BrSource &s0 = P.job[job].s0, &s1 = P.job[job].s1;
    const i32 c0 = P.job[job].c0;
    switch (w) {
    case 0: brxq_bootstrap<0, 1, ONE_TEAM>(out_a, out_b, s0, s1, c0, bit, (const cplx *)P.bk, P.n, P.mu, lds, L, team_sync); break;
    case 1: brxq_bootstrap<1, 1, ONE_TEAM>(out_a, out_b, s0, s1, c0, bit, (const cplx *)P.bk, P.n, P.mu, lds, L, team_sync); break;
    case 2: brxq_bootstrap<2, 1, ONE_TEAM>(out_a, out_b, s0, s1, c0, bit, (const cplx *)P.bk, P.n, P.mu, lds, L, team_sync); break;
    default: brxq_bootstrap<3, 1, ONE_TEAM>(out_a, out_b, s0, s1, c0, bit, (const cplx *)P.bk, P.n, P.mu, lds, L, team_sync); break;
    }
}

// tlwe_mask_size = 2, small batches: six waves per bit, the work-group is the team (batches up to 1 x CUs bits).
// LDS: tables | pad | ACC 12 KiB, bara 1 KiB, 6 + 6 exchange buffers
static constexpr size_t brxq2_lds_bytes() { return FTABLE_LDS_BYTES + BR_PACE_BYTES + BRX_BASE_PAD + 3 * 1024 * 4 + BR_MAX_LWE * 2 + 12 * WAVE_FXBUF_BYTES; }
static_assert(brxq2_lds_bytes() <= 160 * 1024, "LDS budget of k_bootstrap_xfft_hex_k2");
__global__ __launch_bounds__(384, 1) void k_bootstrap_xfft_hex_k2(BrLaunch P)
{
    load_ftables((const cplx *)P.tw_a, (const cplx *)P.tw_b);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long gbit = (long)blockIdx.x;
    const int job = gbit >= P.bits_per_job ? 1 : 0;
    const long bit = gbit - (job ? P.bits_per_job : 0);
    unsigned char *base = g_smem + FTABLE_LDS_BYTES + BR_PACE_BYTES + BRX_BASE_PAD;
    BrXfftQuadLds lds;
    lds.acc = (i32 *)base;
    lds.bara = (uint16_t *)(base + 3 * 1024 * 4);
    lds.xbuf_team = (const cplx *)(base + 3 * 1024 * 4 + BR_MAX_LWE * 2);
    lds.xbuf = (cplx *)lds.xbuf_team + w * FFT_XBUF_ELEMS;
    lds.xbuf_inv = lds.xbuf + 6 * FFT_XBUF_ELEMS;
    lds.tw1 = (const cplx *)g_smem;
    lds.tw2 = (const cplx *)g_smem + FFT_TW1_ELEMS;
    auto team_sync = [] { __syncthreads(); };
    const FftLane L = fft_lane_init(lane);
    i32 *out_a = P.out_a + gbit * P.out_a_stride, *out_b = P.out_b + gbit * P.out_b_stride;
    const BrSource &s0 = P.job[job].s0, &s1 = P.job[job].s1;
    const i32 c0 = P.job[job].c0;
    switch (w) {
    case 0: brxq_bootstrap<0, 2, true>(out_a, out_b, s0, s1, c0, bit, (const cplx *)P.bk, P.n, P.mu, lds, L, team_sync); break;
    case 1: brxq_bootstrap<1, 2, true>(out_a, out_b, s0, s1, c0, bit, (const cplx *)P.bk, P.n, P.mu, lds, L, team_sync); break;
    case 2: brxq_bootstrap<2, 2, true>(out_a, out_b, s0, s1, c0, bit, (const cplx *)P.bk, P.n, P.mu, lds, L, team_sync); break;
    case 3: brxq_bootstrap<3, 2, true>(out_a, out_b, s0, s1, c0, bit, (const cplx *)P.bk, P.n, P.mu, lds, L, team_sync); break;
    case 4: brxq_bootstrap<4, 2, true>(out_a, out_b, s0, s1, c0, bit, (const cplx *)P.bk, P.n, P.mu, lds, L, team_sync); break;
    default: brxq_bootstrap<5, 2, true>(out_a, out_b, s0, s1, c0, bit, (const cplx *)P.bk, P.n, P.mu, lds, L, team_sync); break;
    }
}

// ---- FFT keys, batches up to 1 x CUs bits: four waves per bit (blind_rotate_fft.h, brfq_*), the work-group is the team.
// (In this unit beside its exact-engine twin: kernels.hip holds the frozen kernels.)  LDS as brxq_lds_bytes(1).
__global__ __launch_bounds__(256, 1) void k_bootstrap_fft_quad(BrLaunch P)
{
    load_ftables((const cplx *)P.tw_a, (const cplx *)P.tw_b);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long gbit = (long)blockIdx.x;
    const int job = gbit >= P.bits_per_job ? 1 : 0;
    const long bit = gbit - (job ? P.bits_per_job : 0);
    unsigned char *base = g_smem + FTABLE_LDS_BYTES + BR_PACE_BYTES + BRX_BASE_PAD;
    BrFftQuadLds lds;
    lds.acc = (i32 *)base;
    lds.bara = (uint16_t *)(base + 2 * 1024 * 4);
    lds.xbuf_team = (const cplx *)(base + 2 * 1024 * 4 + BR_MAX_LWE * 2);
    lds.xbuf = (cplx *)lds.xbuf_team + w * FFT_XBUF_ELEMS;
    lds.xbuf_inv = lds.xbuf + 4 * FFT_XBUF_ELEMS;
    lds.partner_inv = lds.xbuf_team + (4 + (w ^ 1)) * FFT_XBUF_ELEMS;
    lds.tw1 = (const cplx *)g_smem;
    lds.tw2 = (const cplx *)g_smem + FFT_TW1_ELEMS;
    auto team_sync = [] { __syncthreads(); };
    const FftLane L = fft_lane_init(lane);
    i32 *out_a = P.out_a + gbit * P.out_a_stride, *out_b = P.out_b + gbit * P.out_b_stride;
    const BrSource &s0 = P.job[job].s0, &s1 = P.job[job].s1;
    const i32 c0 = P.job[job].c0;
    switch (w) {
    case 0: brfq_bootstrap<0, 1>(out_a, out_b, s0, s1, c0, bit, (const cplx *)P.bk, P.n, P.mu, lds, L, team_sync); break;
    case 1: brfq_bootstrap<1, 1>(out_a, out_b, s0, s1, c0, bit, (const cplx *)P.bk, P.n, P.mu, lds, L, team_sync); break;
    case 2: brfq_bootstrap<2, 1>(out_a, out_b, s0, s1, c0, bit, (const cplx *)P.bk, P.n, P.mu, lds, L, team_sync); break;
    default: brfq_bootstrap<3, 1>(out_a, out_b, s0, s1, c0, bit, (const cplx *)P.bk, P.n, P.mu, lds, L, team_sync); break;
    }
}

hipError_t launch_bootstrap_fft_quad(const BrLaunch &P, hipStream_t stream)
{
    if (P.nbits_total == 0) return hipSuccess;
    hipLaunchKernelGGL(k_bootstrap_fft_quad, dim3((unsigned)P.nbits_total), dim3(256), brxq_lds_bytes(1), stream, P);
    return hipGetLastError();
}

// tlwe_mask_size = 2, FFT keys, batches up to 1 x CUs bits: six waves per bit (brfq_* with K = 2); LDS as brxq2_lds_bytes()
__global__ __launch_bounds__(384, 1) void k_bootstrap_fft_hex_k2(BrLaunch P)
{
    load_ftables((const cplx *)P.tw_a, (const cplx *)P.tw_b);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long gbit = (long)blockIdx.x;
    const int job = gbit >= P.bits_per_job ? 1 : 0;
    const long bit = gbit - (job ? P.bits_per_job : 0);
    unsigned char *base = g_smem + FTABLE_LDS_BYTES + BR_PACE_BYTES + BRX_BASE_PAD;
    BrFftQuadLds lds;
    lds.acc = (i32 *)base;
    lds.bara = (uint16_t *)(base + 3 * 1024 * 4);
    lds.xbuf_team = (const cplx *)(base + 3 * 1024 * 4 + BR_MAX_LWE * 2);
    lds.xbuf = (cplx *)lds.xbuf_team + w * FFT_XBUF_ELEMS;
    lds.xbuf_inv = lds.xbuf + 6 * FFT_XBUF_ELEMS;
    lds.partner_inv = lds.xbuf_team + (6 + (w ^ 1)) * FFT_XBUF_ELEMS;
    lds.tw1 = (const cplx *)g_smem;
    lds.tw2 = (const cplx *)g_smem + FFT_TW1_ELEMS;
    auto team_sync = [] { __syncthreads(); };
    const FftLane L = fft_lane_init(lane);
    i32 *out_a = P.out_a + gbit * P.out_a_stride, *out_b = P.out_b + gbit * P.out_b_stride;
    const BrSource &s0 = P.job[job].s0, &s1 = P.job[job].s1;
    const i32 c0 = P.job[job].c0;
    switch (w) {
    case 0: brfq_bootstrap<0, 2>(out_a, out_b, s0, s1, c0, bit, (const cplx *)P.bk, P.n, P.mu, lds, L, team_sync); break;
    case 1: brfq_bootstrap<1, 2>(out_a, out_b, s0, s1, c0, bit, (const cplx *)P.bk, P.n, P.mu, lds, L, team_sync); break;
    case 2: brfq_bootstrap<2, 2>(out_a, out_b, s0, s1, c0, bit, (const cplx *)P.bk, P.n, P.mu, lds, L, team_sync); break;
    case 3: brfq_bootstrap<3, 2>(out_a, out_b, s0, s1, c0, bit, (const cplx *)P.bk, P.n, P.mu, lds, L, team_sync); break;
    case 4: brfq_bootstrap<4, 2>(out_a, out_b, s0, s1, c0, bit, (const cplx *)P.bk, P.n, P.mu, lds, L, team_sync); break;
    default: brfq_bootstrap<5, 2>(out_a, out_b, s0, s1, c0, bit, (const cplx *)P.bk, P.n, P.mu, lds, L, team_sync); break;
    }
}

hipError_t launch_bootstrap_fft_hex_k2(const BrLaunch &P, hipStream_t stream)
{
    if (P.nbits_total == 0) return hipSuccess;
    hipLaunchKernelGGL(k_bootstrap_fft_hex_k2, dim3((unsigned)P.nbits_total), dim3(384), brxq2_lds_bytes(), stream, P);
    return hipGetLastError();
}

// Split key image: one wave per (polynomial, half).  K = lo + 2^16 hi (xfft_split), each half through the forward
// transform, stored in the wave layout of the product phase: [poly][half][reg][lane].
#define BKX_WAVES_PER_BLOCK 4
__global__ __launch_bounds__(64 * BKX_WAVES_PER_BLOCK) void k_bkx_from_coeffs(cplx *__restrict__ out, const i32 *__restrict__ in,
                                                                               long polys, const cplx *__restrict__ tw1,
                                                                               const cplx *__restrict__ tw2)
{
    load_ftables(tw1, tw2);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long ph = (long)blockIdx.x * BKX_WAVES_PER_BLOCK + wave;
    if (ph >= 2 * polys) return;
    const long p = ph >> 1;
    const int h = (int)(ph & 1);
    const FftLane L = fft_lane_init(lane);
    cplx x[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        i32 lo0, hi0, lo1, hi1;
        xfft_split(in[p * 1024 + lane + 64 * r], lo0, hi0);
        xfft_split(in[p * 1024 + lane + 64 * r + 512], lo1, hi1);
        x[r] = h == 0 ? cplx{(double)lo0, -(double)lo1} : cplx{(double)hi0, -(double)hi1};
    }
    cplx *xbuf = (cplx *)(g_smem + FTABLE_LDS_BYTES + BR_PACE_BYTES + BRX_BASE_PAD + wave * WAVE_FXBUF_BYTES);
    fft_forward(x, xbuf, (const cplx *)g_smem, (const cplx *)g_smem + FFT_TW1_ELEMS, L);
#pragma unroll
    for (int r = 0; r < 8; r++) out[ph * BKF_POLY_ELEMS + bkf_elem_offset(lane, r)] = x[r];
}

hipError_t xfft_init()
{
    hipError_t e = hipFuncSetAttribute((const void *)k_bootstrap_xfft, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)brx_lds_bytes(BR_WAVES_PER_BLOCK));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)k_blind_rotate_accum_xfft, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)brx_lds_bytes(BR_WAVES_PER_BLOCK));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)k_bootstrap_xfft_k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)brx2_lds_bytes(BRX2_WAVES));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)k_bootstrap_xfft_quad<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)brxq_lds_bytes(BRXQ_MAX_TEAMS));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)k_bootstrap_xfft_quad<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)brxq_lds_bytes(1));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)k_bootstrap_fft_quad, hipFuncAttributeMaxDynamicSharedMemorySize, (int)brxq_lds_bytes(1));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)k_bootstrap_fft_hex_k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)brxq2_lds_bytes());
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)k_bootstrap_xfft_hex_k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)brxq2_lds_bytes());
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void *)k_blind_rotate_accum_xfft_k2, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)brx2_lds_bytes(BRX2_WAVES));
}

hipError_t launch_bootstrap_xfft(const BrLaunch &P, int mask_size, int num_cus, long quad_max_bits, hipStream_t stream)
{
    if (P.nbits_total == 0) return hipSuccess;
    if (mask_size == 1 && P.nbits_total <= quad_max_bits) {
        // small batch: 4 waves per bit; as few teams per work-group as still give one round
        long teams = (P.nbits_total + num_cus - 1) / num_cus;
        if (teams > BRXQ_MAX_TEAMS) teams = BRXQ_MAX_TEAMS;
        if (teams == 1)
            hipLaunchKernelGGL(k_bootstrap_xfft_quad<true>, dim3((unsigned)P.nbits_total), dim3(256), brxq_lds_bytes(1), stream, P);
        else
            hipLaunchKernelGGL(k_bootstrap_xfft_quad<false>, dim3(blocks_for(P.nbits_total, (int)teams)), dim3(256 * (unsigned)teams),
                               brxq_lds_bytes((int)teams), stream, P);
        return hipGetLastError();
    }
    // k = 2: six waves per bit, one bit per CU at a time (rounds of 1 x CUs bits in 3.45 ms), or one wave per bit (rounds of
    // 4 x CUs bits in 12.9 ms): whichever needs less time for this batch (measured: 3.06 ms per further round, ratio 0.24)
    const long hex_rounds = (P.nbits_total + num_cus - 1) / num_cus, wave_rounds = (P.nbits_total + 4L * num_cus - 1) / (4L * num_cus);
    if (mask_size == 2 && quad_max_bits > 0 && 240 * hex_rounds < 1000 * wave_rounds) {
        hipLaunchKernelGGL(k_bootstrap_xfft_hex_k2, dim3((unsigned)P.nbits_total), dim3(384), brxq2_lds_bytes(), stream, P);
        return hipGetLastError();
    }
    if (mask_size == 2) {
        const int w = br_pick_waves(P.nbits_total, BRX2_WAVES, num_cus);
        hipLaunchKernelGGL(k_bootstrap_xfft_k2, dim3(blocks_for(P.nbits_total, w)), dim3(64 * w), brx2_lds_bytes(w), stream, P);
        return hipGetLastError();
    }
    const int w = br_pick_waves(P.nbits_total, BR_WAVES_PER_BLOCK, num_cus);
    hipLaunchKernelGGL(k_bootstrap_xfft, dim3(blocks_for(P.nbits_total, w)), dim3(64 * w), brx_lds_bytes(w), stream, P);
    return hipGetLastError();
}

hipError_t launch_blind_rotate_accum_xfft(i32 *accum, const cplx *bkx, const i32 *bara, long bara_stride, int row0, int n_rows,
                                          int external_mul_only, long batch, const cplx *tw1, const cplx *tw2, u32 *park,
                                          int mask_size, hipStream_t stream)
{
    if (batch == 0) return hipSuccess;
    if (mask_size == 2) {
        hipLaunchKernelGGL(k_blind_rotate_accum_xfft_k2, dim3(blocks_for(batch, BRX2_WAVES)), dim3(64 * BRX2_WAVES),
                           brx2_lds_bytes(BRX2_WAVES), stream, accum, bkx, bara, bara_stride, row0, n_rows, external_mul_only,
                           batch, tw1, tw2);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_blind_rotate_accum_xfft, dim3(blocks_for(batch, BR_WAVES_PER_BLOCK)), dim3(BR_BLOCK_THREADS),
                       brx_lds_bytes(BR_WAVES_PER_BLOCK), stream, accum, bkx, bara, bara_stride, row0, n_rows, external_mul_only,
                       batch, tw1, tw2, park);
    return hipGetLastError();
}

hipError_t launch_bkx_from_coeffs(cplx *out, const i32 *in, long polys, const cplx *tw1, const cplx *tw2, hipStream_t stream)
{
    if (polys == 0) return hipSuccess;
    hipLaunchKernelGGL(k_bkx_from_coeffs, dim3(blocks_for(2 * polys, BKX_WAVES_PER_BLOCK)), dim3(64 * BKX_WAVES_PER_BLOCK),
                       FTABLE_LDS_BYTES + BR_PACE_BYTES + BRX_BASE_PAD + (size_t)BKX_WAVES_PER_BLOCK * WAVE_FXBUF_BYTES, stream,
                       out, in, polys, tw1, tw2);
    return hipGetLastError();
}

#if defined(BR_PROBE)
// variant builds only (blind_rotate.h, BR_PROBE): read and clear the phase tick counters of this unit's kernels
extern "C" int nufhe_probe_read_xfft(unsigned long long *out16)
{
    unsigned long long zero[16] = {0};
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_br_probe), sizeof(zero)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_br_probe), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif
