// kernels.hip -- gfx950 kernels of the bootstrapped-gate hot path and their host launchers.
// Device-side bodies live in ff.h / ntt1024.h / blind_rotate.h / keyswitch.h; this file adds the
// __global__ wrappers (LDS carving, wave -> bit mapping) and the launch geometry.
#include <hip/hip_runtime.h>
#include <string.h>

#define BR_KEY_BUFFER_LOADS 1   /* key words through buffer loads (blind_rotate.h, br_key_stream): every row pointer in this unit is wave-uniform */
#include "blind_rotate.h"
#include "ff.h"
#include "kernels.h"
#include "ntt1024.h"
#include "l4_hook.h"
#include "kernel_parts.h"

// ------------------------------------------------------------------------------------------
// LDS carving
// ------------------------------------------------------------------------------------------
// [0, 8 KiB)   tw1f (tw1x for the blind-rotation kernels)     [8 KiB, 16 KiB)  tw1i      then one region per wave:
//   xbuf (8704 B) | acc mirror (8192 B) | bara (1024 B)            = 17920 B
#define TABLE_LDS_BYTES (2 * 1024 * 8)
#define WAVE_XBUF_BYTES (NTT_XBUF_ELEMS * 8)
#define WAVE_ACC_BYTES_K(K) (((K) + 1) * 1024 * 4)
#define WAVE_ACC_BYTES WAVE_ACC_BYTES_K(1)
#define WAVE_BARA_BYTES (BR_MAX_LWE * 2)
#define WAVE_BR_LDS_BYTES_K(K) (WAVE_XBUF_BYTES + WAVE_ACC_BYTES_K(K) + WAVE_BARA_BYTES)
// waves (= bits) per work-group, one work-group per CU: mask size 1 runs 2 waves/SIMD (<= 256 VGPRs,
// ~19 KiB of LDS each); mask size 2 has a 12 KiB accumulator mirror per wave, 6 waves fill the LDS
#define BR_WAVES_K(K) ((K) == 1 ? 8 : 6)

__device__ __forceinline__ void load_tables(const u64 *__restrict__ g_tw1f, const u64 *__restrict__ g_tw1i)
{
    u64 *t = (u64 *)g_smem;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) {
        t[i] = g_tw1f[i];
        t[1024 + i] = g_tw1i[i];
    }
    if (threadIdx.x < BR_PACE_BYTES / 4) ((u32 *)(g_smem + TABLE_LDS_BYTES))[threadIdx.x] = 0;
    __syncthreads();
}

template <int K>
__device__ __forceinline__ BrLds carve_br_lds(int wave)
{
    unsigned char *base = g_smem + TABLE_LDS_BYTES + BR_PACE_BYTES + wave * WAVE_BR_LDS_BYTES_K(K);
    BrLds lds;
    lds.pace.mine = nullptr;
    lds.pace.other = nullptr;
    lds.xbuf = (u64 *)base;
    lds.acc = (i32 *)(base + WAVE_XBUF_BYTES);
    lds.bara = (uint16_t *)(base + WAVE_XBUF_BYTES + WAVE_ACC_BYTES_K(K));
    lds.tw1x = (const u64 *)g_smem;
    lds.tw1i = (const u64 *)g_smem + 1024;
    return lds;
}

// ------------------------------------------------------------------------------------------
// K1: fused bootstrap (prologue + blind rotate + extract), one wave per bit
// ------------------------------------------------------------------------------------------
template <int K>
__device__ __forceinline__ void bootstrap_wave_body(const BrLaunch &P)
{
    load_tables((const u64 *)P.tw_a, (const u64 *)P.tw_b);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long gbit = (long)blockIdx.x * (blockDim.x >> 6) + wave;   // waves per group chosen at launch
    const BrPace pace = carve_pace(g_smem + TABLE_LDS_BYTES, wave, gbit < P.nbits_total);
    if (gbit >= P.nbits_total) return;
    const int job = gbit >= P.bits_per_job ? 1 : 0;
    const long bit = gbit - (job ? P.bits_per_job : 0);

    BrLds lds = carve_br_lds<K>(wave);
    lds.pace = pace;
    const NttLane L = ntt_lane_init(lane);
    const u32 barb = br_prologue(P.job[job].s0, P.job[job].s1, P.job[job].c0, bit, P.n, lds, lane);
    WAVE_SYNC();
    u32 acc[K + 1][16];
    ClockProbe probe;
    probe.begin(P);
    br_blind_rotate<K>(acc, (const u64 *)P.bk, P.n, barb, P.mu, lds, L);
    probe.end(P);
    br_extract<K>(P.out_a + gbit * P.out_a_stride, P.out_b + gbit * P.out_b_stride, acc, lane);
}

template <int K>
__global__ __launch_bounds__(64 * BR_WAVES_K(K), 2) void k_bootstrap(BrLaunch P)
{
    bootstrap_wave_body<K>(P);
}

// k = 2 with ONE wave per SIMD (4 per CU, up to 512 registers): the 3 x 16 field-element sums, two transforms' worth
// of limbs and the pipelined key loads fit without the 360 bytes of scratch the 2-waves-per-SIMD build spills, and a
// round of 4 x CUs bits takes 27 ms against 42 ms for 6 x CUs; launch_bootstrap picks whichever needs less time for
// the batch.
#define BR_K2_ROOMY_WAVES 4
__global__ __launch_bounds__(64 * BR_K2_ROOMY_WAVES, 1) void k_bootstrap_k2_roomy(BrLaunch P)
{
    bootstrap_wave_body<2>(P);
}

// Smallest batches (k = 1): the half-ring team, EIGHT waves per bit -- k_bootstrap_team8 lives in kernels_team8.hip, a
// translation unit of its own because it is the one kernel that gains from the back end's max-ilp scheduling strategy
// (its reducer waves run alone on their SIMDs; NOTES.md round 3: -3 %, every other kernel loses with that flag).
hipError_t team8_init();
hipError_t launch_team8(const BrLaunch &P, hipStream_t stream);

// The lone-wave team kernels k_bootstrap_team (k = 1, 4 waves per bit) and k_bootstrap_team_k2 (k = 2, 3 waves per bit) live
// in kernels_team.hip: built with the carry-free form of the 64 x 64 product (ff.h, FF_MULWIDE_PLAIN) -- one wave per
// SIMD is bound by dependent-instruction latency, where the shorter chain wins (round 4: k = 2 team 9.90 -> 9.62 ms).
hipError_t team_init();
hipError_t launch_team(const BrLaunch &P, hipStream_t stream);
hipError_t launch_team_k2(const BrLaunch &P, hipStream_t stream);

// Test hook / multi-kernel-style entry: blind rotate (or a single external product) on
// accumulators held in global memory: accum int32 [batch][2][1024]
// Medium batches (2 x CUs < bits <= 4 x CUs): two waves per bit, up to 4 pairs per work-group, see blind_rotate.h
// (brp_*).  LDS: tables | progress / barrier words | per pair: ACC 8 KiB, bara 1 KiB, 2 exchange buffers
#define PAIR_LDS_BYTES (2 * 1024 * 4 + WAVE_BARA_BYTES + 2 * WAVE_XBUF_BYTES)
#define BRP_MAX_PAIRS 4
__global__ __launch_bounds__(128 * BRP_MAX_PAIRS, 2) void k_bootstrap_pair(BrLaunch P)
{
    load_tables((const u64 *)P.tw_a, (const u64 *)P.tw_b);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int pair = wave >> 1;
    const long gbit = (long)blockIdx.x * (blockDim.x >> 7) + pair;   // pairs per group chosen at launch
    const BrPace pace = carve_pace(g_smem + TABLE_LDS_BYTES, wave, gbit < P.nbits_total);
    if (gbit >= P.nbits_total) return;                                // (both waves of the pair)
    const int job = gbit >= P.bits_per_job ? 1 : 0;
    const long bit = gbit - (job ? P.bits_per_job : 0);
    unsigned char *base = g_smem + TABLE_LDS_BYTES + BR_PACE_BYTES + pair * PAIR_LDS_BYTES;
    BrPairLds lds;
    lds.acc = (i32 *)base;
    lds.bara = (uint16_t *)(base + 2 * 1024 * 4);
    lds.xbuf = (u64 *)(base + 2 * 1024 * 4 + WAVE_BARA_BYTES + (wave & 1) * WAVE_XBUF_BYTES);
    lds.xbuf_other = (const u64 *)(base + 2 * 1024 * 4 + WAVE_BARA_BYTES + ((wave & 1) ^ 1) * WAVE_XBUF_BYTES);
    lds.tw1x = (const u64 *)g_smem;
    lds.tw1i = (const u64 *)g_smem + 1024;
    lds.pace = pace;
    // pair barrier: arrival counters of the two waves behind the 8 progress words (zeroed by load_tables).  The
    // release store orders this wave's earlier LDS traffic before its counter, the acquire load the partner's
    // counter before this wave's later reads.
    u32 *arrive = (u32 *)(g_smem + TABLE_LDS_BYTES) + 8;
    u32 *mine = arrive + wave;
    const u32 *other = arrive + (wave ^ 1);
    u32 seq = 0;
    auto pair_sync = [&] {
        seq++;
        __hip_atomic_store(mine, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        while ((i32)((u32)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(other, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) - seq) < 0)
            __builtin_amdgcn_s_sleep(1);
    };
    const NttLane L = ntt_lane_init(lane);
    i32 *out_a = P.out_a + gbit * P.out_a_stride, *out_b = P.out_b + gbit * P.out_b_stride;
    if ((wave & 1) == 0)
        brp_bootstrap<0>(out_a, out_b, P.job[job].s0, P.job[job].s1, P.job[job].c0, bit, (const u64 *)P.bk, P.n, P.mu,
                         lds, L, pair_sync);
    else
        brp_bootstrap<1>(out_a, out_b, P.job[job].s0, P.job[job].s1, P.job[job].c0, bit, (const u64 *)P.bk, P.n, P.mu,
                         lds, L, pair_sync);
}

// k = 2 without the partial-sum buffer: 3 waves per bit handing their partial sums round through the exchange buffers
// (blind_rotate.h, brr_*), up to 2 teams per work-group.  LDS: tables | pacing / barrier words | per team: ACC 12 KiB,
// bara 1 KiB, 3 exchange buffers
#define RING2_LDS_BYTES (3 * 1024 * 4 + WAVE_BARA_BYTES + 3 * WAVE_XBUF_BYTES)
#define BRR_MAX_TEAMS 2
__global__ __launch_bounds__(192 * BRR_MAX_TEAMS, 2) void k_bootstrap_ring_k2(BrLaunch P)
{
    load_tables((const u64 *)P.tw_a, (const u64 *)P.tw_b);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int team = wave / 3, w = wave - 3 * team;
    const long gbit = (long)blockIdx.x * (blockDim.x / 192) + team;   // teams per group chosen at launch
    const BrPace pace = carve_pace(g_smem + TABLE_LDS_BYTES, wave, gbit < P.nbits_total);
    if (gbit >= P.nbits_total) return;                                 // (all three waves of the team)
    const int job = gbit >= P.bits_per_job ? 1 : 0;
    const long bit = gbit - (job ? P.bits_per_job : 0);
    unsigned char *base = g_smem + TABLE_LDS_BYTES + BR_PACE_BYTES + team * RING2_LDS_BYTES;
    BrRingLds lds;
    lds.acc = (i32 *)base;
    lds.bara = (uint16_t *)(base + 3 * 1024 * 4);
    lds.xbuf_team = (u64 *)(base + 3 * 1024 * 4 + WAVE_BARA_BYTES);
    lds.tw1x = (const u64 *)g_smem;
    lds.tw1i = (const u64 *)g_smem + 1024;
    lds.pace = pace;
    // team barrier over 3 waves: arrival counters in LDS (words 8..13 of the pacing block, zeroed by load_tables)
    u32 *arrive = (u32 *)(g_smem + TABLE_LDS_BYTES) + 8 + 3 * team;
    u32 seq = 0;
    auto team_sync = [&] {
        seq++;
        __hip_atomic_store(arrive + w, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        for (int o = 0; o < 3; o++)
            while ((i32)((u32)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(arrive + o, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) - seq) < 0)
                __builtin_amdgcn_s_sleep(1);
    };
    const NttLane L = ntt_lane_init(lane);
    brr_bootstrap<2>(P.out_a + gbit * P.out_a_stride, P.out_b + gbit * P.out_b_stride, P.job[job].s0, P.job[job].s1,
                     P.job[job].c0, bit, (const u64 *)P.bk, P.n, P.mu, lds, L, w, team_sync);
}

template <int K>
__global__ __launch_bounds__(64 * BR_WAVES_K(K), K == 1 ? 2 : 1) void k_blind_rotate_accum(
    i32 *__restrict__ accum, const u64 *__restrict__ bk, const i32 *__restrict__ bara, long bara_stride,
    int row0, int n_rows, int external_mul_only, long batch, const u64 *__restrict__ tw1f,
    const u64 *__restrict__ tw1i)
{
    load_tables(tw1f, tw1i);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long bit = (long)blockIdx.x * BR_WAVES_K(K) + wave;
    if (bit >= batch) return;
    const BrLds lds = carve_br_lds<K>(wave);
    const NttLane L = ntt_lane_init(lane);
    i32 *my = accum + bit * (K + 1) * 1024;
    u32 acc[K + 1][16];
#pragma unroll
    for (int m = 0; m <= K; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            acc[m][r] = (u32)my[m * 1024 + lane + 64 * r];
            lds.acc[m * 1024 + lane + 64 * r] = (i32)acc[m][r];
        }
    WAVE_SYNC();
    // (the accumulator is read from / updated in the LDS mirror, see blind_rotate.h)
#if defined(BR_PROBE) && defined(__HIP_DEVICE_COMPILE__)
    BrProbe probe_ = {};
#endif
    if (external_mul_only) {
        br_external_product<K>(
            [&](int m, u32(&T)[16]) {
#pragma unroll
                for (int r = 0; r < 16; r++) T[r] = (u32)lds.acc[m * 1024 + lane + 64 * r];
            },
            [&](int mo, int r, u32 v) { lds.acc[mo * 1024 + lane + 64 * r] = (i32)v; },
            bk + (long)row0 * BK_ROW_ELEMS_K(K), lds, L BR_PROBE_PASS);
    } else {
        for (int i = 0; i < n_rows; i++) {
            const u32 a = WAVE_UNIFORM((u32)bara[bit * bara_stride + i]) & 2047u;
            if (a == 0) continue;
            br_step<K>(a, bk + (long)(row0 + i) * BK_ROW_ELEMS_K(K), lds, L BR_PROBE_PASS);
        }
    }
    WAVE_SYNC();
#pragma unroll
    for (int m = 0; m <= K; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[m][r] = (u32)lds.acc[m * 1024 + lane + 64 * r];
#pragma unroll
    for (int m = 0; m <= K; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) my[m * 1024 + lane + 64 * r] = (i32)acc[m][r];
}

// ------------------------------------------------------------------------------------------
// FFT variants (BASELINE config 5).  LDS: tw1 (8 KiB) | tw2 (1 KiB) | per wave: xbuf 9216 B,
// accumulator mirror 8 KiB, bara 1 KiB
// ------------------------------------------------------------------------------------------
#define WAVE_BRF_LDS_BYTES (2 * WAVE_FXBUF_BYTES + BRF_PARK_EXTRA_BYTES)   /* acc mirror and bara are aliased: blind_rotate_fft.h */
#define BRF_BASE_PAD 128   /* tables | pacing words | pad: every per-wave region then starts at a multiple of 256 bytes and the
                              constant part of the accumulator-mirror addresses folds into the offsets of ds_read2st64_b32 */
static_assert((FTABLE_LDS_BYTES + BR_PACE_BYTES + BRF_BASE_PAD) % 256 == 0 && WAVE_BRF_LDS_BYTES % 256 == 0 && WAVE_FXBUF_BYTES % 256 == 0,
              "per-wave LDS regions of k_bootstrap_fft are 256-byte aligned");

__device__ __forceinline__ BrFftLds carve_brf_lds(int wave)
{
    unsigned char *base = g_smem + FTABLE_LDS_BYTES + BR_PACE_BYTES + BRF_BASE_PAD + wave * WAVE_BRF_LDS_BYTES;
    BrFftLds lds;
    lds.pace.mine = nullptr;
    lds.pace.other = nullptr;
    lds.xbufA = (cplx *)base;
    lds.xbufB = (cplx *)(base + WAVE_FXBUF_BYTES);
    lds.park = (u32 *)(base + 2 * WAVE_FXBUF_BYTES);
    lds.tw1 = (const cplx *)g_smem;
    lds.tw2 = (const cplx *)g_smem + FFT_TW1_ELEMS;
    return lds;
}

__global__ __launch_bounds__(BR_BLOCK_THREADS, 2) void k_bootstrap_fft(BrLaunch P)
{
#if defined(BR_PROBE)
    if (blockIdx.x == 0 && threadIdx.x == 0) g_br_probe[8] = wall_clock64();
#endif
    load_ftables((const cplx *)P.tw_a, (const cplx *)P.tw_b);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long gbit = (long)blockIdx.x * (blockDim.x >> 6) + wave;   // waves per group chosen at launch
    const BrPace pace = carve_pace(g_smem + FTABLE_LDS_BYTES, wave, gbit < P.nbits_total);
    if (gbit >= P.nbits_total) return;
    const int job = gbit >= P.bits_per_job ? 1 : 0;
    const long bit = gbit - (job ? P.bits_per_job : 0);

    BrFftLds lds = carve_brf_lds(wave);
    lds.pace = pace;
    const FftLane L = fft_lane_init(lane);
    const u32 barb = brf_prologue(P.job[job].s0, P.job[job].s1, P.job[job].c0, bit, P.n, lds, lane);
    WAVE_SYNC();
    u32 acc[2][16];
    ClockProbe probe;
    probe.begin(P);
#if defined(BR_PROBE)
    if (blockIdx.x == 0 && threadIdx.x == 0) g_br_probe[9] = wall_clock64();
#endif
    brf_blind_rotate(acc, (const cplx *)P.bk, P.n, barb, P.mu, lds, L);
    probe.end(P);
#if defined(BR_PROBE)
    if (blockIdx.x == 0 && threadIdx.x == 0) g_br_probe[10] = wall_clock64();
#endif
    br_extract<1>(P.out_a + gbit * P.out_a_stride, P.out_b + gbit * P.out_b_stride, acc, lane);
#if defined(BR_PROBE)
    __threadfence();
    if (blockIdx.x == 0 && threadIdx.x == 0) g_br_probe[11] = wall_clock64();
#endif
}

// Medium batches, FFT (CUs < bits <= 3 x CUs): two waves per bit, up to 3 pairs per work-group (blind_rotate_fft.h,
// brfp_*).  LDS: tables | pacing / barrier words | per pair: ACC 8 KiB, bara 1 KiB, 2 x 2 exchange buffers
#define FPAIR_LDS_BYTES (2 * 1024 * 4 + WAVE_BARA_BYTES + 4 * WAVE_FXBUF_BYTES)
#define BRFP_MAX_PAIRS 3
static_assert(FTABLE_LDS_BYTES + BR_PACE_BYTES + BRFP_MAX_PAIRS * FPAIR_LDS_BYTES <= 160 * 1024, "LDS budget of the FFT pair kernel");
__global__ __launch_bounds__(128 * BRFP_MAX_PAIRS, 2) void k_bootstrap_fft_pair(BrLaunch P)
{
    load_ftables((const cplx *)P.tw_a, (const cplx *)P.tw_b);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int pair = wave >> 1;
    const long gbit = (long)blockIdx.x * (blockDim.x >> 7) + pair;   // pairs per group chosen at launch
    const BrPace pace = carve_pace(g_smem + FTABLE_LDS_BYTES, wave, gbit < P.nbits_total);
    if (gbit >= P.nbits_total) return;                                // (both waves of the pair)
    const int job = gbit >= P.bits_per_job ? 1 : 0;
    const long bit = gbit - (job ? P.bits_per_job : 0);
    unsigned char *base = g_smem + FTABLE_LDS_BYTES + BR_PACE_BYTES + pair * FPAIR_LDS_BYTES;
    unsigned char *xb = base + 2 * 1024 * 4 + WAVE_BARA_BYTES;
    BrFftPairLds lds;
    lds.acc = (i32 *)base;
    lds.bara = (uint16_t *)(base + 2 * 1024 * 4);
    lds.xbufA = (cplx *)(xb + (wave & 1) * 2 * WAVE_FXBUF_BYTES);
    lds.xbufB = lds.xbufA + FFT_XBUF_ELEMS;
    lds.xbufA_other = (const cplx *)(xb + ((wave & 1) ^ 1) * 2 * WAVE_FXBUF_BYTES);
    lds.tw1 = (const cplx *)g_smem;
    lds.tw2 = (const cplx *)g_smem + FFT_TW1_ELEMS;
    lds.pace = pace;
    // pair barrier: arrival counters in LDS (see k_bootstrap_pair)
    u32 *arrive = (u32 *)(g_smem + FTABLE_LDS_BYTES) + 8;
    u32 *mine = arrive + wave;
    const u32 *other = arrive + (wave ^ 1);
    u32 seq = 0;
    auto pair_sync = [&] {
        seq++;
        __hip_atomic_store(mine, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        while ((i32)((u32)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(other, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) - seq) < 0)
            __builtin_amdgcn_s_sleep(1);
    };
    const FftLane L = fft_lane_init(lane);
    i32 *out_a = P.out_a + gbit * P.out_a_stride, *out_b = P.out_b + gbit * P.out_b_stride;
    if ((wave & 1) == 0)
        brfp_bootstrap<0>(out_a, out_b, P.job[job].s0, P.job[job].s1, P.job[job].c0, bit, (const cplx *)P.bk, P.n, P.mu,
                          lds, L, pair_sync);
    else
        brfp_bootstrap<1>(out_a, out_b, P.job[job].s0, P.job[job].s1, P.job[job].c0, bit, (const cplx *)P.bk, P.n, P.mu,
                          lds, L, pair_sync);
}

// Small-batch FFT variant: 4 waves per bit (blind_rotate_fft.h, brft_*).
// LDS: tables 9 KiB | ACC 8 KiB | bara 1 KiB | partial sums 64 KiB | 4 exchange buffers
#define TEAMF_LDS_BYTES (FTABLE_LDS_BYTES + 2 * 1024 * 4 + WAVE_BARA_BYTES + BRFT_PART_ELEMS * 16 + BRT_WAVES * WAVE_FXBUF_BYTES)
__global__ __launch_bounds__(64 * BRT_WAVES, 1) void k_bootstrap_fft_team(BrLaunch P)
{
    load_ftables((const cplx *)P.tw_a, (const cplx *)P.tw_b);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long gbit = blockIdx.x;
    const int job = gbit >= P.bits_per_job ? 1 : 0;
    const long bit = gbit - (job ? P.bits_per_job : 0);
    unsigned char *base = g_smem + FTABLE_LDS_BYTES;
    BrFftTeamLds lds;
    lds.acc = (i32 *)base;
    lds.bara = (uint16_t *)(base + 2 * 1024 * 4);
    lds.part = (cplx *)(base + 2 * 1024 * 4 + WAVE_BARA_BYTES);
    lds.xbuf = (cplx *)(base + 2 * 1024 * 4 + WAVE_BARA_BYTES + BRFT_PART_ELEMS * 16 + wave * WAVE_FXBUF_BYTES);
    lds.tw1 = (const cplx *)g_smem;
    lds.tw2 = (const cplx *)g_smem + FFT_TW1_ELEMS;
    const FftLane L = fft_lane_init(lane);
    brft_bootstrap(P.out_a + gbit * P.out_a_stride, P.out_b + gbit * P.out_b_stride, P.job[job].s0, P.job[job].s1,
                   P.job[job].c0, bit, (const cplx *)P.bk, P.n, P.mu, lds, L, WAVE_UNIFORM(wave),
                   [] { __syncthreads(); });
}

__global__ __launch_bounds__(BR_BLOCK_THREADS, 2) void k_blind_rotate_accum_fft(
    i32 *__restrict__ accum, const cplx *__restrict__ bk, const i32 *__restrict__ bara, long bara_stride,
    int row0, int n_rows, int external_mul_only, long batch, const cplx *__restrict__ tw1,
    const cplx *__restrict__ tw2)
{
    load_ftables(tw1, tw2);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long bit = (long)blockIdx.x * BR_WAVES_PER_BLOCK + wave;
    if (bit >= batch) return;
    const BrFftLds lds = carve_brf_lds(wave);
    const FftLane L = fft_lane_init(lane);
    i32 *my = accum + bit * 2048;
    i32 *mirror = brf_acc_mirror(lds);
    u32 acc[2][16];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            acc[m][r] = (u32)my[m * 1024 + lane + 64 * r];
            mirror[m * 1024 + lane + 64 * r] = (i32)acc[m][r];
        }
    WAVE_SYNC();
#if defined(BR_PROBE) && defined(__HIP_DEVICE_COMPILE__)
    BrProbe probe_ = {};
#endif
    if (external_mul_only) {
        u32 res[2][16];
        brf_external_product(res, acc, bk + (long)row0 * BKF_ROW_ELEMS, lds, lds.tw2, L BR_PROBE_PASS);
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][r] = res[m][r];
    } else {
        for (int i = 0; i < n_rows; i++) {
            const u32 a = WAVE_UNIFORM((u32)bara[bit * bara_stride + i]) & 2047u;
            if (a == 0) continue;
            brf_step(acc, a, bk + (long)(row0 + i) * BKF_ROW_ELEMS, lds, lds.tw2, L BR_PROBE_PASS);
        }
    }
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) my[m * 1024 + lane + 64 * r] = (i32)acc[m][r];
}

// tlwe_mask_size = 2 with the FFT transform (blind_rotate_fft.h, brfk_*): per wave two exchange buffers
// + the accumulator int32[3][1024] = 30 KiB.  Four waves per CU, one per SIMD: the (K+1) x 8 complex sums,
// two transforms in flight and the pipelined key loads need more than the 256 VGPRs a second wave on
// the SIMD would leave (a fifth wave would fit the LDS but not the register file)
#define BRF2_WAVES 4
#define WAVE_BRF2_LDS_BYTES (2 * WAVE_FXBUF_BYTES + 3 * 1024 * 4)

__device__ __forceinline__ BrFftLdsK carve_brf2_lds(int wave)
{
    unsigned char *base = g_smem + FTABLE_LDS_BYTES + wave * WAVE_BRF2_LDS_BYTES;
    BrFftLdsK lds;
    lds.xbufA = (cplx *)base;
    lds.xbufB = (cplx *)(base + WAVE_FXBUF_BYTES);
    lds.acc = (i32 *)(base + 2 * WAVE_FXBUF_BYTES);
    lds.tw1 = (const cplx *)g_smem;
    lds.tw2 = (const cplx *)g_smem + FFT_TW1_ELEMS;
    return lds;
}

__global__ __launch_bounds__(64 * BRF2_WAVES, 1) void k_bootstrap_fft_k2(BrLaunch P)
{
    load_ftables((const cplx *)P.tw_a, (const cplx *)P.tw_b);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long gbit = (long)blockIdx.x * (blockDim.x >> 6) + wave;
    if (gbit >= P.nbits_total) return;
    const int job = gbit >= P.bits_per_job ? 1 : 0;
    const long bit = gbit - (job ? P.bits_per_job : 0);
    const BrFftLdsK lds = carve_brf2_lds(wave);
    const FftLane L = fft_lane_init(lane);
    u32 acc[3][16];
    brfk_bootstrap_body<2>(acc, P.job[job].s0, P.job[job].s1, P.job[job].c0, bit, (const cplx *)P.bk, P.n, P.mu, lds, L);
    br_extract<2>(P.out_a + gbit * P.out_a_stride, P.out_b + gbit * P.out_b_stride, acc, lane);
}

// k = 2, FFT, without the partial-sum buffer: 3 waves per bit handing their partial sums round through the exchange
// buffers (blind_rotate_fft.h, brfr_*), up to 2 teams per work-group.  LDS: tables | pacing / barrier words | per team:
// ACC 12 KiB, bara 1 KiB, 3 x 2 exchange buffers
#define FRING2_LDS_BYTES (3 * 1024 * 4 + WAVE_BARA_BYTES + 3 * 2 * WAVE_FXBUF_BYTES)
#define BRFR_MAX_TEAMS 2
static_assert(FTABLE_LDS_BYTES + BR_PACE_BYTES + BRFR_MAX_TEAMS * FRING2_LDS_BYTES <= 160 * 1024, "LDS budget of the FFT ring kernel");
__global__ __launch_bounds__(192 * BRFR_MAX_TEAMS, 2) void k_bootstrap_fft_ring_k2(BrLaunch P)
{
    load_ftables((const cplx *)P.tw_a, (const cplx *)P.tw_b);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int team = wave / 3, w = wave - 3 * team;
    const long gbit = (long)blockIdx.x * (blockDim.x / 192) + team;   // teams per group chosen at launch
    const BrPace pace = carve_pace(g_smem + FTABLE_LDS_BYTES, wave, gbit < P.nbits_total);
    if (gbit >= P.nbits_total) return;                                 // (all three waves of the team)
    const int job = gbit >= P.bits_per_job ? 1 : 0;
    const long bit = gbit - (job ? P.bits_per_job : 0);
    unsigned char *base = g_smem + FTABLE_LDS_BYTES + BR_PACE_BYTES + team * FRING2_LDS_BYTES;
    BrFftRingLds lds;
    lds.acc = (i32 *)base;
    lds.bara = (uint16_t *)(base + 3 * 1024 * 4);
    lds.xbuf_team = (cplx *)(base + 3 * 1024 * 4 + WAVE_BARA_BYTES);
    lds.tw1 = (const cplx *)g_smem;
    lds.tw2 = (const cplx *)g_smem + FFT_TW1_ELEMS;
    lds.pace = pace;
    u32 *arrive = (u32 *)(g_smem + FTABLE_LDS_BYTES) + 8 + 3 * team;   // team barrier: see k_bootstrap_ring_k2
    u32 seq = 0;
    auto team_sync = [&] {
        seq++;
        __hip_atomic_store(arrive + w, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        for (int o = 0; o < 3; o++)
            while ((i32)((u32)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(arrive + o, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) - seq) < 0)
                __builtin_amdgcn_s_sleep(1);
    };
    const FftLane L = fft_lane_init(lane);
    brfr_bootstrap<2>(P.out_a + gbit * P.out_a_stride, P.out_b + gbit * P.out_b_stride, P.job[job].s0, P.job[job].s1,
                      P.job[job].c0, bit, (const cplx *)P.bk, P.n, P.mu, lds, L, w, team_sync);
}

// Small-batch variant of the k = 2 FFT path: 3 waves per bit (blind_rotate_fft.h, brftk_*).
// LDS: tables | ACC 12 KiB | bara 1 KiB | partial sums 72 KiB | 3 x 2 exchange buffers
#define TEAMF2_LDS_BYTES (FTABLE_LDS_BYTES + 3 * 1024 * 4 + WAVE_BARA_BYTES + BRFTK_PART_ELEMS(2) * 16 + 3 * 2 * WAVE_FXBUF_BYTES)
__global__ __launch_bounds__(64 * 3, 1) void k_bootstrap_fft_team_k2(BrLaunch P)
{
    load_ftables((const cplx *)P.tw_a, (const cplx *)P.tw_b);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long gbit = blockIdx.x;
    const int job = gbit >= P.bits_per_job ? 1 : 0;
    const long bit = gbit - (job ? P.bits_per_job : 0);
    unsigned char *base = g_smem + FTABLE_LDS_BYTES;
    BrFftTeamLdsK lds;
    lds.acc = (i32 *)base;
    lds.bara = (uint16_t *)(base + 3 * 1024 * 4);
    lds.part = (cplx *)(base + 3 * 1024 * 4 + WAVE_BARA_BYTES);
    lds.xbufA = (cplx *)(base + 3 * 1024 * 4 + WAVE_BARA_BYTES + BRFTK_PART_ELEMS(2) * 16 + wave * 2 * WAVE_FXBUF_BYTES);
    lds.xbufB = lds.xbufA + FFT_XBUF_ELEMS;
    lds.tw1 = (const cplx *)g_smem;
    lds.tw2 = (const cplx *)g_smem + FFT_TW1_ELEMS;
    const FftLane L = fft_lane_init(lane);
    brftk_bootstrap<2>(P.out_a + gbit * P.out_a_stride, P.out_b + gbit * P.out_b_stride, P.job[job].s0, P.job[job].s1,
                       P.job[job].c0, bit, (const cplx *)P.bk, P.n, P.mu, lds, L, WAVE_UNIFORM(wave),
                       [] { __syncthreads(); });
}

__global__ __launch_bounds__(64 * BRF2_WAVES, 1) void k_blind_rotate_accum_fft_k2(
    i32 *__restrict__ accum, const cplx *__restrict__ bk, const i32 *__restrict__ bara, long bara_stride,
    int row0, int n_rows, int external_mul_only, long batch, const cplx *__restrict__ tw1,
    const cplx *__restrict__ tw2)
{
    load_ftables(tw1, tw2);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long bit = (long)blockIdx.x * BRF2_WAVES + wave;
    if (bit >= batch) return;
    const BrFftLdsK lds = carve_brf2_lds(wave);
    const FftLane L = fft_lane_init(lane);
    i32 *my = accum + bit * 3 * 1024;
#pragma unroll
    for (int m = 0; m < 3; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) lds.acc[m * 1024 + lane + 64 * r] = my[m * 1024 + lane + 64 * r];
    WAVE_SYNC();
    constexpr long kRow = (long)BK_ROW_POLYS(2) * BKF_POLY_ELEMS;
    if (external_mul_only) {
        brfk_external_product<2>(
            [&](int m, u32(&T)[16]) {
#pragma unroll
                for (int r = 0; r < 16; r++) T[r] = (u32)lds.acc[m * 1024 + lane + 64 * r];
            },
            [&](int mo, int r, u32 v) { lds.acc[mo * 1024 + lane + 64 * r] = (i32)v; }, bk + row0 * kRow, lds, L);
    } else {
        for (int i = 0; i < n_rows; i++) {
            const u32 a = WAVE_UNIFORM((u32)bara[bit * bara_stride + i]) & 2047u;
            if (a == 0) continue;
            brfk_step<2>(a, bk + (row0 + i) * kRow, lds, L);
        }
    }
    WAVE_SYNC();
#pragma unroll
    for (int m = 0; m < 3; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) my[m * 1024 + lane + 64 * r] = lds.acc[m * 1024 + lane + 64 * r];
}

#define FFT_BLOCK_THREADS 256
#define FFT_WAVES_PER_BLOCK 4

__device__ __forceinline__ cplx *carve_fft_xbuf(int wave)
{
    return (cplx *)(g_smem + FTABLE_LDS_BYTES + wave * WAVE_FXBUF_BYTES);
}

// forward folded FFT of int32 polynomials; WAVE_LAYOUT = false: natural frequency order
// (fft_transform_ref), true: the key layout
template <bool WAVE_LAYOUT>
__global__ __launch_bounds__(FFT_BLOCK_THREADS) void k_fft_forward(cplx *__restrict__ out, const i32 *__restrict__ in,
                                                                   long batch, const cplx *__restrict__ tw1,
                                                                   const cplx *__restrict__ tw2)
{
    load_ftables(tw1, tw2);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long p = (long)blockIdx.x * FFT_WAVES_PER_BLOCK + wave;
    if (p >= batch) return;
    const FftLane L = fft_lane_init(lane);
    cplx x[8];
#pragma unroll
    for (int r = 0; r < 8; r++)
        x[r] = cplx{(double)in[p * 1024 + lane + 64 * r], -(double)in[p * 1024 + lane + 64 * r + 512]};
    fft_forward(x, carve_fft_xbuf(wave), (const cplx *)g_smem, (const cplx *)g_smem + FFT_TW1_ELEMS, L);
#pragma unroll
    for (int r = 0; r < 8; r++)
        out[p * 512 + (WAVE_LAYOUT ? bkf_elem_offset(lane, r) : fft_freq_index(lane, r))] = x[r];
}

__global__ __launch_bounds__(FFT_BLOCK_THREADS) void k_fft_inverse(i32 *__restrict__ out, const cplx *__restrict__ in,
                                                                   long batch, const cplx *__restrict__ tw1,
                                                                   const cplx *__restrict__ tw2)
{
    load_ftables(tw1, tw2);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long p = (long)blockIdx.x * FFT_WAVES_PER_BLOCK + wave;
    if (p >= batch) return;
    const FftLane L = fft_lane_init(lane);
    cplx x[8];
#pragma unroll
    for (int r = 0; r < 8; r++) x[r] = in[p * 512 + fft_freq_index(lane, r)];
    fft_inverse(x, carve_fft_xbuf(wave), (const cplx *)g_smem, (const cplx *)g_smem + FFT_TW1_ELEMS, L);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        out[p * 1024 + lane + 64 * r] = (i32)fft_round_to_u32(x[r].re);
        out[p * 1024 + lane + 64 * r + 512] = (i32)fft_round_to_u32(-x[r].im);
    }
}

// natural order <-> wave layout of FFT-domain key polynomials (pure permutation)
__global__ void k_bkf_permute(cplx *__restrict__ out, const cplx *__restrict__ in, long polys, int to_reference)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= polys * 512) return;
    const long p = t >> 9;
    const int e = (int)(t & 511), lane = e & 63, reg = e >> 6;
    if (to_reference) out[p * 512 + fft_freq_index(lane, reg)] = in[p * 512 + bkf_elem_offset(lane, reg)];
    else out[p * 512 + bkf_elem_offset(lane, reg)] = in[p * 512 + fft_freq_index(lane, reg)];
}

// ------------------------------------------------------------------------------------------
// K3/K4: standalone batched transforms, natural order, one wave per polynomial
// ------------------------------------------------------------------------------------------
#define NTT_BLOCK_THREADS 256
#define NTT_WAVES_PER_BLOCK 4

__device__ __forceinline__ u64 *carve_ntt_xbuf(int wave)
{
    return (u64 *)(g_smem + TABLE_LDS_BYTES + wave * WAVE_XBUF_BYTES);
}

// mode: 0 = i32 in (forward) / i32 out (inverse); 1 = u64 field elements
template <int MODE>
__global__ __launch_bounds__(NTT_BLOCK_THREADS) void k_ntt_forward(u64 *__restrict__ out, const void *__restrict__ in,
                                                                   long batch, const u64 *__restrict__ tw1f,
                                                                   const u64 *__restrict__ tw1i)
{
    load_tables(tw1f, tw1i);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long p = (long)blockIdx.x * NTT_WAVES_PER_BLOCK + wave;
    if (p >= batch) return;
    const NttLane L = ntt_lane_init(lane);
    u64 x[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const long idx = p * 1024 + ntt_coef_index(lane, r);
        x[r] = MODE == 0 ? ff_from_i32(((const i32 *)in)[idx]) : ff_canon(((const u64 *)in)[idx]);
    }
    ntt_forward(x, carve_ntt_xbuf(wave), (const u64 *)g_smem, L);
#pragma unroll
    for (int r = 0; r < 16; r++) out[p * 1024 + ntt_freq_index(lane, r)] = x[r];
}

template <int MODE>
__global__ __launch_bounds__(NTT_BLOCK_THREADS) void k_ntt_inverse(void *__restrict__ out, const u64 *__restrict__ in,
                                                                   long batch, const u64 *__restrict__ tw1f,
                                                                   const u64 *__restrict__ tw1i)
{
    load_tables(tw1f, tw1i);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long p = (long)blockIdx.x * NTT_WAVES_PER_BLOCK + wave;
    if (p >= batch) return;
    const NttLane L = ntt_lane_init(lane);
    u64 x[16];
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = ff_canon(in[p * 1024 + ntt_freq_index(lane, r)]);
    ntt_inverse(x, carve_ntt_xbuf(wave), (const u64 *)g_smem + 1024, L);
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const long idx = p * 1024 + ntt_coef_index(lane, r);
        if (MODE == 0) ((i32 *)out)[idx] = ff_to_i32(x[r]);
        else ((u64 *)out)[idx] = x[r];
    }
}

// Negacyclic product of int32 polynomials mod 2^32: out[b] = [base[b] +] x[b] * y[b % y_batch]
// (strides in elements; base may alias out: TLweEncryptZero accumulates b = noise + sum_i a_i * s_i,
// tlwe_cpu.py:76-84)
__global__ __launch_bounds__(NTT_BLOCK_THREADS) void k_poly_mul(i32 *out, long out_stride, const i32 *__restrict__ x,
                                                                long x_stride, const i32 *__restrict__ y,
                                                                const i32 *base, long base_stride, long batch,
                                                                long y_batch, const u64 *__restrict__ tw1f,
                                                                const u64 *__restrict__ tw1i)
{
    load_tables(tw1f, tw1i);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long p = (long)blockIdx.x * NTT_WAVES_PER_BLOCK + wave;
    if (p >= batch) return;
    const NttLane L = ntt_lane_init(lane);
    u64 *xbuf = carve_ntt_xbuf(wave);
    u64 fx[16], fy[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        fx[r] = ff_from_i32(x[p * x_stride + ntt_coef_index(lane, r)]);
        fy[r] = ff_from_i32(y[(p % y_batch) * 1024 + ntt_coef_index(lane, r)]);
    }
    ntt_forward(fx, xbuf, (const u64 *)g_smem, L);
    ntt_forward(fy, xbuf, (const u64 *)g_smem, L);
#pragma unroll
    for (int r = 0; r < 16; r++) fx[r] = ff_mul(fx[r], fy[r]);
    ntt_inverse(fx, xbuf, (const u64 *)g_smem + 1024, L);
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int j = ntt_coef_index(lane, r);
        u32 v = (u32)ff_to_i32(fx[r]);
        if (base) v += (u32)base[p * base_stride + j];
        out[p * out_stride + j] = (i32)v;
    }
}

// ------------------------------------------------------------------------------------------
// Bootstrapping-key format conversion
// ------------------------------------------------------------------------------------------
// reference format (natural-order NTT, Montgomery: x * 2^64) -> wave layout, plain residues
__global__ void k_bk_from_reference(u64 *__restrict__ out, const u64 *__restrict__ in, long polys)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= polys * 1024) return;
    const long p = t >> 10;
    const int e = (int)(t & 1023), lane = e & 63, reg = e >> 6;
    out[p * 1024 + bk_elem_offset(lane, reg)] = ff_mul_pow2<128>(ff_canon(in[p * 1024 + ntt_freq_index(lane, reg)]));
}

__global__ void k_bk_to_reference(u64 *__restrict__ out, const u64 *__restrict__ in, long polys)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= polys * 1024) return;
    const long p = t >> 10;
    const int e = (int)(t & 1023), lane = e & 63, reg = e >> 6;
    out[p * 1024 + ntt_freq_index(lane, reg)] = ff_mul_pow2<64>(in[p * 1024 + bk_elem_offset(lane, reg)]);
}

// ------------------------------------------------------------------------------------------
// Unit-granularity hooks for the two steps that are otherwise only reachable fused inside the blind rotation
// (reference test surface: test/test_tgsw.py:44-69 and :72-115).
// ------------------------------------------------------------------------------------------

// TGswTorus32PolynomialDecompH (tgsw_gpu.py / tgsw_cpu.py:26-49): sample int32 [polys][1024] ->
// result int32 [polys][2][1024], the two gadget digits of every coefficient (br_digit, the hot path's function)
__global__ void k_tgsw_decompose(i32 *__restrict__ result, const i32 *__restrict__ sample, long polys)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= polys * 1024) return;
    const long p = t >> 10;
    const int j = (int)(t & 1023);
    const u32 v = (u32)sample[t];
    result[(p * 2 + 0) * 1024 + j] = br_digit<0>(v);
    result[(p * 2 + 1) * 1024 + j] = br_digit<1>(v);
}

// TLweTransformedAddMulTo (tgsw_cpu.py:52-79; tgsw_gpu.mako MAC of the fused kernel): transformed-domain
//   result[b][mo][f] = sum_{m, d} sample[b][m][d][f] * bk[row][m][d][mo][f] / 2^64   (mod P)
// on arrays in the REFERENCE's format (natural order, key Montgomery-prepared: the factor 2^-64 is the shift the key
// upload applies, k_bk_from_reference).  The products run through ff_dot2 -- two products, one reduction, the
// multiply-accumulate of br_mac2 -- in the order of the hot path (d = 0, 1 paired, then over m); field addition is
// exact, so the value equals the reference's sequential sum.
template <int K>
__global__ void k_tgsw_mac(u64 *__restrict__ result, const u64 *__restrict__ sample, const u64 *__restrict__ bk,
                           int bk_row, long batch)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= batch * (K + 1) * 1024) return;
    const int f = (int)(t & 1023);
    const int mo = (int)((t >> 10) % (K + 1));
    const long b = (t >> 10) / (K + 1);
    const u64 *row = bk + (long)bk_row * BK_ROW_ELEMS_K(K);
    u64 acc = 0;
#pragma unroll
    for (int m = 0; m <= K; m++) {
        const u64 x0 = ff_canon(sample[((b * (K + 1) + m) * 2 + 0) * 1024 + f]);
        const u64 x1 = ff_canon(sample[((b * (K + 1) + m) * 2 + 1) * 1024 + f]);
        const u64 k0 = ff_mul_pow2<128>(ff_canon(row[((m * 2 + 0) * (K + 1) + mo) * 1024 + f]));   // * 2^-64 (2^192 = 1)
        const u64 k1 = ff_mul_pow2<128>(ff_canon(row[((m * 2 + 1) * (K + 1) + mo) * 1024 + f]));
        acc = (m == 0) ? ff_dot2<false>(x0, k0, x1, k1, 0) : ff_dot2<true>(x0, k0, x1, k1, acc);
    }
    result[t] = ff_canon(acc);
}

// wave layout -> half-ring layout [poly][h][reg 8][lane 64] of the same field elements (k_bootstrap_team8)
__global__ void k_bk_to_half(u64 *__restrict__ out, const u64 *__restrict__ in, long polys)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= polys * 1024) return;
    const long p = t >> 10;
    const int e = (int)(t & 1023), h = e >> 9, reg = (e >> 6) & 7, lane = e & 63;
    const int k = nth_freq_index(h, lane, reg);
    const int k2 = k & 15, k1a = (k >> 4) & 15, k1b = k >> 8;                  // inverse of ntt_freq_index
    out[t] = in[p * 1024 + bk_elem_offset(4 * k2 + (k1a & 3), 4 * (k1a >> 2) + k1b)];
}

// coefficient-domain TGSW polynomials (int32) -> wave layout (forward NTT only; no Montgomery)
__global__ __launch_bounds__(NTT_BLOCK_THREADS) void k_bk_from_coeffs(u64 *__restrict__ out, const i32 *__restrict__ in,
                                                                      long polys, const u64 *__restrict__ tw1f,
                                                                      const u64 *__restrict__ tw1i)
{
    load_tables(tw1f, tw1i);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long p = (long)blockIdx.x * NTT_WAVES_PER_BLOCK + wave;
    if (p >= polys) return;
    const NttLane L = ntt_lane_init(lane);
    u64 x[16];
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = ff_from_i32(in[p * 1024 + ntt_coef_index(lane, r)]);
    ntt_forward(x, carve_ntt_xbuf(wave), (const u64 *)g_smem, L);
#pragma unroll
    for (int r = 0; r < 16; r++) out[p * 1024 + bk_elem_offset(lane, r)] = x[r];
}

// NTT key in the wave layout -> coefficient-domain TGSW polynomials (int32): the inverse of k_bk_from_coeffs (the exact-FFT
// engine builds its split key image from them, nufhe_cloudkey_set_engine)
// *not_int32 is set when a coefficient is not a centred 32-bit integer, i.e. the key is not the transform of int32 torus
// polynomials (synthetic keys of random field elements): the exact engine cannot serve such a key and says so.
__global__ __launch_bounds__(NTT_BLOCK_THREADS) void k_bk_to_coeffs(i32 *__restrict__ out, const u64 *__restrict__ in,
                                                                    long polys, const u64 *__restrict__ tw1f,
                                                                    const u64 *__restrict__ tw1i, int *__restrict__ not_int32)
{
    load_tables(tw1f, tw1i);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long p = (long)blockIdx.x * NTT_WAVES_PER_BLOCK + wave;
    if (p >= polys) return;
    const NttLane L = ntt_lane_init(lane);
    u64 x[16];
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = ff_canon(in[p * 1024 + bk_elem_offset(lane, r)]);
    ntt_inverse(x, carve_ntt_xbuf(wave), (const u64 *)g_smem + 1024, L);
#pragma unroll
    for (int r = 0; r < 16; r++) out[p * 1024 + ntt_coef_index(lane, r)] = ff_to_i32(x[r]);
    bool bad = false;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const u64 v = ff_canon(x[r]);
        bad |= !(v < (1ull << 31) || v >= FF_P - (1ull << 31));
    }
    if (bad) atomicOr(not_int32, 1);
}

// ------------------------------------------------------------------------------------------
// K2: LWE keyswitch (lwe_gpu.mako:59-120, lwe_cpu.py:62-93)
//   out_a[bit][c] = - sum_{j < 1024, k < 8} KS_a[j][k][digit_jk(src_a[bit][j])][c]
// A block owns KS_TILE_BITS bits x all columns for one slice of j: the non-zero key rows of every
// (j, k) are loaded ONCE per block (coalesced over columns), staged in LDS, and each bit picks its row
// by its wave-uniform digit (an LDS address); per-thread accumulators stay in registers.  j-slices
// combine with integer atomics (order independent => bit-exact).
// ------------------------------------------------------------------------------------------
#define KS_ROW_WORDS 512
#define KS_READLANE(v, lane) ((u32)__builtin_amdgcn_readlane((int)(v), (lane)))

// Two digit positions (k, k+1) of an input coefficient are handled at once ("window"): the block
// pre-adds the 3 + 3 non-zero rows of the pair into the 15 non-zero combinations
// row[4 d0 + d1] = KS[j][k][d0] + KS[j][k+1][d1] (sums mod 2^32: exact, the result is a sum anyway) in
// LDS, and every bit then needs ONE row per pair, selected by its 4-bit double digit: half the LDS
// reads and half the per-bit instructions of a digit-by-digit loop.  LDS: 2 stages x 16 rows x 2 KiB.
#define KSW_STAGE_WORDS (16 * KS_ROW_WORDS)
#define KS_LDS_BYTES (2 * KSW_STAGE_WORDS * sizeof(u32))
__global__ __launch_bounds__(KS_BLOCK_THREADS) void k_keyswitch_a(KsLaunch P)
{
    constexpr int TILE = KS_TILE_BITS;
    extern __shared__ __attribute__((aligned(16))) u32 ks_lds[];   // 2 * KSW_STAGE_WORDS
    const int tid = threadIdx.x;
    const long bit0 = (long)blockIdx.x * TILE;
    const int j0 = blockIdx.y * P.j_per_block;
    const int c0 = 2 * tid, c1 = 2 * tid + 1;
    const bool v0 = c0 < P.n, v1 = c1 < P.n;
    const int nb = (int)((P.nbits - bit0) < TILE ? (P.nbits - bit0) : TILE);

    u32 acc0[TILE], acc1[TILE];
#pragma unroll
    for (int t = 0; t < TILE; t++) { acc0[t] = 0; acc1[t] = 0; }
    *(u64 *)&ks_lds[c0] = 0;
    *(u64 *)&ks_lds[KSW_STAGE_WORDS + c0] = 0;

    u32 pa[6], pb[6];
    auto load_rows = [&](int step) {
        const i32 *rows = P.ks_a3 + ((long)((j0 * 4 + step) * 2) * 3) * P.n;   // (j, k = 2 kp): 6 consecutive rows
#pragma unroll
        for (int h = 0; h < 6; h++) {
            pa[h] = v0 ? (u32)rows[h * P.n + c0] : 0u;
            pb[h] = v1 ? (u32)rows[h * P.n + c1] : 0u;
        }
    };
    load_rows(0);
    const int steps = P.j_per_block * 4;
    u32 aj[TILE];
    const int my_bit = tid & 63;
    auto load_aj = [&](int j) -> u32 {
        u32 v = 0;
        if (my_bit < nb) {
            v = (u32)P.src1_a[(bit0 + my_bit) * P.src1_stride + j];
            if (P.src2_a) v += (u32)P.src2_a[(bit0 + my_bit) * P.src2_stride + j];
        }
        return v;
    };
    u32 aj_lane = load_aj(j0);
    for (int s = 0; s < steps; s++) {
        const int j = j0 + (s >> 2), kp = s & 3;
        u64 *stage = (u64 *)(ks_lds + (s & 1) * KSW_STAGE_WORDS) + tid;
#pragma unroll
        for (int d0 = 0; d0 < 4; d0++)
#pragma unroll
            for (int d1 = 0; d1 < 4; d1++) {
                if (d0 == 0 && d1 == 0) continue;
                const u32 x0 = (d0 ? pa[d0 - 1] : 0u) + (d1 ? pa[2 + d1] : 0u);
                const u32 x1 = (d0 ? pb[d0 - 1] : 0u) + (d1 ? pb[2 + d1] : 0u);
                stage[(4 * d0 + d1) * (KS_ROW_WORDS / 2)] = ((u64)x1 << 32) | x0;
            }
        if (s + 1 < steps) load_rows(s + 1);
        if (kp == 0) {
#pragma unroll
            for (int t = 0; t < TILE; t++) aj[t] = KS_READLANE(aj_lane, t) + (1u << 15);
            if (s + 4 < steps) aj_lane = load_aj(j + 1);
        }
        __syncthreads();
        const int sh = 28 - 4 * kp;
#pragma unroll
        for (int t = 0; t < TILE; t++) {
            const u32 dd = (aj[t] >> sh) & 15u;          // 4 d0 + d1, lwe_cpu.py:76
            const u64 v = stage[dd * (KS_ROW_WORDS / 2)];
            acc0[t] -= (u32)v;
            acc1[t] -= (u32)(v >> 32);
        }
    }
#pragma unroll
    for (int t = 0; t < TILE; t++) {
        if (t < nb) {
            if (v0) atomicAdd(&P.acc[(bit0 + t) * P.n + c0], acc0[t]);
            if (v1) atomicAdd(&P.acc[(bit0 + t) * P.n + c1], acc1[t]);
        }
    }
}

// entry of an ascending-`start` table that holds row `r` (the last one whose start <= r; empty entries are skipped
// because their successor has the same start)
template <typename T>
__device__ __forceinline__ int batch_find(const T *tab, int n, long r)
{
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].start <= r) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// finalize: copy the accumulated mask into the result view, and compute b and the variance.
// The float32 variance must be summed in the reference's sequential (j, k) order to be
// bit-identical (lwe_cpu.py:80-92): the block gathers the selected terms into LDS in parallel
// (1024 input coefficients = 8192 terms at a time), then one lane adds them in order.
__global__ __launch_bounds__(256) void k_keyswitch_finalize(KsFinal P)
{
    __shared__ float cv_terms[8192];
    __shared__ u32 b_sum, nz_sum;
    const long bit = blockIdx.x;
    if (threadIdx.x == 0) { b_sum = 0; nz_sum = 0; }
    __syncthreads();
    i32 *out_a = P.out_a + bit * P.out_a_stride, *out_b = P.out_b + bit * P.out_b_stride;
    float *out_cv = P.out_cv ? P.out_cv + bit * P.out_b_stride : nullptr;
    if (P.batch_outs) {
        const BatchOut &o = P.batch_outs[batch_find(P.batch_outs, P.n_batch_outs, bit)];
        out_a = o.a + (bit - o.start) * o.a_stride;
        out_b = o.b + (bit - o.start) * o.b_stride;
        out_cv = o.cv ? o.cv + (bit - o.start) * o.b_stride : nullptr;
    }
    for (int c = threadIdx.x; c < P.n; c += blockDim.x)
        out_a[c] = (i32)P.acc[bit * P.n + c];
    const bool by_count = P.cv_table != nullptr;     // uniform-variance key: count instead of adding
    u32 bpart = 0, nz = 0;
    float cv = 0.0f;
    for (int j0 = 0; j0 < P.input_size; j0 += 1024) {
        for (int jj = threadIdx.x; jj < 1024; jj += blockDim.x) {
            const int j = j0 + jj;
            u32 a = (u32)P.src1_a[bit * P.src1_stride + j];
            if (P.src2_a) a += (u32)P.src2_a[bit * P.src2_stride + j];
            a += 1u << 15;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const u32 dg = (a >> (30 - 2 * k)) & 3u;
                const long idx = ((long)j * 8 + k) * 4 + (long)dg;
                bpart += (u32)P.ks_b[idx];
                nz += dg != 0u;
                if (!by_count) cv_terms[jj * 8 + k] = P.ks_cv[idx];
            }
        }
        if (!by_count) {
            __syncthreads();
            if (threadIdx.x == 0 && out_cv)
                for (int i = 0; i < 8192; i++) cv += cv_terms[i];
            __syncthreads();
        }
    }
    atomicAdd(&b_sum, bpart);
    if (by_count) atomicAdd(&nz_sum, nz);
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 b = (u32)P.c0 + (u32)P.src1_b[bit * P.src1_bstride];
        if (P.src2_b) b += (u32)P.src2_b[bit * P.src2_bstride];
        *out_b = (i32)(b - b_sum);
        if (out_cv) *out_cv = by_count ? P.cv_table[nz_sum] : cv;
    }
}

// ------------------------------------------------------------------------------------------
// Small element-wise kernels
// ------------------------------------------------------------------------------------------
// LweLinear, lwe_gpu.mako:123-169 / lwe_cpu.py:115-123
__global__ void k_lwe_linear(LweView res, LweView src, i32 p, int add_result, long nbits, int size)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nbits * (size + 1)) return;
    const long bit = t / (size + 1);
    const int i = (int)(t - bit * (size + 1));
    if (i < size) {
        const u32 v = (u32)p * (u32)src.a[bit * src.a_stride + i];
        i32 *r = &res.a[bit * res.a_stride + i];
        *r = (i32)((add_result ? (u32)*r : 0u) + v);
    } else {
        const u32 v = (u32)p * (u32)src.b[bit * src.b_stride];
        i32 *r = &res.b[bit * res.b_stride];
        *r = (i32)((add_result ? (u32)*r : 0u) + v);
        if (res.cv) {
            const float sv = src.cv ? src.cv[bit * src.b_stride] : 0.0f;
            float *c = &res.cv[bit * res.b_stride];
            *c = (add_result ? *c : 0.0f) + (float)((long)p * p) * sv;
        }
    }
}

// Heterogeneous gate batch, step 0: the job tables reach the device as KERNEL ARGUMENTS (a few hundred bytes for a typical
// circuit step) -- no host staging buffer whose lifetime the stream would have to respect, and capturable into a hipGraph
struct TableChunk {
    uint4 w[TABLE_CHUNK_BYTES / 16];
};
__global__ __launch_bounds__(64) void k_write_table(uint4 *__restrict__ dst, TableChunk c, int words)
{
    for (int i = threadIdx.x; i < words; i += 64) dst[i] = c.w[i];
}

// Heterogeneous gate batch, step 1: the linear pre-combination of every gate of the batch (gates.py:104-110 and
// siblings: (0, c) + pa a + pb b, int32 wraparound) into the rows of ONE LWE(n) array, so that a single bootstrap launch
// covers independent gates of different kinds.  One work-group per row.
__global__ __launch_bounds__(256) void k_batch_combine(i32 *__restrict__ out_a, i32 *__restrict__ out_b,
                                                       const BatchRot *__restrict__ rots, int n_rots, int n)
{
    const long row = blockIdx.x;
    const BatchRot &g = rots[batch_find(rots, n_rots, row)];
    const long bit = row - g.start;
    const i32 *a0 = g.a0 + bit * g.a0_stride;
    const i32 *a1 = g.p1 ? g.a1 + bit * g.a1_stride : nullptr;
    for (int c = threadIdx.x; c < n; c += blockDim.x) {
        u32 v = (u32)g.p0 * (u32)a0[c];
        if (a1) v += (u32)g.p1 * (u32)a1[c];
        out_a[row * n + c] = (i32)v;
    }
    if (threadIdx.x == 0) {
        u32 v = (u32)g.c0 + (u32)g.p0 * (u32)g.b0[bit * g.b0_stride];
        if (a1) v += (u32)g.p1 * (u32)g.b1[bit * g.b1_stride];
        out_b[row] = (i32)v;
    }
}

// step 3 (MUX gates of a batch only): (0, mu) + u1 + u2 under the extracted key (gates.py:657-661), in place in u1's rows
__global__ __launch_bounds__(256) void k_batch_mux_fold(i32 *__restrict__ ext_a, i32 *__restrict__ ext_b,
                                                        const BatchOut *__restrict__ outs, int n_outs, int ext, i32 mu)
{
    const long bit = blockIdx.x;
    const BatchOut &o = outs[batch_find(outs, n_outs, bit)];
    if (o.second < 0) return;
    const long row2 = o.second + (bit - o.start);
    for (int c = threadIdx.x; c < ext; c += blockDim.x)
        ext_a[bit * ext + c] = (i32)((u32)ext_a[bit * ext + c] + (u32)ext_a[row2 * ext + c]);
    if (threadIdx.x == 0) ext_b[bit] = (i32)((u32)ext_b[bit] + (u32)ext_b[row2] + (u32)mu);
}

// LweNoiselessTrivialConstant, lwe_cpu.py:136-143
__global__ void k_lwe_trivial_const(LweView res, i32 mu, long nbits, int size)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nbits * (size + 1)) return;
    const long bit = t / (size + 1);
    const int i = (int)(t - bit * (size + 1));
    if (i < size) res.a[bit * res.a_stride + i] = 0;
    else {
        res.b[bit * res.b_stride] = mu;
        if (res.cv) res.cv[bit * res.b_stride] = 0.0f;
    }
}

// LWE phase / dot product: out[i] = base[i] + sign * sum_j a[i][j] * key[j]  (int32 wraparound).
// LweEncrypt (b = mu + e + a.s, lwe_cpu.py:96-104), LweDecrypt (phi = b - a.s, :107-112) and the
// body of MakeLweKeyswitchKey (:36) are all this kernel.  One wave per sample.
__global__ __launch_bounds__(256) void k_lwe_phase(i32 *__restrict__ out, long out_stride, const i32 *__restrict__ a,
                                                   long a_stride, const i32 *__restrict__ base, long base_stride,
                                                   const i32 *__restrict__ key, i32 sign, long count, int n)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long i = (long)blockIdx.x * 4 + wave;
    if (i >= count) return;
    u32 acc = 0;
    for (int j = lane; j < n; j += 64) acc += (u32)a[i * a_stride + j] * (u32)key[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += (u32)__shfl_xor((int)acc, off, 64);
    if (lane == 0) out[i * out_stride] = (i32)((u32)base[i * base_stride] + (u32)sign * acc);
}

// MakeLweKeyswitchKey (lwe_gpu.mako:18-56, lwe_cpu.py:27-59) on the device, straight into the library's
// key layout.  Row r = (j, k, h - 1), j < ext_size, k < 8, h = 1..3; the uniform masks noises_a
// [rows][n] ARE the key's `a` (digits 1..3, the base-0 slice is all zero and is not stored);
//   b[j][k][h] = in_key[j] * h * 2^(32 - 2 (k + 1)) + noises_b[r] + <noises_a[r], out_key>
//   cv[j][k][h] = variance;  b[j][k][0] = cv[j][k][0] = 0.       One wave per row.
__global__ __launch_bounds__(256) void k_ks_make(i32 *__restrict__ ks_b, float *__restrict__ ks_cv,
                                                 const i32 *__restrict__ noises_a, const i32 *__restrict__ noises_b,
                                                 const i32 *__restrict__ in_key, const i32 *__restrict__ out_key,
                                                 float variance, long rows, int n)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + wave;
    if (r >= rows) return;
    u32 acc = 0;
    for (int i = lane; i < n; i += 64) acc += (u32)noises_a[r * n + i] * (u32)out_key[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += (u32)__shfl_xor((int)acc, off, 64);
    if (lane == 0) {
        const long jk = r / 3;
        const int h = (int)(r % 3) + 1, k = (int)(jk % 8);
        const long j = jk / 8;
        const u32 message = (u32)in_key[j] * (u32)h * (1u << (32 - 2 * (k + 1)));      // lwe_cpu.py:54
        ks_b[jk * 4 + h] = (i32)(message + (u32)noises_b[r] + acc);
        ks_cv[jk * 4 + h] = variance;
        if (h == 1) { ks_b[jk * 4] = 0; ks_cv[jk * 4] = 0.0f; }
    }
}

// keyswitch key, library layout -> reference layout a [rows/3][4][n] (base-0 slice zero-filled)
__global__ void k_ks_to_reference(i32 *__restrict__ out_a, const i32 *__restrict__ ks_a3, long groups, int n)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= groups * 4 * n) return;
    const int i = (int)(t % n), h = (int)((t / n) % 4);
    const long g = t / ((long)4 * n);
    out_a[t] = h == 0 ? 0 : ks_a3[(g * 3 + (h - 1)) * n + i];
}

// TGswAddMessage (tgsw_gpu.mako:18-39, tgsw_cpu.py:109-126): TGSW sample s of shape [k+1][l = 2][k+1][1024];
// coefficient 0 of polynomial m of row (m, d) += message[s] * 2^(32 - 10 (d + 1))
__global__ void k_tgsw_add_message(i32 *__restrict__ tgsw, const i32 *__restrict__ messages, long count, int k1)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count * k1 * 2) return;
    const int d = (int)(t % 2), m = (int)((t / 2) % k1);
    const long s = t / (2 * k1);
    i32 *p = tgsw + (((s * k1 + m) * 2 + d) * k1 + m) * 1024;
    p[0] = (i32)((u32)p[0] + (u32)messages[s] * (1u << (32 - 10 * (d + 1))));
}

// Torus32ToPhase, numeric_functions_cpu.py:23-37
__global__ void k_t32_to_phase(i32 *__restrict__ result, const i32 *__restrict__ phase, long count, u32 interv)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    result[t] = (i32)(((u32)phase[t] + interv / 2) / interv);
}

// ShiftTorusPolynomial, polynomials_cpu.py:25-59 (N = 1024)
__global__ void k_shift_tp(i32 *__restrict__ result, const i32 *__restrict__ source, const i32 *__restrict__ powers,
                           long powers_stride, long powers_idx, long batch, int polys, int minus_one,
                           int invert_powers)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= batch * polys * 1024) return;
    const long b = t / ((long)polys * 1024);
    const u32 j = (u32)(t & 1023);
    u32 pw = (u32)powers[b * powers_stride + powers_idx];
    if (invert_powers) pw = 2048u - pw;
    const u32 s = (j - pw) & 2047u;
    const long base = t - j;
    const u32 v = (u32)source[base + (s & 1023u)];
    u32 r = (s & 1024u) ? 0u - v : v;
    if (minus_one) r -= (u32)source[t];
    result[t] = (i32)r;
}

// tlwe_extract_lwe_samples, tlwe_cpu.py:41-60: tlwe [batch][k+1][1024] -> a [batch][k*1024], b [batch]
__global__ void k_tlwe_extract(i32 *__restrict__ ra, i32 *__restrict__ rb, const i32 *__restrict__ tlwe, long batch,
                               int mask_size)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= batch * mask_size * 1024) return;
    const long b = t / (mask_size * 1024);
    const int mj = (int)(t - b * mask_size * 1024), m = mj >> 10, j = mj & 1023;
    const i32 *A = tlwe + (b * (mask_size + 1) + m) * 1024;
    ra[t] = j == 0 ? A[0] : (i32)(0u - (u32)A[1024 - j]);
    if (mj == 0) rb[b] = tlwe[(b * (mask_size + 1) + mask_size) * 1024];
}

// ------------------------------------------------------------------------------------------
// Test hook for the GF(P) primitives as the DEVICE compiles them (ff.h has device-only instruction
// sequences: borrow chains, v_mad_u64_u32 carry-out); the reference tests its finite-field module the
// same way (test/test_transform/test_arithmetic.py).  op: 0 add, 1 sub, 2 mul, 3 a*b + c*d,
// 4 a*b + c*d + e, 5 a * 2^(b & 31) (per-element shift), 6 a * 2^shift (compile-time shift family,
// any shift in [0, 192)), 7 reduce96(a, (u32)b), 8 from_i32((i32)a) / to_i32 round trip (out = to_i32)
// ------------------------------------------------------------------------------------------
template <int S>
struct FfShiftDispatch {
    __device__ static u64 run(u64 x, int s) { return s == S ? ff_mul_pow2<S>(x) : FfShiftDispatch<S - 1>::run(x, s); }
};
template <>
struct FfShiftDispatch<-1> {
    __device__ static u64 run(u64 x, int) { return x; }
};

__global__ void k_ff_op(u64 *__restrict__ out, const u64 *__restrict__ a, const u64 *__restrict__ b,
                        const u64 *__restrict__ c, const u64 *__restrict__ d, const u64 *__restrict__ e, int op,
                        int shift, long count)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    u64 r = 0;
    switch (op) {
    case 0: r = ff_add(a[i], b[i]); break;
    case 1: r = ff_sub(a[i], b[i]); break;
    case 2: r = ff_mul(a[i], b[i]); break;
    case 3: r = ff_dot2<false>(a[i], b[i], c[i], d[i], 0); break;
    case 4: r = ff_dot2<true>(a[i], b[i], c[i], d[i], e[i]); break;
    case 5: r = ff_mul_pow2_var(a[i], (u32)b[i] & 31u); break;
    case 6: r = FfShiftDispatch<191>::run(a[i], shift); break;
    case 7: r = ff_reduce96(a[i], (u32)b[i]); break;
    case 8: r = (u64)(u32)ff_to_i32(ff_from_i32((i32)a[i])) | ((u64)(u32)ff_to_i32(a[i]) << 32); break;
    }
    out[i] = r;
}

// Test hook for the redundant-limb arithmetic (ff24.h, ntt1024_l4.h): csrc/l4_hook.h, arrays u32[count][4]
__global__ void k_l4_op(u32 *__restrict__ out, u32 *__restrict__ out2, const u32 *__restrict__ a,
                        const u32 *__restrict__ b, const u32 *__restrict__ c, int op, int shift, long count)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    l4_hook(out + 4 * i, out2 + 4 * i, a + 4 * i, b ? b + 4 * i : nullptr, c ? c + 4 * i : nullptr, op, shift);
}

// ------------------------------------------------------------------------------------------
// Host launchers
// ------------------------------------------------------------------------------------------

static constexpr size_t br_lds_bytes(int K, int waves) { return TABLE_LDS_BYTES + BR_PACE_BYTES + (size_t)waves * WAVE_BR_LDS_BYTES_K(K); }
static constexpr size_t br_lds_bytes(int K) { return br_lds_bytes(K, BR_WAVES_K(K)); }
static constexpr size_t brp_lds_bytes(int pairs) { return TABLE_LDS_BYTES + BR_PACE_BYTES + (size_t)pairs * PAIR_LDS_BYTES; }
static constexpr size_t brf_lds_bytes(int waves) { return FTABLE_LDS_BYTES + BR_PACE_BYTES + BRF_BASE_PAD + (size_t)waves * WAVE_BRF_LDS_BYTES; }
static_assert(brf_lds_bytes(BR_WAVES_PER_BLOCK) <= 160 * 1024, "LDS budget of k_bootstrap_fft");
static const size_t kBrfLds = brf_lds_bytes(BR_WAVES_PER_BLOCK);

static const size_t kFftLds = FTABLE_LDS_BYTES + (size_t)FFT_WAVES_PER_BLOCK * WAVE_FXBUF_BYTES;

// per-device one-time setup: the fused kernels use up to ~156 KiB of dynamic LDS (> the 64 KiB default cap)
// *num_cus receives the CU count of the CURRENT device (kept per context: contexts on devices of
// different sizes may coexist in one process)
hipError_t kernels_init_device(int *num_cus, char *arch_name, size_t arch_len)
{
    int dev = 0;
    hipDeviceProp_t prop;
    hipError_t pe = hipGetDevice(&dev);
    if (pe == hipSuccess) pe = hipGetDeviceProperties(&prop, dev);
    if (pe != hipSuccess) return pe;
    if (prop.multiProcessorCount <= 0) return hipErrorInvalidDevice;
    *num_cus = prop.multiProcessorCount;
    if (arch_name && arch_len) {
        strncpy(arch_name, prop.gcnArchName, arch_len - 1);
        arch_name[arch_len - 1] = 0;
    }
    hipError_t e = hipFuncSetAttribute((const void *)k_bootstrap<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)br_lds_bytes(1));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)k_bootstrap<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)br_lds_bytes(2));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)k_bootstrap_ring_k2, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(TABLE_LDS_BYTES + BR_PACE_BYTES + BRR_MAX_TEAMS * RING2_LDS_BYTES));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)k_bootstrap_k2_roomy, hipFuncAttributeMaxDynamicSharedMemorySize, (int)br_lds_bytes(2, BR_K2_ROOMY_WAVES));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)k_keyswitch_a, hipFuncAttributeMaxDynamicSharedMemorySize, (int)KS_LDS_BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)k_bootstrap_fft_team, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TEAMF_LDS_BYTES);
    if (e != hipSuccess) return e;
    e = team8_init();
    if (e != hipSuccess) return e;
    e = team_init();
    if (e != hipSuccess) return e;
    e = xfft_init();
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)k_bootstrap_pair, hipFuncAttributeMaxDynamicSharedMemorySize, (int)brp_lds_bytes(BRP_MAX_PAIRS));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)k_bootstrap_fft_team_k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TEAMF2_LDS_BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)k_bootstrap_fft_ring_k2, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(FTABLE_LDS_BYTES + BR_PACE_BYTES + BRFR_MAX_TEAMS * FRING2_LDS_BYTES));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)k_bootstrap_fft_pair, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(FTABLE_LDS_BYTES + BR_PACE_BYTES + BRFP_MAX_PAIRS * FPAIR_LDS_BYTES));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)k_blind_rotate_accum<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)br_lds_bytes(1));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)k_blind_rotate_accum<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)br_lds_bytes(2));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)k_bootstrap_fft, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBrfLds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)k_blind_rotate_accum_fft, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBrfLds);
    if (e != hipSuccess) return e;
    const int brf2 = (int)(FTABLE_LDS_BYTES + (size_t)BRF2_WAVES * WAVE_BRF2_LDS_BYTES);
    e = hipFuncSetAttribute((const void *)k_bootstrap_fft_k2, hipFuncAttributeMaxDynamicSharedMemorySize, brf2);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void *)k_blind_rotate_accum_fft_k2, hipFuncAttributeMaxDynamicSharedMemorySize, brf2);
}


hipError_t launch_ff_op(u64 *out, const u64 *a, const u64 *b, const u64 *c, const u64 *d, const u64 *e, int op,
                        int shift, long count, hipStream_t stream)
{
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(k_ff_op, dim3(blocks_for(count, 256)), dim3(256), 0, stream, out, a, b, c, d, e, op, shift, count);
    return hipGetLastError();
}

hipError_t launch_l4_op(u32 *out, u32 *out2, const u32 *a, const u32 *b, const u32 *c, int op, int shift, long count,
                        hipStream_t stream)
{
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(k_l4_op, dim3(blocks_for(count, 64)), dim3(64), 0, stream, out, out2, a, b, c, op, shift, count);
    return hipGetLastError();
}

// P restricted to the global bits [g0, g1): the same kernels run on it (sources and outputs moved to g0)
static BrLaunch br_sub_launch(const BrLaunch &P, long g0, long g1)
{
    auto moved = [](BrSource s, long bits) {
        if (s.p) { s.a += bits * s.a_stride; s.b += bits * s.b_stride; }
        return s;
    };
    BrLaunch Q = P;
    if (g0 >= P.bits_per_job) {          // entirely inside the second job
        Q.job[0] = P.job[1];
        Q.job[0].s0 = moved(P.job[1].s0, g0 - P.bits_per_job);
        Q.job[0].s1 = moved(P.job[1].s1, g0 - P.bits_per_job);
        Q.bits_per_job = g1 - g0;
    } else {
        Q.job[0].s0 = moved(P.job[0].s0, g0);
        Q.job[0].s1 = moved(P.job[0].s1, g0);
        Q.bits_per_job = P.bits_per_job - g0;
    }
    Q.nbits_total = g1 - g0;
    Q.out_a = P.out_a + g0 * P.out_a_stride;
    Q.out_b = P.out_b + g0 * P.out_b_stride;
    return Q;
}

// Measured switch points (MI355X, 256 CUs: profiles/r0*_latency_*.json, r04f_small_batch_kernel_stats.csv,
// r02q_k2_timing.txt): team kernels up to 1 x CUs bits (one bit per CU at a time), NTT pair kernel up to 4 x CUs, FFT
// pair kernel up to 3 x CUs, matrix-core keyswitch above 2 x CUs, k = 2 round-time ratio 1.56.  They are all "bits per
// CU" figures, so a part with another CU count starts from the same ratios; a part that is measured gets its own row.
struct TuningRow { const char *arch; int cus; double team, pair_ntt, pair_fft, ks_mfma; int k2_ratio_pct; };
static const TuningRow kTuningTable[] = {
    {"gfx950", 256, 1.0, (double)BRP_MAX_PAIRS, (double)BRFP_MAX_PAIRS, 2.0, 156},
};

BrTuning br_tuning_for(const char *arch_name, int num_cus)
{
    const TuningRow *row = &kTuningTable[0];      // unknown part: the ratios of the measured one
    int measured = 0;
    for (const TuningRow &r : kTuningTable) {
        // gcnArchName carries feature suffixes ("gfx950:sramecc+:xnack-")
        const size_t n = strlen(r.arch);
        if (arch_name && strncmp(arch_name, r.arch, n) == 0 && (arch_name[n] == 0 || arch_name[n] == ':') && r.cus == num_cus) {
            row = &r;
            measured = 1;
        }
    }
    BrTuning T;
    T.num_cus = num_cus;
    T.team_max_bits = (long)(row->team * num_cus);
    T.team_max_bits_fft = (long)(row->team * num_cus);
    T.pair_max_bits_ntt = (long)(row->pair_ntt * num_cus);
    T.pair_max_bits_fft = (long)(row->pair_fft * num_cus);
    T.ring_k2 = 1;
    T.k2_roomy_ratio_pct = row->k2_ratio_pct;
    T.ks_mfma_min_bits = (long)(row->ks_mfma * num_cus);
    T.measured = measured;
    return T;
}

#ifndef BRX_QUAD_CUS
#define BRX_QUAD_CUS 2
#endif
hipError_t launch_bootstrap(const BrLaunch &P, int transform, int mask_size, const BrTuning &T, hipStream_t stream)
{
    if (P.nbits_total == 0) return hipSuccess;
    // the switches (BrTuning, kernels.h): NTT: team kernel up to team_max_bits, then the pair (k = 1) / ring (k = 2, any
    // size) kernel; FFT k = 1: pair kernel up to pair_max_bits_fft (team kernel up to team_max_bits_fft when the pair
    // switch is 0), k = 2: team kernel, then the ring kernel
    const int num_cus = T.num_cus;
    if (transform == BR_TRANSFORM_XFFT) {
        if (mask_size == 1 && !P.park) return hipErrorInvalidValue;
        // exact engine: four waves per bit up to BRX_QUAD_CUS x CUs bits (the team switch at 0 turns it off), then one wave per bit
        const long quad_max = T.team_max_bits > 0 ? (long)BRX_QUAD_CUS * num_cus : 0;
        if (mask_size == 1 && quad_max > 0 && P.nbits_total > quad_max && P.nbits_total <= 2 * quad_max) {
            // up to twice that: two rounds of the quad kernel (3.3 ms + 2.0 ... 3.3 ms) are ahead of one round of lone
            // waves on the one-wave kernel (7.2 ms)
            hipError_t e = launch_bootstrap_xfft(br_sub_launch(P, 0, quad_max), mask_size, num_cus, quad_max, stream);
            if (e != hipSuccess) return e;
            return launch_bootstrap_xfft(br_sub_launch(P, quad_max, P.nbits_total), mask_size, num_cus, quad_max, stream);
        }
        // larger ragged batches: whole rounds of 8 x CUs bits on the one-wave kernel, a tail of up to 2 x CUs bits on the quad
        // kernel (2.0 ... 3.3 ms instead of another 5 ... 7 ms round of thinly filled work-groups: 2060 bits 15.9 -> 12.8 ms;
        // with a tail of 3 ... 4 x CUs bits the single launch is ahead: 3000 bits 16.8 against 17.5 ms)
        const long round_bits = (long)BR_WAVES_PER_BLOCK * num_cus;
        const long tail = P.nbits_total % round_bits, head = P.nbits_total - tail;
        if (mask_size == 1 && quad_max > 0 && head > 0 && tail > 0 && tail <= quad_max) {
            hipError_t e = launch_bootstrap_xfft(br_sub_launch(P, 0, head), mask_size, num_cus, quad_max, stream);
            if (e != hipSuccess) return e;
            return launch_bootstrap(br_sub_launch(P, head, P.nbits_total), transform, mask_size, T, stream);
        }
        return launch_bootstrap_xfft(P, mask_size, num_cus, quad_max, stream);
    }
    const long team_max_bits = transform == 0 ? T.team_max_bits : 2 * T.team_max_bits_fft;    // (FFT: halved again below)
    const long pair_max_bits = transform == 0 ? (mask_size == 1 ? T.pair_max_bits_ntt : (long)T.ring_k2)
                                              : (mask_size == 1 ? T.pair_max_bits_fft : (long)T.ring_k2);
    if (transform == 0 && mask_size == 1 && P.nbits_total <= team_max_bits && P.bk_half != nullptr) {
        // smallest batches: 8 waves per bit (two per digit transform, half rings), one bit per CU at a time
        hipError_t e8 = launch_team8(P, stream);
        if (e8 != hipSuccess) return e8;
    } else if (transform == 0 && mask_size == 1 && P.nbits_total <= team_max_bits) {
        // small batch: 4 waves per bit, one bit per CU at a time
        hipError_t e4 = launch_team(P, stream);
        if (e4 != hipSuccess) return e4;
    } else if (transform == 0 && mask_size == 1 && P.nbits_total <= pair_max_bits) {
        // medium batch: 2 waves per bit; as few pairs per work-group as still give one round
        long pairs = (P.nbits_total + num_cus - 1) / num_cus;
        if (pairs > BRP_MAX_PAIRS) pairs = BRP_MAX_PAIRS;
        hipLaunchKernelGGL(k_bootstrap_pair, dim3(blocks_for(P.nbits_total, (int)pairs)), dim3(128 * (unsigned)pairs),
                           brp_lds_bytes((int)pairs), stream, P);
    } else if (transform == 0 && mask_size == 1) {
        // large batch: whole rounds of 8 waves per CU; a last partial round that the small- or medium-batch kernel
        // finishes sooner than a round of the wave kernel (21 ms) goes to that kernel in a second launch
        const long round_bits = (long)BR_WAVES_K(1) * num_cus;
        const long tail = P.nbits_total % round_bits, head = P.nbits_total - tail;
        const long tail_limit = pair_max_bits > team_max_bits ? pair_max_bits : team_max_bits;
        if (head > 0 && tail > 0 && tail <= tail_limit) {
            const BrLaunch H = br_sub_launch(P, 0, head);
            hipLaunchKernelGGL(k_bootstrap<1>, dim3(blocks_for(head, BR_WAVES_K(1))), dim3(64 * BR_WAVES_K(1)), br_lds_bytes(1),
                               stream, H);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
            return launch_bootstrap(br_sub_launch(P, head, P.nbits_total), transform, mask_size, T, stream);
        }
        const int w = br_pick_waves(P.nbits_total, BR_WAVES_K(1), num_cus);
        hipLaunchKernelGGL(k_bootstrap<1>, dim3(blocks_for(P.nbits_total, w)), dim3(64 * w), br_lds_bytes(1, w), stream, P);
    } else if (transform == 0 && mask_size == 2 && P.nbits_total <= team_max_bits) {
        // small batch, k = 2: 3 waves per bit (one team per CU at a time)
        hipError_t e3 = launch_team_k2(P, stream);
        if (e3 != hipSuccess) return e3;
    } else if (transform == 0 && mask_size == 2 && pair_max_bits > 0) {
        // k = 2 beyond the team kernel: 3 waves per bit without the partial-sum buffer, 2 teams per work-group: rounds of
        // 2 x CUs bits in 12.3 ms (the wave kernels below: 4 x CUs in 27 ms / 6 x CUs in 43 ms); any non-zero pair limit
        // enables it
        const long teams = P.nbits_total > num_cus ? BRR_MAX_TEAMS : 1;
        hipLaunchKernelGGL(k_bootstrap_ring_k2, dim3(blocks_for(P.nbits_total, (int)teams)), dim3(192 * (unsigned)teams),
                           TABLE_LDS_BYTES + BR_PACE_BYTES + (size_t)teams * RING2_LDS_BYTES, stream, P);
    } else if (transform == 0 && mask_size == 2) {
        // rounds of 6 x CUs bits at 2 waves per SIMD (42 ms, spills) or rounds of 4 x CUs bits at 1 wave per SIMD
        // (27 ms, no spills): measured ratio 1.56
        const long r6 = (P.nbits_total + 6L * num_cus - 1) / (6L * num_cus), r4 = (P.nbits_total + 4L * num_cus - 1) / (4L * num_cus);
        if (100 * r4 < (long)T.k2_roomy_ratio_pct * r6) {
            const int w = br_pick_waves(P.nbits_total, BR_K2_ROOMY_WAVES, num_cus);
            hipLaunchKernelGGL(k_bootstrap_k2_roomy, dim3(blocks_for(P.nbits_total, w)), dim3(64 * w), br_lds_bytes(2, w), stream, P);
        } else {
            const int w = br_pick_waves(P.nbits_total, BR_WAVES_K(2), num_cus);
            hipLaunchKernelGGL(k_bootstrap<2>, dim3(blocks_for(P.nbits_total, w)), dim3(64 * w), br_lds_bytes(2, w), stream, P);
        }
    } else if (transform == 1 && mask_size == 1 && pair_max_bits > 0 && T.team_max_bits_fft > 0 && P.nbits_total <= num_cus) {
        // latency case, FFT: four waves per bit, one bit per CU (1.9 ms against 2.7 ms of the pair kernel); both the pair and
        // the team switch at 0 turn it off
        return launch_bootstrap_fft_quad(P, stream);
    } else if (transform == 1 && mask_size == 1 && pair_max_bits > 0 && P.nbits_total <= pair_max_bits) {
        // small and medium batches, FFT: 2 waves per bit, 1 to 3 pairs per work-group (ahead of the 4-wave team kernel at
        // every size: 2.9 vs 3.0 ms up to 1 x CUs bits, 3.1 ms up to 2 x CUs, 3.7 ms up to 3 x CUs); any non-zero pair
        // limit enables it
        long pairs = (P.nbits_total + num_cus - 1) / num_cus;
        if (pairs > BRFP_MAX_PAIRS) pairs = BRFP_MAX_PAIRS;
        hipLaunchKernelGGL(k_bootstrap_fft_pair, dim3(blocks_for(P.nbits_total, (int)pairs)), dim3(128 * (unsigned)pairs),
                           FTABLE_LDS_BYTES + BR_PACE_BYTES + (size_t)pairs * FPAIR_LDS_BYTES, stream, P);
    } else if (transform == 1 && mask_size == 1 && P.nbits_total <= team_max_bits / 2) {
        // (pair kernel switched off:) small batch, FFT: 4 waves per bit, one bit per CU
        hipLaunchKernelGGL(k_bootstrap_fft_team, dim3((unsigned)P.nbits_total), dim3(64 * BRT_WAVES), TEAMF_LDS_BYTES, stream, P);
    } else if (transform == 1 && mask_size == 1) {
        // whole rounds of 8 x CUs bits here; a tail of up to 2 x CUs bits, which the quad / pair kernels finish sooner (2.0 ... 2.9 ms)
        // than another round of thinly filled work-groups (4 ms), goes to them in a second launch (2060 bits 10.1 -> 8.7 ms)
        const long round_bits = (long)BR_WAVES_PER_BLOCK * num_cus;
        const long tail = P.nbits_total % round_bits, head = P.nbits_total - tail;
        if (head > 0 && tail > 0 && pair_max_bits > 0 && tail <= 2L * num_cus) {
            const BrLaunch H = br_sub_launch(P, 0, head);
            hipLaunchKernelGGL(k_bootstrap_fft, dim3(blocks_for(head, BR_WAVES_PER_BLOCK)), dim3(64 * BR_WAVES_PER_BLOCK),
                               brf_lds_bytes(BR_WAVES_PER_BLOCK), stream, H);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
            return launch_bootstrap(br_sub_launch(P, head, P.nbits_total), transform, mask_size, T, stream);
        }
        const int w = br_pick_waves(P.nbits_total, BR_WAVES_PER_BLOCK, num_cus);
        hipLaunchKernelGGL(k_bootstrap_fft, dim3(blocks_for(P.nbits_total, w)), dim3(64 * w), brf_lds_bytes(w), stream, P);
    } else if (transform == 1 && mask_size == 2 && pair_max_bits > 0 && T.team_max_bits_fft > 0 && P.nbits_total <= num_cus) {
        // latency case, FFT, k = 2: six waves per bit, one bit per CU; both the ring and the team switch at 0 turn it off
        return launch_bootstrap_fft_hex_k2(P, stream);
    } else if (transform == 1 && mask_size == 2 && P.nbits_total <= team_max_bits / 2) {
        // small batch, FFT, k = 2: 3 waves per bit
        hipLaunchKernelGGL(k_bootstrap_fft_team_k2, dim3((unsigned)P.nbits_total), dim3(64 * 3), TEAMF2_LDS_BYTES, stream, P);
    } else if (transform == 1 && mask_size == 2 && pair_max_bits > 0) {
        // k = 2, FFT, beyond the team kernel: 3 waves per bit without the partial-sum buffer, 2 teams per work-group: rounds
        // of 2 x CUs bits in 4.2 ms (one wave per bit: 4 x CUs in 9.3 ms); any non-zero pair limit enables it
        const long teams = P.nbits_total > num_cus ? BRFR_MAX_TEAMS : 1;
        hipLaunchKernelGGL(k_bootstrap_fft_ring_k2, dim3(blocks_for(P.nbits_total, (int)teams)), dim3(192 * (unsigned)teams),
                           FTABLE_LDS_BYTES + BR_PACE_BYTES + (size_t)teams * FRING2_LDS_BYTES, stream, P);
    } else if (transform == 1 && mask_size == 2) {
        const int w = br_pick_waves(P.nbits_total, BRF2_WAVES, num_cus);
        hipLaunchKernelGGL(k_bootstrap_fft_k2, dim3(blocks_for(P.nbits_total, w)), dim3(64 * w),
                           FTABLE_LDS_BYTES + (size_t)w * WAVE_BRF2_LDS_BYTES, stream, P);
    } else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_blind_rotate_accum(i32 *accum, const void *bk, const i32 *bara, long bara_stride, int row0,
                                     int n_rows, int external_mul_only, long batch, const void *tw_a,
                                     const void *tw_b, int transform, int mask_size, hipStream_t stream)
{
    if (batch == 0) return hipSuccess;
    if (transform == 0 && mask_size == 1)
        hipLaunchKernelGGL(k_blind_rotate_accum<1>, dim3(blocks_for(batch, BR_WAVES_K(1))), dim3(64 * BR_WAVES_K(1)),
                           br_lds_bytes(1), stream, accum, (const u64 *)bk, bara, bara_stride, row0, n_rows,
                           external_mul_only, batch, (const u64 *)tw_a, (const u64 *)tw_b);
    else if (transform == 0 && mask_size == 2)
        hipLaunchKernelGGL(k_blind_rotate_accum<2>, dim3(blocks_for(batch, BR_WAVES_K(2))), dim3(64 * BR_WAVES_K(2)),
                           br_lds_bytes(2), stream, accum, (const u64 *)bk, bara, bara_stride, row0, n_rows,
                           external_mul_only, batch, (const u64 *)tw_a, (const u64 *)tw_b);
    else if (transform == 1 && mask_size == 1)
        hipLaunchKernelGGL(k_blind_rotate_accum_fft, dim3(blocks_for(batch, BR_WAVES_PER_BLOCK)), dim3(BR_BLOCK_THREADS),
                           kBrfLds, stream, accum, (const cplx *)bk, bara, bara_stride, row0, n_rows,
                           external_mul_only, batch, (const cplx *)tw_a, (const cplx *)tw_b);
    else if (transform == 1 && mask_size == 2)
        hipLaunchKernelGGL(k_blind_rotate_accum_fft_k2, dim3(blocks_for(batch, BRF2_WAVES)), dim3(64 * BRF2_WAVES),
                           FTABLE_LDS_BYTES + (size_t)BRF2_WAVES * WAVE_BRF2_LDS_BYTES, stream, accum, (const cplx *)bk,
                           bara, bara_stride, row0, n_rows, external_mul_only, batch, (const cplx *)tw_a,
                           (const cplx *)tw_b);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_fft_forward(cplx *out, const i32 *in, long batch, const cplx *tw1, const cplx *tw2, hipStream_t stream)
{
    if (batch == 0) return hipSuccess;
    hipLaunchKernelGGL(k_fft_forward<false>, dim3(blocks_for(batch, FFT_WAVES_PER_BLOCK)), dim3(FFT_BLOCK_THREADS),
                       kFftLds, stream, out, in, batch, tw1, tw2);
    return hipGetLastError();
}

hipError_t launch_fft_inverse(i32 *out, const cplx *in, long batch, const cplx *tw1, const cplx *tw2, hipStream_t stream)
{
    if (batch == 0) return hipSuccess;
    hipLaunchKernelGGL(k_fft_inverse, dim3(blocks_for(batch, FFT_WAVES_PER_BLOCK)), dim3(FFT_BLOCK_THREADS), kFftLds,
                       stream, out, in, batch, tw1, tw2);
    return hipGetLastError();
}

hipError_t launch_bkf_from_coeffs(cplx *out, const i32 *in, long polys, const cplx *tw1, const cplx *tw2,
                                  hipStream_t stream)
{
    if (polys == 0) return hipSuccess;
    hipLaunchKernelGGL(k_fft_forward<true>, dim3(blocks_for(polys, FFT_WAVES_PER_BLOCK)), dim3(FFT_BLOCK_THREADS),
                       kFftLds, stream, out, in, polys, tw1, tw2);
    return hipGetLastError();
}

hipError_t launch_bkf_permute(cplx *out, const cplx *in, long polys, int to_reference, hipStream_t stream)
{
    if (polys == 0) return hipSuccess;
    hipLaunchKernelGGL(k_bkf_permute, dim3(blocks_for(polys * 512, 256)), dim3(256), 0, stream, out, in, polys,
                       to_reference);
    return hipGetLastError();
}

static const size_t kNttLds = TABLE_LDS_BYTES + (size_t)NTT_WAVES_PER_BLOCK * WAVE_XBUF_BYTES;

hipError_t launch_ntt_forward(u64 *out, const void *in, int mode, long batch, const u64 *tw1f, const u64 *tw1i,
                              hipStream_t stream)
{
    if (batch == 0) return hipSuccess;
    const dim3 grid(blocks_for(batch, NTT_WAVES_PER_BLOCK)), block(NTT_BLOCK_THREADS);
    if (mode == 0) hipLaunchKernelGGL(k_ntt_forward<0>, grid, block, kNttLds, stream, out, in, batch, tw1f, tw1i);
    else hipLaunchKernelGGL(k_ntt_forward<1>, grid, block, kNttLds, stream, out, in, batch, tw1f, tw1i);
    return hipGetLastError();
}

hipError_t launch_ntt_inverse(void *out, const u64 *in, int mode, long batch, const u64 *tw1f, const u64 *tw1i,
                              hipStream_t stream)
{
    if (batch == 0) return hipSuccess;
    const dim3 grid(blocks_for(batch, NTT_WAVES_PER_BLOCK)), block(NTT_BLOCK_THREADS);
    if (mode == 0) hipLaunchKernelGGL(k_ntt_inverse<0>, grid, block, kNttLds, stream, out, in, batch, tw1f, tw1i);
    else hipLaunchKernelGGL(k_ntt_inverse<1>, grid, block, kNttLds, stream, out, in, batch, tw1f, tw1i);
    return hipGetLastError();
}

hipError_t launch_bk_to_coeffs(i32 *out, const u64 *bk_wave, long polys, const u64 *tw1f, const u64 *tw1i, int *not_int32,
                               hipStream_t stream)
{
    if (polys == 0) return hipSuccess;
    hipLaunchKernelGGL(k_bk_to_coeffs, dim3(blocks_for(polys, NTT_WAVES_PER_BLOCK)), dim3(NTT_BLOCK_THREADS), kNttLds, stream,
                       out, bk_wave, polys, tw1f, tw1i, not_int32);
    return hipGetLastError();
}

hipError_t launch_poly_mul_strided(i32 *out, long out_stride, const i32 *x, long x_stride, const i32 *y,
                                   const i32 *base, long base_stride, long batch, const u64 *tw1f, const u64 *tw1i,
                                   hipStream_t stream)
{
    if (batch == 0) return hipSuccess;
    hipLaunchKernelGGL(k_poly_mul, dim3(blocks_for(batch, NTT_WAVES_PER_BLOCK)), dim3(NTT_BLOCK_THREADS), kNttLds,
                       stream, out, out_stride, x, x_stride, y, base, base_stride, batch, 1L, tw1f, tw1i);
    return hipGetLastError();
}

hipError_t launch_poly_mul(i32 *out, const i32 *x, const i32 *y, const i32 *base, long out_stride, long batch,
                           long y_batch, const u64 *tw1f, const u64 *tw1i, hipStream_t stream)
{
    if (batch == 0) return hipSuccess;
    hipLaunchKernelGGL(k_poly_mul, dim3(blocks_for(batch, NTT_WAVES_PER_BLOCK)), dim3(NTT_BLOCK_THREADS), kNttLds,
                       stream, out, out_stride, x, 1024L, y, base, 1024L, batch, y_batch, tw1f, tw1i);
    return hipGetLastError();
}

hipError_t launch_bk_from_reference(u64 *out, const u64 *in, long polys, hipStream_t stream)
{
    hipLaunchKernelGGL(k_bk_from_reference, dim3(blocks_for(polys * 1024, 256)), dim3(256), 0, stream, out, in, polys);
    return hipGetLastError();
}

hipError_t launch_tgsw_decompose(i32 *result, const i32 *sample, long polys, hipStream_t stream)
{
    if (polys == 0) return hipSuccess;
    hipLaunchKernelGGL(k_tgsw_decompose, dim3(blocks_for(polys * 1024, 256)), dim3(256), 0, stream, result, sample, polys);
    return hipGetLastError();
}

hipError_t launch_tgsw_mac(u64 *result, const u64 *sample, const u64 *bk, int bk_row, long batch, int mask_size,
                           hipStream_t stream)
{
    if (batch == 0) return hipSuccess;
    const dim3 grid(blocks_for(batch * (mask_size + 1) * 1024, 256));
    if (mask_size == 1) hipLaunchKernelGGL(k_tgsw_mac<1>, grid, dim3(256), 0, stream, result, sample, bk, bk_row, batch);
    else hipLaunchKernelGGL(k_tgsw_mac<2>, grid, dim3(256), 0, stream, result, sample, bk, bk_row, batch);
    return hipGetLastError();
}

hipError_t launch_bk_to_half(u64 *out, const u64 *in, long polys, hipStream_t stream)
{
    hipLaunchKernelGGL(k_bk_to_half, dim3(blocks_for(polys * 1024, 256)), dim3(256), 0, stream, out, in, polys);
    return hipGetLastError();
}

hipError_t launch_bk_to_reference(u64 *out, const u64 *in, long polys, hipStream_t stream)
{
    hipLaunchKernelGGL(k_bk_to_reference, dim3(blocks_for(polys * 1024, 256)), dim3(256), 0, stream, out, in, polys);
    return hipGetLastError();
}

hipError_t launch_bk_from_coeffs(u64 *out, const i32 *in, long polys, const u64 *tw1f, const u64 *tw1i,
                                 hipStream_t stream)
{
    hipLaunchKernelGGL(k_bk_from_coeffs, dim3(blocks_for(polys, NTT_WAVES_PER_BLOCK)), dim3(NTT_BLOCK_THREADS),
                       kNttLds, stream, out, in, polys, tw1f, tw1i);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// K2 on the matrix cores.  The keyswitch sum is a matrix product: out[bit][c] = - sum_K A[bit][K] B[K][c] with
// K = (j, k, d), A one-hot in d (the digit k of coefficient j of the bit; digit 0 selects the all-zero row) and B the
// keyswitch key.  MFMA multiplies int8, so B is split into four signed byte planes, B = sum_p B_p 2^(8p) (balanced
// digits: exact mod 2^32), each |plane sum| <= 8192 x 128 < 2^21 in the int32 accumulators, and the planes are
// recombined with shifts at the end: bit-exact, like every integer reformulation of this sum.
//   v_mfma_i32_16x16x64_i8: lane l holds 16 bytes of A for row l % 16 and of B for column l % 16, both for the
//   K-slots 16 (l / 16) .. + 15; one instruction = 2 coefficients j: slot = 32 (j & 1) + 4 k + d.  The A operand of a
//   lane is therefore 4 dwords, each 1 << (8 digit) -- built from the source coefficient with 3 instructions -- and
//   the B operand one 16-byte load from the plane array laid out for it (k_ks_planes).
// One wave owns 64 bits x 32 columns x all four planes (128 accumulator registers), a work-group of 4 waves 256 bits of
// the same columns (the B loads of its waves hit L1).  4096 bits: 256 work-groups x 4 waves, 32 MFMAs per step of 2
// coefficients.
// ------------------------------------------------------------------------------------------
typedef int ksm_v4i __attribute__((ext_vector_type(4)));

// planes[p][j][h][c][16]: byte 4 m + d = plane p of KS[j][4 h + m][d][c] (d = 0 and c >= n: 0)
__global__ void k_ks_planes(signed char *__restrict__ planes, const i32 *__restrict__ ks_a3, int input_size, int n)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;          // (j, h, c)
    const long total = (long)input_size * 2 * KSM_COLS;
    if (idx >= total) return;
    const int c = (int)(idx % KSM_COLS), h = (int)((idx / KSM_COLS) & 1);
    const long j = idx / (2 * KSM_COLS);
    signed char b[4][16];
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int d = 0; d < 4; d++) {
            u32 v = 0;
            if (d > 0 && c < n) v = (u32)ks_a3[((j * 8 + 4 * h + m) * 3 + (d - 1)) * n + c];
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const signed char s8 = (signed char)(v & 0xFFu);     // balanced byte: value in [-128, 127]
                b[p][4 * m + d] = s8;
                v = (u32)((i32)(v - (u32)(i32)s8) >> 8);            // (mod 2^32 throughout: the carry out of plane 3 is dropped)
            }
        }
#pragma unroll
    for (int p = 0; p < 4; p++) {
        ksm_v4i *dst = (ksm_v4i *)(planes + ((((long)p * input_size + j) * 2 + h) * KSM_COLS + c) * 16);
        *dst = *(const ksm_v4i *)b[p];
    }
}

// The 8 base-4 digits of a source coefficient are the top 16 bits of a' + 2^15 (lwe_cpu.py:76).  k_keyswitch_mfma
// wants them by (coefficient, bit) -- 64 consecutive bits of one coefficient are one 128-byte line -- while the
// extracted samples lie by (bit, coefficient): this pre-pass adds the two sources of a MUX, takes the top half and
// transposes through LDS (4096 bits: 16 MB in, 8 MB out, a few microseconds).
__global__ __launch_bounds__(256) void k_ks_digits_t(unsigned short *__restrict__ out, KsLaunch P, long nbits_pad)
{
    __shared__ unsigned short tile[64][66];                 // [j][bit], padded
    const long bit0 = (long)blockIdx.x * 64;
    const int j0 = blockIdx.y * 64;
    const int tj = threadIdx.x & 63, tb = threadIdx.x >> 6;  // read: 64 coefficients of a row are contiguous
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const int b = tb + 4 * q;
        u32 v = 0;
        if (bit0 + b < P.nbits) {
            v = (u32)P.src1_a[(bit0 + b) * P.src1_stride + j0 + tj];
            if (P.src2_a) v += (u32)P.src2_a[(bit0 + b) * P.src2_stride + j0 + tj];
        }
        tile[tj][b] = (unsigned short)((v + (1u << 15)) >> 16);
    }
    __syncthreads();
    const int ob = threadIdx.x & 63, oj = threadIdx.x >> 6;  // write: 64 bits of a coefficient are contiguous
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const int j = oj + 4 * q;
        out[(long)(j0 + j) * nbits_pad + bit0 + ob] = tile[j][ob];
    }
}

#define NUFHE_KS_MIN_INPUT 1024     /* input sizes are multiples of 1024 (mask_size x N) */
#define KSM_WAVE_BITS 64
#define KSM_WAVE_COLS 32
#define KSM_DEPTH 4                 /* 2: L2 latency exposed (K2 0.35 ms); 8: spills (0.43 ms) */
static_assert((NUFHE_KS_MIN_INPUT / 2) % KSM_DEPTH == 0, "steps must be a multiple of the prefetch depth");
__global__ __launch_bounds__(256, 1) void k_keyswitch_mfma(KsLaunch P)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const long bit0 = ((long)blockIdx.x * 4 + wave) * KSM_WAVE_BITS;
    if (bit0 >= P.nbits) return;
    const int col0 = blockIdx.y * KSM_WAVE_COLS;
    const int jj = g >> 1, h = g & 1;                 // this lane group's coefficient within the step, half of the digits
    ksm_v4i acc[4][4][2];
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
#pragma unroll
            for (int nt = 0; nt < 2; nt++) acc[p][mt][nt] = ksm_v4i{0, 0, 0, 0};
    // digits of this lane's rows (bits bit0 + 16 mt + r) for coefficient 2 s + jj of step s, from the transposed array
    // (k_ks_digits_t): the 64 bits of the wave are one cache line per coefficient
    const long nbits_pad = (P.nbits + 63) & ~63L;
    const unsigned short *dsrc = P.digits_t + (long)jj * nbits_pad + bit0 + r;
    const long plane_stride = (long)P.input_size * 2 * KSM_COLS * 16;
    const signed char *bbase = P.ks_planes + (((long)jj * 2 + h) * KSM_COLS + col0 + r) * 16;
    auto load_a = [&](u32 (&aj)[4], int s) {
#pragma unroll
        for (int mt = 0; mt < 4; mt++) aj[mt] = dsrc[(long)(2 * s) * nbits_pad + 16 * mt];
    };
    auto load_b = [&](ksm_v4i (&bf)[4][2], int s) {
        const signed char *q = bbase + (long)s * (2 * 2 * KSM_COLS * 16);
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int nt = 0; nt < 2; nt++) bf[p][nt] = *(const ksm_v4i *)(q + p * plane_stride + nt * 16 * 16);
    };
    const int steps = P.input_size / 2;
    // one wave per SIMD (128 accumulator registers): the loads run KSM_DEPTH steps ahead of the multiplications in a ring
    // of register slots
    constexpr int D = KSM_DEPTH;
    u32 aj[D][4];
    ksm_v4i bf[D][4][2];
#pragma unroll
    for (int u = 0; u < D; u++) {
        load_a(aj[u], u);
        load_b(bf[u], u);
    }
    for (int s0 = 0; s0 < steps; s0 += D) {
#pragma unroll
        for (int u = 0; u < D; u++) {
            const int s = s0 + u;
            ksm_v4i af[4];
#pragma unroll
            for (int mt = 0; mt < 4; mt++)
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const u32 dgt = (aj[u][mt] >> (14 - 2 * (4 * h + m))) & 3u;     // digit k = 4 h + m (lwe_cpu.py:76)
                    af[mt][m] = (int)(1u << (8 * dgt));
                }
            ksm_v4i bcur[4][2];
#pragma unroll
            for (int p = 0; p < 4; p++)
#pragma unroll
                for (int nt = 0; nt < 2; nt++) bcur[p][nt] = bf[u][p][nt];
            if (s + D < steps) {
                load_a(aj[u], s + D);
                load_b(bf[u], s + D);
            }
#pragma unroll
            for (int p = 0; p < 4; p++)
#pragma unroll
                for (int mt = 0; mt < 4; mt++)
#pragma unroll
                    for (int nt = 0; nt < 2; nt++)
                        acc[p][mt][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[mt], bcur[p][nt], acc[p][mt][nt], 0, 0, 0);
        }
    }
    // C layout: column l % 16, rows 4 (l / 16) + i
#pragma unroll
    for (int mt = 0; mt < 4; mt++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const long bit = bit0 + 16 * mt + 4 * g + i;
                const int c = col0 + 16 * nt + r;
                if (bit < P.nbits && c < P.n) {
                    const u32 total = (u32)acc[0][mt][nt][i] + ((u32)acc[1][mt][nt][i] << 8) + ((u32)acc[2][mt][nt][i] << 16) +
                                      ((u32)acc[3][mt][nt][i] << 24);
                    P.acc[bit * P.n + c] = 0u - total;
                }
            }
}

hipError_t launch_ks_planes(signed char *planes, const i32 *ks_a3, int input_size, int n, hipStream_t stream)
{
    if (n > KSM_COLS) return hipErrorInvalidValue;
    const long total = (long)input_size * 2 * KSM_COLS;
    hipLaunchKernelGGL(k_ks_planes, dim3(blocks_for(total, 256)), dim3(256), 0, stream, planes, ks_a3, input_size, n);
    return hipGetLastError();
}

hipError_t launch_keyswitch(const KsLaunch &P, const KsFinal &F, hipStream_t stream)
{
    if (P.nbits == 0) return hipSuccess;
    hipError_t e;
    if (P.ks_planes) {
        // every accumulator word is written exactly once: no zero fill, no atomics
        const long nbits_pad = (P.nbits + 63) & ~63L;
        hipLaunchKernelGGL(k_ks_digits_t, dim3(blocks_for(P.nbits, 64), P.input_size / 64), dim3(256), 0, stream, P.digits_t, P,
                           nbits_pad);
        const dim3 grid(blocks_for(P.nbits, 4 * KSM_WAVE_BITS), (KSM_COLS + KSM_WAVE_COLS - 1) / KSM_WAVE_COLS);
        hipLaunchKernelGGL(k_keyswitch_mfma, grid, dim3(256), 0, stream, P);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_keyswitch_finalize, dim3((unsigned)P.nbits), dim3(256), 0, stream, F);
        return hipGetLastError();
    }
    e = hipMemsetAsync(P.acc, 0, (size_t)P.nbits * P.n * sizeof(u32), stream);
    if (e != hipSuccess) return e;
    const dim3 grid(blocks_for(P.nbits, KS_TILE_BITS), P.input_size / P.j_per_block);
    hipLaunchKernelGGL(k_keyswitch_a, grid, dim3(KS_BLOCK_THREADS), KS_LDS_BYTES, stream, P);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_keyswitch_finalize, dim3((unsigned)P.nbits), dim3(256), 0, stream, F);
    return hipGetLastError();
}

hipError_t launch_write_table(void *d_dst, const void *h_src, size_t bytes, hipStream_t stream)
{
    // (device buffers are sized in multiples of 16 bytes and the tables' records are 16-byte multiples)
    for (size_t off = 0; off < bytes; off += TABLE_CHUNK_BYTES) {
        const size_t n = bytes - off < TABLE_CHUNK_BYTES ? bytes - off : TABLE_CHUNK_BYTES;
        TableChunk c;
        memset(&c, 0, sizeof(c));
        memcpy(&c, (const char *)h_src + off, n);
        hipLaunchKernelGGL(k_write_table, dim3(1), dim3(64), 0, stream, (uint4 *)((char *)d_dst + off), c, (int)((n + 15) / 16));
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_batch_combine(i32 *out_a, i32 *out_b, const BatchRot *rots, int n_rots, long rows, int n, hipStream_t stream)
{
    if (rows == 0) return hipSuccess;
    hipLaunchKernelGGL(k_batch_combine, dim3((unsigned)rows), dim3(256), 0, stream, out_a, out_b, rots, n_rots, n);
    return hipGetLastError();
}

hipError_t launch_batch_mux_fold(i32 *ext_a, i32 *ext_b, const BatchOut *outs, int n_outs, long out_bits, int ext, i32 mu,
                                 hipStream_t stream)
{
    if (out_bits == 0) return hipSuccess;
    hipLaunchKernelGGL(k_batch_mux_fold, dim3((unsigned)out_bits), dim3(256), 0, stream, ext_a, ext_b, outs, n_outs, ext, mu);
    return hipGetLastError();
}

hipError_t launch_lwe_linear(const LweView &res, const LweView &src, i32 p, int add_result, long nbits, int size,
                             hipStream_t stream)
{
    if (nbits == 0) return hipSuccess;
    hipLaunchKernelGGL(k_lwe_linear, dim3(blocks_for(nbits * (size + 1), 256)), dim3(256), 0, stream, res, src, p,
                       add_result, nbits, size);
    return hipGetLastError();
}

hipError_t launch_lwe_trivial_const(const LweView &res, i32 mu, long nbits, int size, hipStream_t stream)
{
    if (nbits == 0) return hipSuccess;
    hipLaunchKernelGGL(k_lwe_trivial_const, dim3(blocks_for(nbits * (size + 1), 256)), dim3(256), 0, stream, res, mu,
                       nbits, size);
    return hipGetLastError();
}

hipError_t launch_lwe_phase(i32 *out, long out_stride, const i32 *a, long a_stride, const i32 *base, long base_stride,
                            const i32 *key, i32 sign, long count, int n, hipStream_t stream)
{
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(k_lwe_phase, dim3(blocks_for(count, 4)), dim3(256), 0, stream, out, out_stride, a, a_stride,
                       base, base_stride, key, sign, count, n);
    return hipGetLastError();
}

hipError_t launch_ks_make(i32 *ks_b, float *ks_cv, const i32 *noises_a, const i32 *noises_b, const i32 *in_key,
                          const i32 *out_key, float variance, long rows, int n, hipStream_t stream)
{
    if (rows == 0) return hipSuccess;
    hipLaunchKernelGGL(k_ks_make, dim3(blocks_for(rows, 4)), dim3(256), 0, stream, ks_b, ks_cv, noises_a, noises_b,
                       in_key, out_key, variance, rows, n);
    return hipGetLastError();
}

hipError_t launch_ks_to_reference(i32 *out_a, const i32 *ks_a3, long groups, int n, hipStream_t stream)
{
    if (groups == 0) return hipSuccess;
    hipLaunchKernelGGL(k_ks_to_reference, dim3(blocks_for(groups * 4 * n, 256)), dim3(256), 0, stream, out_a, ks_a3,
                       groups, n);
    return hipGetLastError();
}

hipError_t launch_tgsw_add_message(i32 *tgsw, const i32 *messages, long count, int mask_size, hipStream_t stream)
{
    if (count == 0) return hipSuccess;
    const int k1 = mask_size + 1;
    hipLaunchKernelGGL(k_tgsw_add_message, dim3(blocks_for(count * k1 * 2, 256)), dim3(256), 0, stream, tgsw, messages,
                       count, k1);
    return hipGetLastError();
}

hipError_t launch_t32_to_phase(i32 *result, const i32 *phase, long count, u32 mspace, hipStream_t stream)
{
    if (count == 0) return hipSuccess;
    const u32 interv = (u32)((1ULL << 32) / mspace);
    hipLaunchKernelGGL(k_t32_to_phase, dim3(blocks_for(count, 256)), dim3(256), 0, stream, result, phase, count, interv);
    return hipGetLastError();
}

hipError_t launch_shift_tp(i32 *result, const i32 *source, const i32 *powers, long powers_stride, long powers_idx,
                           long batch, int polys, int minus_one, int invert_powers, hipStream_t stream)
{
    if (batch == 0) return hipSuccess;
    hipLaunchKernelGGL(k_shift_tp, dim3(blocks_for(batch * polys * 1024, 256)), dim3(256), 0, stream, result, source,
                       powers, powers_stride, powers_idx, batch, polys, minus_one, invert_powers);
    return hipGetLastError();
}

hipError_t launch_tlwe_extract(i32 *ra, i32 *rb, const i32 *tlwe, long batch, int mask_size, hipStream_t stream)
{
    if (batch == 0) return hipSuccess;
    hipLaunchKernelGGL(k_tlwe_extract, dim3(blocks_for(batch * mask_size * 1024, 256)), dim3(256), 0, stream, ra, rb,
                       tlwe, batch, mask_size);
    return hipGetLastError();
}

#if defined(BR_PROBE)
// variant builds only (blind_rotate.h, BR_PROBE): read and clear the segment tick counters
extern "C" int nufhe_probe_read(unsigned long long *out16)
{
    unsigned long long zero[16] = {0};
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_br_probe), sizeof(zero)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_br_probe), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
extern "C" int nufhe_probe_lifetimes(unsigned int *out, int count)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_br_probe_life), sizeof(unsigned) * count) == hipSuccess ? 0 : -1;
}
#endif
