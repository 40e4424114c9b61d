// kernels_team.hip -- the lone-wave team kernels of the smallest NTT batches that still use whole-ring transforms:
// k_bootstrap_team (k = 1, four waves per bit; the fallback of the half-ring kernel) and k_bootstrap_team_k2 (k = 2, three
// waves per bit).  A translation unit of their own because they are built with -DFF_MULWIDE_PLAIN (Makefile): with one wave
// per SIMD a dependent instruction costs 8 cycles instead of 4, and the carry-out form of the 64 x 64 product (ff.h, round 4)
// -- fewer instructions, but a longer chain with a VALU -> SGPR -> VALU hop -- is 1.4-3 % SLOWER here while the kernels with
// two paced waves per SIMD gain 2.5 % from it.
#include <hip/hip_runtime.h>

#include "blind_rotate.h"
#include "ff.h"
#include "kernels.h"
#include "ntt1024.h"

#define TABLE_LDS_BYTES (2 * 1024 * 8)
#define WAVE_XBUF_BYTES (NTT_XBUF_ELEMS * 8)
#define WAVE_BARA_BYTES (BR_MAX_LWE * 2)
#define BR_PACE_BYTES 128

extern __shared__ __attribute__((aligned(16))) unsigned char g_smem[];

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

// Small-batch variant: a team of 4 waves (one work-group) per bit, see blind_rotate.h (brt_*).
// LDS: tables | ACC 8 KiB | bara 1 KiB | partial sums 64 KiB | 4 exchange buffers
#define TEAM_LDS_BYTES (TABLE_LDS_BYTES + 2 * 1024 * 4 + WAVE_BARA_BYTES + BRT_PART_ELEMS * 8 + BRT_WAVES * WAVE_XBUF_BYTES)
__global__ __launch_bounds__(64 * BRT_WAVES, 1) void k_bootstrap_team(BrLaunch P)
{
    load_tables((const u64 *)P.tw_a, (const u64 *)P.tw_b);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long gbit = blockIdx.x;
    const int job = gbit >= P.bits_per_job ? 1 : 0;
    const long bit = gbit - (job ? P.bits_per_job : 0);
    unsigned char *base = g_smem + TABLE_LDS_BYTES;
    BrTeamLds lds;
    lds.acc = (i32 *)base;
    lds.bara = (uint16_t *)(base + 2 * 1024 * 4);
    lds.part = (u64 *)(base + 2 * 1024 * 4 + WAVE_BARA_BYTES);
    lds.xbuf = (u64 *)(base + 2 * 1024 * 4 + WAVE_BARA_BYTES + BRT_PART_ELEMS * 8 + wave * WAVE_XBUF_BYTES);
    lds.tw1x = (const u64 *)g_smem;
    lds.tw1i = (const u64 *)g_smem + 1024;
    const NttLane L = ntt_lane_init(lane);
    brt_bootstrap(P.out_a + gbit * P.out_a_stride, P.out_b + gbit * P.out_b_stride, P.job[job].s0, P.job[job].s1,
                  P.job[job].c0, bit, (const u64 *)P.bk, P.n, P.mu, lds, L, WAVE_UNIFORM(wave),
                  [] { __syncthreads(); });
}


// Small-batch variant for tlwe_mask_size = 2: a team of 3 waves per bit (blind_rotate.h, brtk_*).
// LDS: tables | ACC 12 KiB | bara 1 KiB | partial sums 72 KiB | 3 exchange buffers
#define TEAM2_LDS_BYTES (TABLE_LDS_BYTES + 3 * 1024 * 4 + WAVE_BARA_BYTES + BRTK_PART_ELEMS(2) * 8 + 3 * WAVE_XBUF_BYTES)
__global__ __launch_bounds__(64 * 3, 1) void k_bootstrap_team_k2(BrLaunch P)
{
    load_tables((const u64 *)P.tw_a, (const u64 *)P.tw_b);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long gbit = blockIdx.x;
    const int job = gbit >= P.bits_per_job ? 1 : 0;
    const long bit = gbit - (job ? P.bits_per_job : 0);
    unsigned char *base = g_smem + TABLE_LDS_BYTES;
    BrTeamLds lds;
    lds.acc = (i32 *)base;
    lds.bara = (uint16_t *)(base + 3 * 1024 * 4);
    lds.part = (u64 *)(base + 3 * 1024 * 4 + WAVE_BARA_BYTES);
    lds.xbuf = (u64 *)(base + 3 * 1024 * 4 + WAVE_BARA_BYTES + BRTK_PART_ELEMS(2) * 8 + wave * WAVE_XBUF_BYTES);
    lds.tw1x = (const u64 *)g_smem;
    lds.tw1i = (const u64 *)g_smem + 1024;
    const NttLane L = ntt_lane_init(lane);
    brtk_bootstrap<2>(P.out_a + gbit * P.out_a_stride, P.out_b + gbit * P.out_b_stride, P.job[job].s0, P.job[job].s1,
                      P.job[job].c0, bit, (const u64 *)P.bk, P.n, P.mu, lds, L, WAVE_UNIFORM(wave),
                      [] { __syncthreads(); });
}

hipError_t team_init()
{
    hipError_t e = hipFuncSetAttribute((const void *)k_bootstrap_team, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TEAM_LDS_BYTES);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void *)k_bootstrap_team_k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TEAM2_LDS_BYTES);
}

hipError_t launch_team(const BrLaunch &P, hipStream_t stream)
{
    hipLaunchKernelGGL(k_bootstrap_team, dim3((unsigned)P.nbits_total), dim3(64 * BRT_WAVES), TEAM_LDS_BYTES, stream, P);
    return hipGetLastError();
}

hipError_t launch_team_k2(const BrLaunch &P, hipStream_t stream)
{
    hipLaunchKernelGGL(k_bootstrap_team_k2, dim3((unsigned)P.nbits_total), dim3(64 * 3), TEAM2_LDS_BYTES, stream, P);
    return hipGetLastError();
}
