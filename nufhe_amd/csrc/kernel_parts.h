// kernel_parts.h -- pieces shared by the translation units that hold fused bootstrap kernels (kernels.hip,
// kernels_xfft.hip): the dynamic LDS symbol, the pacing words, the clock probe, the FFT tables in LDS, launch geometry.
// (Moved out of kernels.hip unchanged in round 6.)
#pragma once
#include <hip/hip_runtime.h>

#include "blind_rotate.h"
#include "fft512.h"
#include "kernels.h"

#define BR_PACE_BYTES 128                 /* pacing words of the one-bit kernels, see carve_pace */

extern __shared__ __attribute__((aligned(16))) unsigned char g_smem[];


// Pacing words behind the tables at `base` (zeroed by load_tables / load_ftables): [0..7] progress counter of wave w,
// [8..15] arrival counters of the pair kernel, [16..19] number of registered waves per SIMD, [20..27] their indices.
// Every wave of the work-group must call this (it contains a work-group barrier); waves that will run a bit
// (`active`) register under the SIMD they were placed on (HW_REG_HW_ID bits 5:4 -- observed: SIMDs are dealt in the
// cyclic order 0, 2, 1, 3 from a varying start, so waves w and w + 4 meet, but nothing guarantees it), and a wave
// that finds exactly one other active wave on its SIMD paces itself against it (BrPace, blind_rotate.h).
__device__ __forceinline__ BrPace carve_pace(unsigned char *base, int wave, bool active)
{
    u32 *words = (u32 *)base;
    wave = __builtin_amdgcn_readfirstlane(wave);
    const int simd = (int)__builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11));    // HW_ID.SIMD_ID
    u32 slot = 0;
    if (active && (threadIdx.x & 63) == 0) {
        slot = atomicAdd(&words[16 + simd], 1u);
        if (slot < 2) words[20 + 2 * simd + slot] = (u32)wave;
    }
    slot = (u32)__builtin_amdgcn_readfirstlane((int)slot);
    __syncthreads();
    BrPace p;
    p.mine = nullptr;
    p.other = nullptr;
    if (active && words[16 + simd] == 2u) {
        const int other = __builtin_amdgcn_readfirstlane((int)words[20 + 2 * simd + (slot ^ 1u)]);
        p.mine = words + wave;
        p.other = words + other;
    }
    return p;
}

// life time of the waves of work-group 0 in both clocks (BrLaunch::clock_probe): words [0], [1] = shader-clock and
// 100 MHz ticks of wave 0 (their ratio is the sustained shader clock of the launch); then, for each of its up to 8
// waves w, words [2 + 3 w ..] = start, end (100 MHz ticks) and 1 + the SIMD the wave ran on -- what the pacing
// self-check reads (two waves that share a SIMD must end together)
struct ClockProbe {
    long long t0, r0;
    __device__ __forceinline__ void begin(const BrLaunch &P)
    {
        if (P.clock_probe && blockIdx.x == 0 && (threadIdx.x & 63) == 0) { t0 = clock64(); r0 = wall_clock64(); }
    }
    __device__ __forceinline__ void end(const BrLaunch &P) const
    {
        if (P.clock_probe && blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
            const long long t1 = clock64(), r1 = wall_clock64();
            const unsigned wave = threadIdx.x >> 6;
            if (wave == 0) {
                P.clock_probe[0] = (unsigned long long)(t1 - t0);
                P.clock_probe[1] = (unsigned long long)(r1 - r0);
            }
            if (wave < 8) {
                unsigned hw;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                P.clock_probe[2 + 3 * wave] = (unsigned long long)r0;
                P.clock_probe[3 + 3 * wave] = (unsigned long long)r1;
                P.clock_probe[4 + 3 * wave] = 1ull + ((hw >> 4) & 3u);
            }
        }
    }
};

#define FTABLE_LDS_BYTES ((FFT_TW1_ELEMS + FFT_TW2_ELEMS) * 16)
#define WAVE_FXBUF_BYTES (FFT_XBUF_ELEMS * 16)

__device__ __forceinline__ void load_ftables(const cplx *__restrict__ g_tw1, const cplx *__restrict__ g_tw2)
{
    cplx *t = (cplx *)g_smem;
    for (int i = threadIdx.x; i < FFT_TW1_ELEMS; i += blockDim.x) t[i] = g_tw1[i];
    for (int i = threadIdx.x; i < FFT_TW2_ELEMS; i += blockDim.x) t[FFT_TW1_ELEMS + i] = g_tw2[i];
    if (threadIdx.x < BR_PACE_BYTES / 4) ((u32 *)(g_smem + FTABLE_LDS_BYTES))[threadIdx.x] = 0;
    __syncthreads();
}

static inline unsigned blocks_for(long n, int per) { return (unsigned)((n + per - 1) / per); }

// Waves (= bits) per work-group for a batch of nbits: one work-group per CU and `max_waves` waves
// fill a CU, so a batch that does not fill the chip is spread over as many CUs as possible (a wave
// that has a SIMD to itself runs its 500 iterations ~1.7x sooner than two waves sharing one) and a
// batch needing r rounds uses the smallest group size that still needs r rounds.
static int br_pick_waves(long nbits, int max_waves, int num_cus)
{
    const long per_round = (long)num_cus * max_waves;
    const long rounds = (nbits + per_round - 1) / per_round;
    const long groups = (long)num_cus * rounds;
    long w = (nbits + groups - 1) / groups;
    if (w < 1) w = 1;
    if (w > max_waves) w = max_waves;
    return (int)w;
}
