// kernels_team8.hip -- k_bootstrap_team8, the 8-waves-per-bit half-ring kernel of the smallest NTT batches
// (blind_rotate.h brh_*, ntt512_half.h), in a translation unit of its own: it is compiled with
// `-mllvm -amdgpu-sched-strategy=max-ilp` (Makefile).  In its inverse phase four waves run alone on their SIMDs, where an
// instruction that reads the result of the one before it costs 8 cycles instead of 4; scheduling for instruction-level
// parallelism is worth -3 % there (NOTES.md round 3: 4.79 -> 4.65 ms per gate), while the kernels that run two paced waves
// per SIMD lose 1.5-3 % with the same flag -- hence the separate unit.  For the same reason this unit is built with
// -DFF_MULWIDE_PLAIN: the carry-out form of the 64 x 64 product (ff.h, round 4) has fewer instructions but a longer
// dependent chain with an SGPR hop; the wave kernels gain 2.5 % from it, this kernel loses 3.5 % (4.60 -> 4.76 ms).
#include <hip/hip_runtime.h>

#include "blind_rotate.h"
#include "ff.h"
#include "kernels.h"
#include "ntt1024.h"

#define WAVE_BARA_BYTES (BR_MAX_LWE * 2)

extern __shared__ __attribute__((aligned(16))) unsigned char g_smem[];

// LDS: half-ring tables | ACC 8 KiB | bara 1 KiB | partial sums 64 KiB | join 16 KiB | 8 exchange buffers
#define NTH_TABLE_BYTES (NTH_TABLE_ELEMS * 8)
#define TEAM8_LDS_BYTES (NTH_TABLE_BYTES + 2 * 1024 * 4 + WAVE_BARA_BYTES + BRH_PART_ELEMS * 8 + BRH_JOIN_ELEMS * 8 + BRH_WAVES * NTH_XBUF_ELEMS * 8)
static_assert(TEAM8_LDS_BYTES <= 160 * 1024, "LDS budget of the half-ring team kernel");
__global__ __launch_bounds__(64 * BRH_WAVES, 1) void k_bootstrap_team8(BrLaunch P)
{
    {
        u64 *t = (u64 *)g_smem;
        const u64 *g = (const u64 *)P.tw_half;
        for (int i = threadIdx.x; i < NTH_TABLE_ELEMS; i += blockDim.x) t[i] = g[i];
        __syncthreads();
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long gbit = blockIdx.x;
    const int job = gbit >= P.bits_per_job ? 1 : 0;
    const long bit = gbit - (job ? P.bits_per_job : 0);
    unsigned char *base = g_smem + NTH_TABLE_BYTES;
    BrHalfLds lds;
    lds.tables = (const u64 *)g_smem;
    lds.acc = (i32 *)base;
    lds.bara = (uint16_t *)(base + 2 * 1024 * 4);
    lds.part = (u64 *)(base + 2 * 1024 * 4 + WAVE_BARA_BYTES);
    lds.join = lds.part + BRH_PART_ELEMS;
    lds.xbuf = lds.join + BRH_JOIN_ELEMS + wave * NTH_XBUF_ELEMS;
    brh_bootstrap(P.out_a + gbit * P.out_a_stride, P.out_b + gbit * P.out_b_stride, P.job[job].s0, P.job[job].s1,
                  P.job[job].c0, bit, (const u64 *)P.bk_half, P.n, P.mu, lds, lane, WAVE_UNIFORM(wave),
                  [] { __syncthreads(); });
}

hipError_t team8_init()
{
    return hipFuncSetAttribute((const void *)k_bootstrap_team8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TEAM8_LDS_BYTES);
}

hipError_t launch_team8(const BrLaunch &P, hipStream_t stream)
{
    hipLaunchKernelGGL(k_bootstrap_team8, dim3((unsigned)P.nbits_total), dim3(64 * BRH_WAVES), TEAM8_LDS_BYTES, stream, P);
    return hipGetLastError();
}

#if defined(BR_PROBE)
// variant builds only (blind_rotate.h, BR_PROBE): read and clear the phase tick counters of this kernel
extern "C" int nufhe_probe_read_team8(unsigned long long *out8)
{
    unsigned long long zero[8] = {0};
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_brh_probe), sizeof(zero)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_brh_probe), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}

#endif
