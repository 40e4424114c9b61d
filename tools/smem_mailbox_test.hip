// Do scalar stores / loads work as a wave-to-wave mailbox inside a CU on gfx950?  Two waves of one work-group ping-pong a
// counter through a global word with s_store_dword / s_load_dword (no VGPRs involved).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(512) void k(uint32_t *words, long long *out, int rounds)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *mine = words + blockIdx.x * 8 + wave;
    uint32_t *other = words + blockIdx.x * 8 + (wave ^ 4);
    long long t0 = clock64();
    uint32_t seen = 0, spins = 0;
    for (int r = 1; r <= rounds; r++) {
        uint32_t v = (uint32_t)r;
        asm volatile("s_store_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" :: "s"(v), "s"(mine) : "memory");
        // wait until the partner has published >= r
        for (;;) {
            uint32_t o;
            asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(o) : "s"(other) : "memory");
            spins++;
            if (o >= (uint32_t)r || spins > 2000000u) { seen = o; break; }
            asm volatile("s_dcache_inv");       // not expected to be needed inside one CU
        }
    }
    long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 8 + wave) * 3] = t1 - t0; out[(blockIdx.x * 8 + wave) * 3 + 1] = seen; out[(blockIdx.x * 8 + wave) * 3 + 2] = spins; }
}
int main()
{
    uint32_t *w; long long *o;
    hipMalloc(&w, 256 * 8 * 4); hipMemset(w, 0, 256 * 8 * 4);
    hipMalloc(&o, 256 * 8 * 3 * 8);
    const int rounds = 1000;
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, w, o, rounds);
    hipError_t e = hipDeviceSynchronize();
    printf("sync: %s\n", hipGetErrorString(e));
    long long h[256 * 8 * 3];
    hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; i++) printf("wave %d: %lld ticks, last seen %lld, spins %lld (%.0f ticks per round trip)\n", i, h[i * 3], h[i * 3 + 1], h[i * 3 + 2], (double)h[i * 3] / rounds);
    return 0;
}
