#include "ff.h"
#include "ntt1024.h"
#include "blind_rotate.h"
#define LOADX u64 x[16]; for (int r = 0; r < 16; r++) x[r] = io[threadIdx.x + 64 * r];
#define STOREX for (int r = 0; r < 16; r++) io[threadIdx.x + 64 * r] = x[r];
__global__ void p_ntt16(u64 *io) { LOADX ntt16_dif<12>(x); STOREX }
__global__ void p_ntt16inv(u64 *io) { LOADX ntt16_dif<-12>(x); STOREX }
__global__ void p_tablemul(u64 *io, const u64 *tw) { LOADX for (int i = 0; i < 16; i++) x[i] = ff_mul(x[i], tw[i * 64 + threadIdx.x]); STOREX }
__global__ void p_tw2(u64 *io) { LOADX u32 c3 = 3 * (threadIdx.x & 3);
#define TW2(hi, q) x[4 * (hi) + (q)] = ff_mul_pow2<12 * (q) * (hi)>(ff_mul_pow2_var(x[4 * (hi) + (q)], (q) * c3))
    TW2(0, 1); TW2(0, 2); TW2(0, 3); TW2(1, 1); TW2(1, 2); TW2(1, 3); TW2(2, 1); TW2(2, 2); TW2(2, 3); TW2(3, 1); TW2(3, 2); TW2(3, 3);
    STOREX }
__global__ void p_tw2i(u64 *io) { LOADX u32 c3 = 3 * (threadIdx.x & 3);
#define TW2I(hi, q) x[4 * (hi) + (q)] = ff_mul_pow2<-12 * (q) * (hi) - 31>(ff_mul_pow2_var(x[4 * (hi) + (q)], 31u - (q) * c3))
    TW2I(0, 1); TW2I(0, 2); TW2I(0, 3); TW2I(1, 1); TW2I(1, 2); TW2I(1, 3); TW2I(2, 1); TW2I(2, 2); TW2I(2, 3); TW2I(3, 1); TW2I(3, 2); TW2I(3, 3);
    STOREX }
__global__ void p_ntt4(u64 *io) { LOADX ntt4<48>(x[0], x[1], x[2], x[3]); ntt4<48>(x[4], x[5], x[6], x[7]); ntt4<48>(x[8], x[9], x[10], x[11]); ntt4<48>(x[12], x[13], x[14], x[15]); STOREX }
__global__ void p_presmall(u64 *io, const i32 *d) { u64 x[16]; i32 dg[16]; for (int r = 0; r < 16; r++) dg[r] = d[threadIdx.x + 64 * r];
#define PRES(j2) x[j2] = ff_small_times_pow2<6 * (j2)>(dg[j2])
#define PREG(j2) x[j2] = ff_mul_pow2<6 * (j2)>(ff_from_i32(dg[j2]))
    PRES(0); PRES(1); PRES(2); PRES(3); PRES(4); PRES(5); PRES(6); PRES(7); PRES(8);
    PREG(9); PREG(10); PREG(11); PREG(12); PREG(13); PREG(14); PREG(15);
    STOREX }
__global__ void p_post(u64 *io) { LOADX u64 y[16];
#define POST(j2) y[j2] = ((j2) > 0) ? ff_mul_pow2_lt96<(96 - 6 * (j2)) % 96>(x[br4(j2)]) : x[0]
    POST(0); POST(1); POST(2); POST(3); POST(4); POST(5); POST(6); POST(7); POST(8); POST(9); POST(10); POST(11); POST(12); POST(13); POST(14); POST(15);
    for (int r = 0; r < 16; r++) io[threadIdx.x + 64 * r] = y[r]; }
__global__ void p_mac2(u64 *io, const u64 *bk) { LOADX u64 x1[16]; for (int r = 0; r < 16; r++) x1[r] = io[1024 + threadIdx.x + 64 * r];
    u64 sum[2][16]; br_mac2<1, true>(sum, x, x1, bk, threadIdx.x);
    for (int r = 0; r < 16; r++) { io[threadIdx.x + 64 * r] = sum[0][r]; io[1024 + threadIdx.x + 64 * r] = sum[1][r]; } }
__global__ void p_mac2b(u64 *io, const u64 *bk) { LOADX u64 x1[16]; for (int r = 0; r < 16; r++) x1[r] = io[1024 + threadIdx.x + 64 * r];
    u64 sum[2][16]; for (int r = 0; r < 16; r++) { sum[0][r] = io[2048 + threadIdx.x + 64 * r]; sum[1][r] = io[3072 + threadIdx.x + 64 * r]; }
    br_mac2<1, false>(sum, x, x1, bk, threadIdx.x);
    for (int r = 0; r < 16; r++) { io[threadIdx.x + 64 * r] = sum[0][r]; io[1024 + threadIdx.x + 64 * r] = sum[1][r]; } }
