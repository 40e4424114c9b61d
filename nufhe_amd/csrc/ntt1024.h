// ntt1024.h -- negacyclic 1024-point NTT over GF(2^64 - 2^32 + 1), one wavefront (64 lanes)
// per polynomial, 16 coefficients per lane.
//
// Replaces the reference's ntt1024 device module (nufhe/transform/ntt.mako:42-494: 128 threads x 8
// values, cuFHE-derived) with a CDNA4 wave64 design.  Same mathematical transform as
// ntt_transform_ref (nufhe/transform/ntt.py:30-44):
//
//     A_k = sum_j a_j psi^((2k+1) j),   psi = c^(2^32/2048), c = 0xa70dc47e4cbdf43f, psi^32 = 8
//
// but factored 1024 = 16 x 16 x 4 so that every butterfly twiddle is a power of two (a shift):
//
//   j = j1 + 64 j2 (j1 = lane, j2 = register), k = k2 + 16 (k1a + 16 k1b)
//   pass 1 (in lane, over j2 -> k2): 16-point negacyclic, theta = psi^64 = 2^6
//   twiddle 1: psi^((2 k2 + 1) j1)           -- the ONLY general multiplications (1024 / transform,
//                                               the reference spends 2048: pre-twist + table)
//   exchange 1 through LDS: lane (k2, q), register r, j1 = q + 4 r
//   pass 2 (in lane, over r -> k1a): 16-point cyclic, omega = 8^4 = 2^12
//   exchange 2 through LDS: lane (k2, k1a & 3), register (k1a >> 2, q)
//   twiddle 2: 8^(q k1a) = 2^(12 q (k1a>>2)) * 2^(3 q (k1a&3))   (compile-time shift x per-lane shift < 32)
//   pass 3 (in lane, over q -> k1b): 4-point cyclic, omega = 8^16 = 2^48
//
// The transformed polynomial stays in this "wave layout" (lane L = 4 k2 + (k1a & 3), register
// R = 4 (k1a >> 2) + k1b); the bootstrapping key is stored in the same layout, so the
// multiply-accumulate is register-to-register and the inverse transform simply runs the passes
// backwards.  ntt_freq_index() gives the natural-order frequency of (L, R) for the test hooks and
// for converting keys from/to the reference's natural-order format.
//
// All exchanges go through a private 8704-byte LDS buffer per wave (16 rows x 68 u64; the 4-element
// row padding and the (q + lo) & 3 rotation make every ds_write_b64 / ds_read_b64 conflict-free).
// A wave only synchronises with itself (WAVE_SYNC: a compiler fence, no s_barrier).
#pragma once
#include "ff.h"

#define NTT_N 1024
#define NTT_ROW 68                     /* u64 elements per exchange-buffer row (64 + 4 padding) */
#define NTT_XBUF_ELEMS (16 * NTT_ROW)  /* 1088 u64 = 8704 bytes per wave */

#if defined(NUFHE_EMU)
void emu_yield();                      // tests/emu: fibre switch = wave-level barrier
#define WAVE_SYNC() emu_yield()
#elif defined(__HIP_DEVICE_COMPILE__)
#define WAVE_SYNC()                                             \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
    } while (0)
#else
#define WAVE_SYNC() do { } while (0)
#endif

FF_FN constexpr int br4(int i) { return ((i & 1) << 3) | ((i & 2) << 1) | ((i & 4) >> 1) | ((i & 8) >> 3); }

// natural-order frequency index held by (lane, reg) after ntt_forward
FF_FN constexpr int ntt_freq_index(int lane, int reg)
{
    // lane = 4 k2 + lo, reg = 4 hi + k1b, k1a = 4 hi + lo, k = k2 + 16 (k1a + 16 k1b)
    return (lane >> 2) + 16 * ((4 * (reg >> 2) + (lane & 3)) + 16 * (reg & 3));
}

// coefficient index held by (lane, reg) before ntt_forward / after ntt_inverse
FF_FN constexpr int ntt_coef_index(int lane, int reg) { return lane + 64 * reg; }

// per-lane LDS element offsets, computed once per kernel
struct NttLane {
    int lane;
    int x1w;        // exchange 1 write: + k2 * ROW
    int x1r;        // exchange 1 read:  + 4 r
    int x2w[4];     // exchange 2 write, by lo: + 4 lo + 16 hi
    int x2r[4];     // exchange 2 read, by q:   + 16 hi
    u32 c3;         // 3 * (lane & 3): per-lane shift unit of twiddle 2
};

FF_FN NttLane ntt_lane_init(int lane)
{
    NttLane L;
    L.lane = lane;
    const int k2 = lane >> 2, q = lane & 3;
    L.x1w = lane;
    L.x1r = k2 * NTT_ROW + q;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        L.x2w[t] = k2 * NTT_ROW + ((q + t) & 3);             // lane = (k2, q), t = lo
        L.x2r[t] = k2 * NTT_ROW + ((t + q) & 3) + 4 * q;     // lane = (k2, lo = q), t = q index
    }
    L.c3 = 3u * (u32)q;
    return L;
}

// radix-2 butterfly helpers -------------------------------------------------------------------

// 16-point cyclic NTT with omega = 2^W (W = 12 forward, -12 inverse), decimation in frequency:
// natural-order input, bit-reversed output (x[i] = X[br4(i)]).
template <int W>
FF_FN void ntt16_dif(u64 (&x)[16])
{
#define BFLY(i, j, S)                                 \
    do {                                              \
        const u64 _a = x[i], _b = x[j];               \
        x[i] = ff_add(_a, _b);                        \
        x[j] = ff_submul_pow2<(S)>(_a, _b);           \
    } while (0)
    // stage 0: span 8, twiddle omega^j
    BFLY(0, 8, 0);  BFLY(1, 9, W);  BFLY(2, 10, 2 * W);  BFLY(3, 11, 3 * W);
    BFLY(4, 12, 4 * W);  BFLY(5, 13, 5 * W);  BFLY(6, 14, 6 * W);  BFLY(7, 15, 7 * W);
    // stage 1: span 4, twiddle omega^(2j)
    BFLY(0, 4, 0);  BFLY(1, 5, 2 * W);  BFLY(2, 6, 4 * W);  BFLY(3, 7, 6 * W);
    BFLY(8, 12, 0);  BFLY(9, 13, 2 * W);  BFLY(10, 14, 4 * W);  BFLY(11, 15, 6 * W);
    // stage 2: span 2, twiddle omega^(4j)
    BFLY(0, 2, 0);  BFLY(1, 3, 4 * W);  BFLY(4, 6, 0);  BFLY(5, 7, 4 * W);
    BFLY(8, 10, 0);  BFLY(9, 11, 4 * W);  BFLY(12, 14, 0);  BFLY(13, 15, 4 * W);
    // stage 3: span 1
    BFLY(0, 1, 0);  BFLY(2, 3, 0);  BFLY(4, 5, 0);  BFLY(6, 7, 0);
    BFLY(8, 9, 0);  BFLY(10, 11, 0);  BFLY(12, 13, 0);  BFLY(14, 15, 0);
#undef BFLY
}

// 4-point cyclic NTT with omega = 2^W (W = 48 forward, -48 inverse), natural order in and out
template <int W>
FF_FN void ntt4(u64 &x0, u64 &x1, u64 &x2, u64 &x3)
{
    const u64 u0 = ff_add(x0, x2), v0 = ff_sub(x0, x2);
    const u64 u1 = ff_add(x1, x3), v1 = ff_submul_pow2<W>(x1, x3);
    x0 = ff_add(u0, u1);
    x2 = ff_sub(u0, u1);
    x1 = ff_add(v0, v1);
    x3 = ff_sub(v0, v1);
}

// Forward transform.
//   in : x[j2] = canonical field element of coefficient j = lane + 64 j2
//   out: x[R]  = A_k, k = ntt_freq_index(lane, R)
// xbuf: this wave's private LDS exchange buffer (NTT_XBUF_ELEMS u64)
// tw1f: [16][64] table psi^((2 k2 + 1) j1) at [k2 * 64 + j1] (LDS, shared by the workgroup)
// forward transform of already pre-twisted inputs: x[j2] = a_j * 2^(6 j2) (canonical)
FF_FN void ntt_forward_pretwisted(u64 (&x)[16], u64 *xbuf, const u64 *tw1f, const NttLane &L)
{
    // pass 1 (after the pre-twist by theta^j2 = 2^(6 j2)): cyclic 16-point with omega = 2^12
    ntt16_dif<12>(x);
    // twiddle 1 + exchange 1 (lane j1, reg k2) -> (lane (k2, q), reg r)
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int k2 = br4(i);
        xbuf[L.x1w + k2 * NTT_ROW] = ff_mul(x[i], tw1f[k2 * 64 + L.lane]);
    }
    WAVE_SYNC();
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = xbuf[L.x1r + 4 * r];
    WAVE_SYNC();
    // pass 2: cyclic 16-point over r -> k1a
    ntt16_dif<12>(x);
    // exchange 2 (lane (k2, q), reg k1a) -> (lane (k2, lo), reg (hi, q))
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int k1a = br4(i), hi = k1a >> 2, lo = k1a & 3;
        xbuf[L.x2w[lo] + 4 * lo + 16 * hi] = x[i];
    }
    WAVE_SYNC();
#pragma unroll
    for (int hi = 0; hi < 4; hi++)
#pragma unroll
        for (int q = 0; q < 4; q++) x[4 * hi + q] = xbuf[L.x2r[q] + 16 * hi];
    WAVE_SYNC();
    // twiddle 2: 2^(3 q k1a) = 2^(3 q lo) [per lane, < 2^28] * 2^(12 q hi) [compile time]
#define TW2(hi, q) x[4 * (hi) + (q)] = ff_mul_pow2<12 * (q) * (hi)>(ff_mul_pow2_var(x[4 * (hi) + (q)], (q) * L.c3))
    TW2(0, 1); TW2(0, 2); TW2(0, 3);
    TW2(1, 1); TW2(1, 2); TW2(1, 3);
    TW2(2, 1); TW2(2, 2); TW2(2, 3);
    TW2(3, 1); TW2(3, 2); TW2(3, 3);
#undef TW2
    // pass 3: 4-point over q -> k1b
    ntt4<48>(x[0], x[1], x[2], x[3]);
    ntt4<48>(x[4], x[5], x[6], x[7]);
    ntt4<48>(x[8], x[9], x[10], x[11]);
    ntt4<48>(x[12], x[13], x[14], x[15]);
}

FF_FN void ntt_forward(u64 (&x)[16], u64 *xbuf, const u64 *tw1f, const NttLane &L)
{
#define PRE(j2) x[j2] = ff_mul_pow2<6 * (j2)>(x[j2])
    PRE(1); PRE(2); PRE(3); PRE(4); PRE(5); PRE(6); PRE(7); PRE(8);
    PRE(9); PRE(10); PRE(11); PRE(12); PRE(13); PRE(14); PRE(15);
#undef PRE
    ntt_forward_pretwisted(x, xbuf, tw1f, L);
}

// forward transform of a polynomial with SMALL coefficients (gadget digits, -2^9 <= d < 2^9):
// d[j2] = coefficient lane + 64 j2.  The pre-twist d * 2^(6 j2) needs no modular reduction while it
// (or its multiple of eps, for shifts >= 64) fits a signed 64-bit word: ff_small_times_pow2.
FF_FN void ntt_forward_small(u64 (&x)[16], const i32 (&d)[16], u64 *xbuf, const u64 *tw1f, const NttLane &L)
{
#define PRES(j2) x[j2] = ff_small_times_pow2<6 * (j2)>(d[j2])
#define PREG(j2) x[j2] = ff_mul_pow2<6 * (j2)>(ff_from_i32(d[j2]))
    PRES(0); PRES(1); PRES(2); PRES(3); PRES(4); PRES(5); PRES(6); PRES(7); PRES(8); PRES(9);
    PREG(10); PRES(11); PRES(12); PRES(13); PRES(14); PREG(15);     // shifts 60 and 90 take the general route
#undef PRES
#undef PREG
    ntt_forward_pretwisted(x, xbuf, tw1f, L);
}

// Inverse transform (includes the 1/N factor, folded into tw1i).
//   in : x[R]  = A_k, k = ntt_freq_index(lane, R), canonical
//   out: x[j2] = canonical field element of coefficient j = lane + 64 j2
// tw1i: [16][64] table psi^(-(2 k2 + 1) j1) / 1024 at [k2 * 64 + j1]
// NEG_OUT = true: for j2 >= 1 the output is MINUS the coefficient (the post-twist 2^(-6 j2) =
// -2^(96 - 6 j2) is applied without its sign; the caller folds the sign into its accumulation).
template <bool NEG_OUT>
FF_FN void ntt_inverse_t(u64 (&x)[16], u64 *xbuf, const u64 *tw1i, const NttLane &L)
{
    // pass 3 inverse: 4-point over k1b -> q with omega^-1
    ntt4<-48>(x[0], x[1], x[2], x[3]);
    ntt4<-48>(x[4], x[5], x[6], x[7]);
    ntt4<-48>(x[8], x[9], x[10], x[11]);
    ntt4<-48>(x[12], x[13], x[14], x[15]);
    // twiddle 2 inverse: 2^-(12 q hi + 3 q lo) = 2^(31 - 3 q lo) [per lane, in [4, 31]] * 2^(-12 q hi - 31)
#define TW2I(hi, q) x[4 * (hi) + (q)] = ff_mul_pow2<-12 * (q) * (hi) - 31>(ff_mul_pow2_var(x[4 * (hi) + (q)], 31u - (q) * L.c3))
    TW2I(0, 1); TW2I(0, 2); TW2I(0, 3);
    TW2I(1, 1); TW2I(1, 2); TW2I(1, 3);
    TW2I(2, 1); TW2I(2, 2); TW2I(2, 3);
    TW2I(3, 1); TW2I(3, 2); TW2I(3, 3);
#undef TW2I
    // exchange 2 backwards: (lane (k2, lo), reg (hi, q)) -> (lane (k2, q), reg k1a)
#pragma unroll
    for (int hi = 0; hi < 4; hi++)
#pragma unroll
        for (int q = 0; q < 4; q++) xbuf[L.x2r[q] + 16 * hi] = x[4 * hi + q];
    WAVE_SYNC();
#pragma unroll
    for (int k1a = 0; k1a < 16; k1a++) {
        const int hi = k1a >> 2, lo = k1a & 3;
        x[k1a] = xbuf[L.x2w[lo] + 4 * lo + 16 * hi];
    }
    WAVE_SYNC();
    // pass 2 inverse: cyclic 16-point over k1a -> r with omega^-1
    ntt16_dif<-12>(x);
    // exchange 1 backwards: (lane (k2, q), reg r) -> (lane j1, reg k2)
#pragma unroll
    for (int i = 0; i < 16; i++) xbuf[L.x1r + 4 * br4(i)] = x[i];
    WAVE_SYNC();
    // twiddle 1 inverse (and 1/N)
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) x[k2] = ff_mul(xbuf[L.x1w + k2 * NTT_ROW], tw1i[k2 * 64 + L.lane]);
    WAVE_SYNC();
    // pass 1 inverse: cyclic 16-point over k2 -> j2 with omega^-1, then post-twist 2^(-6 j2)
    ntt16_dif<-12>(x);
    u64 y[16];
#define POST(j2) y[j2] = (NEG_OUT && (j2) > 0) ? ff_mul_pow2_lt96<(96 - 6 * (j2)) % 96>(x[br4(j2)]) \
                                               : ff_mul_pow2<-6 * (j2)>(x[br4(j2)])
    POST(0); POST(1); POST(2); POST(3); POST(4); POST(5); POST(6); POST(7); POST(8);
    POST(9); POST(10); POST(11); POST(12); POST(13); POST(14); POST(15);
#undef POST
#pragma unroll
    for (int j2 = 0; j2 < 16; j2++) x[j2] = y[j2];
}

FF_FN void ntt_inverse(u64 (&x)[16], u64 *xbuf, const u64 *tw1i, const NttLane &L)
{
    ntt_inverse_t<false>(x, xbuf, tw1i, L);
}
