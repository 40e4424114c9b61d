// ntt1024_l4.h -- the negacyclic NTT-1024 of ntt1024.h carried out on redundant 24-bit limbs (ff24.h).
//
// Same factorisation, same lane / register layout, same LDS exchange pattern and same tables as
// ntt1024.h (16 x 16 x 4, twiddles that are powers of two everywhere except the one table layer), so
// the transformed values are the same field elements and the bootstrapping key layout is unchanged.
// What changes is the arithmetic between the general multiplications: butterflies are 4 + 4 plain
// 32-bit adds, twiddles 2^(24 k) are limb rotations, twiddles with a sub-limb part cost one split per
// limb.  These are the two transforms of the blind-rotation loop:
//
//   ntt_forward_small_l4   gadget digits (|d| <= 2^9)      -> 64-bit representatives in the wave layout
//   ntt_inverse_l4_i32     64-bit field elements (any rep.) -> coefficients mod 2^32
//
// Replaces ntt1024 of the reference (nufhe/transform/ntt.mako:42-494) on the hot path; the generic
// transforms of ntt1024.h remain for the test hooks and key generation.
//
// Limb bounds (|w| must stay <= 2^30 for l4_to_u64, < 2^31 always), forward:
//   placement <= 2^27 | pass 1: stage 0 puts its two inputs into different limbs (<= 2^27), odd
//   twiddles re-normalise, three more stages <= 2^30 | table product -> (-2^24, 2^24) | pass 2
//   <= 2^28 | twiddle 2 (split) < 2^26, the q = 0 column stays <= 2^28 | pass 3 < 2^29.
// inverse:
//   l4_from_u64 < 2^24 | pass 3 <= 2^26 | twiddle 2 < 2^25 (q = 0: <= 2^26) | pass 2: sum of 16
//   <= 4 * 2^26 + 12 * 2^25 < 2^30 | table product -> (-2^24, 2^24) | pass 1 <= 2^28 | post-twist: limb
//   rotation, then the sub-limb part on the packed word (l4_to_i32_shl).
#pragma once
#include "ff24.h"
#include "ntt1024.h"

// 16-point cyclic pass, omega = 2^W (W = +-12), decimation in frequency, bit-reversed output
template <int W>
FF_FN void l4_ntt16_dif(L4 (&x)[16])
{
#define BFLY(i, j, S) l4_bfly<(S)>(x[i], x[j])
    BFLY(0, 8, 0);  BFLY(1, 9, W);  BFLY(2, 10, 2 * W);  BFLY(3, 11, 3 * W);
    BFLY(4, 12, 4 * W);  BFLY(5, 13, 5 * W);  BFLY(6, 14, 6 * W);  BFLY(7, 15, 7 * W);
    BFLY(0, 4, 0);  BFLY(1, 5, 2 * W);  BFLY(2, 6, 4 * W);  BFLY(3, 7, 6 * W);
    BFLY(8, 12, 0);  BFLY(9, 13, 2 * W);  BFLY(10, 14, 4 * W);  BFLY(11, 15, 6 * W);
    BFLY(0, 2, 0);  BFLY(1, 3, 4 * W);  BFLY(4, 6, 0);  BFLY(5, 7, 4 * W);
    BFLY(8, 10, 0);  BFLY(9, 11, 4 * W);  BFLY(12, 14, 0);  BFLY(13, 15, 4 * W);
    BFLY(0, 1, 0);  BFLY(2, 3, 0);  BFLY(4, 5, 0);  BFLY(6, 7, 0);
    BFLY(8, 9, 0);  BFLY(10, 11, 0);  BFLY(12, 13, 0);  BFLY(14, 15, 0);
#undef BFLY
}

// 4-point cyclic pass, omega = 2^W (W = +-48: a rotation by two limbs), natural order
template <int W>
FF_FN void l4_ntt4(L4 &x0, L4 &x1, L4 &x2, L4 &x3)
{
    constexpr int T = ((W % 192) + 192) % 192;
    static_assert(T % 24 == 0, "whole-limb twiddle");
    L4 u0, v0, u1, v1;
    l4_add(u0, x0, x2);
    l4_sub(v0, x0, x2);
    l4_add(u1, x1, x3);
    l4_sub_rot<T / 24>(v1, x1, x3);
    l4_add(x0, u0, u1);
    l4_sub(x2, u0, u1);
    l4_add(x1, v0, v1);
    l4_sub(x3, v0, v1);
}

// ---------------------------------------------------------------------------------------------
// Per-lane twiddles x *= 2^(BASE + STEP * lo), lo = lane & 3 (the "twiddle 2" layer between the
// 16-point and the 4-point pass: 8^(q k1a), 8 = 2^3 the 64-th root of unity).
// For each of the four values of lo the exponent splits into 24 k + s: the sub-limb amount s comes
// from a 4 x 5-bit table indexed by the lane, the rotation k differs between lanes by at most two
// steps, which are applied as conditional single-limb rotations; the common part is a renaming.
// ---------------------------------------------------------------------------------------------
struct L4LaneTw {
    // exponent of lane class lo (0..3), reduced to [0, 192)
    static constexpr int expo(int base, int step, int lo) { return (((base + step * lo) % 192) + 192) % 192; }
};

template <int BASE, int STEP>
FF_FN void l4_mul_pow2_lane(L4 &x, u32 lo)
{
    constexpr int e0 = L4LaneTw::expo(BASE, STEP, 0), e1 = L4LaneTw::expo(BASE, STEP, 1),
                  e2 = L4LaneTw::expo(BASE, STEP, 2), e3 = L4LaneTw::expo(BASE, STEP, 3);
    constexpr int k0 = e0 / 24, k1 = e1 / 24, k2 = e2 / 24, k3 = e3 / 24;
    // rotation steps of each lane class relative to the smallest one (mod 8, all within 0..2)
    constexpr int d01 = (k1 - k0 + 8) % 8, d02 = (k2 - k0 + 8) % 8, d03 = (k3 - k0 + 8) % 8;
    constexpr bool up = d01 <= 2 && d02 <= 2 && d03 <= 2;          // classes rotate further than lo = 0
    constexpr int kref = up ? k0 : k3;                             // else lo = 3 is the smallest
    constexpr int l0 = (k0 - kref + 8) % 8, l1 = (k1 - kref + 8) % 8, l2 = (k2 - kref + 8) % 8,
                  l3 = (k3 - kref + 8) % 8;
    static_assert(l0 <= 2 && l1 <= 2 && l2 <= 2 && l3 <= 2, "lane classes more than two limb rotations apart");
    constexpr u32 stab = (u32)(e0 % 24) | ((u32)(e1 % 24) << 5) | ((u32)(e2 % 24) << 10) | ((u32)(e3 % 24) << 15);
    constexpr u32 ltab = (u32)l0 | ((u32)l1 << 2) | ((u32)l2 << 4) | ((u32)l3 << 6);
    L4 t = x;
    if constexpr (stab != 0) {
        const u32 s = (stab >> (5u * lo)) & 31u;
        const u32 s24 = 24u - s;
        l4_shl_var(t, t, s, s24, (1u << s24) - 1u);
    }
    if constexpr (ltab != 0) {
        const u32 lev = (ltab >> (2u * lo)) & 3u;
        {
            const bool c = lev >= 1u;
            const u32 n3 = 0u - t.w[3];
            L4 r;
            r.w[0] = c ? n3 : t.w[0];
            r.w[1] = c ? t.w[0] : t.w[1];
            r.w[2] = c ? t.w[1] : t.w[2];
            r.w[3] = c ? t.w[2] : t.w[3];
            t = r;
        }
        if constexpr (l0 == 2 || l1 == 2 || l2 == 2 || l3 == 2) {
            const bool c = lev >= 2u;
            const u32 n3 = 0u - t.w[3];
            L4 r;
            r.w[0] = c ? n3 : t.w[0];
            r.w[1] = c ? t.w[0] : t.w[1];
            r.w[2] = c ? t.w[1] : t.w[2];
            r.w[3] = c ? t.w[2] : t.w[3];
            t = r;
        }
    }
    l4_rot<kref>(x, t);
}

// ---------------------------------------------------------------------------------------------
// LDS exchanges.  Exchange 1 sits next to the general multiplication, which needs a packed 64-bit operand
// anyway: the elements cross as 8-byte representatives in one round (see the transforms below).  Exchange
// 2 moves limbs (16 bytes per element); the per-wave buffer of ntt1024.h holds 8 bytes per slot, so it
// runs in two rounds (limbs 0-1, then limbs 2-3) over the same conflict-free addresses.
// ---------------------------------------------------------------------------------------------
FF_FN u64 l4_pair(const L4 &x, int h) { return ((u64)x.w[2 * h + 1] << 32) | x.w[2 * h]; }
FF_FN void l4_set_pair(L4 &x, int h, u64 v)
{
    x.w[2 * h] = (u32)v;
    x.w[2 * h + 1] = (u32)(v >> 32);
}

// exchange 2 forward: x[i] = value of k1a = br4(i) at lane (k2, q) -> x[4 hi + q] at lane (k2, lo)
FF_FN void l4_exchange2_fwd(L4 (&x)[16], u64 *xbuf, const NttLane &L)
{
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int k1a = br4(i), hi = k1a >> 2, lo = k1a & 3;
            xbuf[L.x2w[lo] + 4 * lo + 16 * hi] = l4_pair(x[i], h);
        }
        WAVE_SYNC();
#pragma unroll
        for (int hi = 0; hi < 4; hi++)
#pragma unroll
            for (int q = 0; q < 4; q++) l4_set_pair(x[4 * hi + q], h, xbuf[L.x2r[q] + 16 * hi]);
        WAVE_SYNC();
    }
}

// exchange 2 backwards: x[4 hi + q] at lane (k2, lo) -> x[k1a] at lane (k2, q)
FF_FN void l4_exchange2_inv(L4 (&x)[16], u64 *xbuf, const NttLane &L)
{
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
        for (int hi = 0; hi < 4; hi++)
#pragma unroll
            for (int q = 0; q < 4; q++) xbuf[L.x2r[q] + 16 * hi] = l4_pair(x[4 * hi + q], h);
        WAVE_SYNC();
#pragma unroll
        for (int k1a = 0; k1a < 16; k1a++) {
            const int hi = k1a >> 2, lo = k1a & 3;
            l4_set_pair(x[k1a], h, xbuf[L.x2w[lo] + 4 * lo + 16 * hi]);
        }
        WAVE_SYNC();
    }
}

// ---------------------------------------------------------------------------------------------
// Forward transform of a polynomial of gadget digits (|d| <= 2^9); d[j2] = coefficient lane + 64 j2.
// out[R]: a 64-bit representative (not canonical) of A_k, k = ntt_freq_index(lane, R).
// ---------------------------------------------------------------------------------------------
FF_FN void ntt_forward_small_l4(u64 (&out)[16], const i32 (&d)[16], u64 *xbuf, const u64 *tw1x, const NttLane &L)
{
    L4 x[16];
    // pre-twist 2^(6 j2): a digit lands in ONE limb; the zero limbs fold away in the first stages
#define PLACE(j2) l4_place<6 * (j2)>(x[j2], d[j2])
    PLACE(0); PLACE(1); PLACE(2); PLACE(3); PLACE(4); PLACE(5); PLACE(6); PLACE(7);
    PLACE(8); PLACE(9); PLACE(10); PLACE(11); PLACE(12); PLACE(13); PLACE(14); PLACE(15);
#undef PLACE
    l4_ntt16_dif<12>(x);
    // exchange 1 carries 64-bit representatives (8 bytes per element, one round); twiddle 1, the one
    // general multiplication layer psi^((2 k2 + 1) j1), is applied by the RECEIVING lane from the table
    // laid out for it (ntt_make_tw1x: tw1x[r * 64 + lane] = psi^((2 k2 + 1)(q + 4 r)), lane = (k2, q)),
    // and its 128-bit products are split straight into limbs
#pragma unroll
    for (int i = 0; i < 16; i++) xbuf[L.x1w + br4(i) * NTT_ROW] = l4_to_u64(x[i]);
    WAVE_SYNC();
#pragma unroll
    for (int r = 0; r < 16; r++) {
        u64 lo, hi;
        ff_mul_wide(xbuf[L.x1r + 4 * r], tw1x[r * 64 + L.lane], lo, hi);
        l4_from_u128(x[r], lo, hi);
    }
    WAVE_SYNC();
    l4_ntt16_dif<12>(x);
    l4_exchange2_fwd(x, xbuf, L);
    // twiddle 2: 2^(12 q hi + 3 q lo)
    const u32 lo = (u32)L.lane & 3u;
#define TW2(hi, q) l4_mul_pow2_lane<12 * (q) * (hi), 3 * (q)>(x[4 * (hi) + (q)], lo)
    TW2(0, 1); TW2(0, 2); TW2(0, 3);
    TW2(1, 1); TW2(1, 2); TW2(1, 3);
    TW2(2, 1); TW2(2, 2); TW2(2, 3);
    TW2(3, 1); TW2(3, 2); TW2(3, 3);
#undef TW2
    l4_ntt4<48>(x[0], x[1], x[2], x[3]);
    l4_ntt4<48>(x[4], x[5], x[6], x[7]);
    l4_ntt4<48>(x[8], x[9], x[10], x[11]);
    l4_ntt4<48>(x[12], x[13], x[14], x[15]);
#pragma unroll
    for (int r = 0; r < 16; r++) out[r] = l4_to_u64(x[r]);
}

// ---------------------------------------------------------------------------------------------
// Inverse transform (1/N folded into tw1i) of field elements in the wave layout, given as limbs
// (w0 in (-2^24, 2^24), w1 in (-2^11, 2^24), w2, w3 in [0, 2^24): what l4_dot2 / l4_from_u64 deliver),
// down to the coefficients modulo 2^32 (ntt.mako:402-408): the caller guarantees that the
// true coefficients are integers of magnitude < 2^62 (here: < 2^52).
//   c[j2] = coefficient lane + 64 j2 for j2 = 0, and MINUS that coefficient for j2 >= 1
// (the post-twist 2^(-6 j2) = -2^(96 - 6 j2) is applied without its sign, as in ntt_inverse_t<true>).
// ---------------------------------------------------------------------------------------------
FF_FN void ntt_inverse_l4_core(u32 (&c)[16], L4 (&x)[16], u64 *xbuf, const u64 *tw1i, const NttLane &L)
{
    l4_ntt4<-48>(x[0], x[1], x[2], x[3]);
    l4_ntt4<-48>(x[4], x[5], x[6], x[7]);
    l4_ntt4<-48>(x[8], x[9], x[10], x[11]);
    l4_ntt4<-48>(x[12], x[13], x[14], x[15]);
    // twiddle 2 inverse: 2^-(12 q hi + 3 q lo)
    const u32 lo = (u32)L.lane & 3u;
#define TW2I(hi, q) l4_mul_pow2_lane<-12 * (q) * (hi), -3 * (q)>(x[4 * (hi) + (q)], lo)
    TW2I(0, 1); TW2I(0, 2); TW2I(0, 3);
    TW2I(1, 1); TW2I(1, 2); TW2I(1, 3);
    TW2I(2, 1); TW2I(2, 2); TW2I(2, 3);
    TW2I(3, 1); TW2I(3, 2); TW2I(3, 3);
#undef TW2I
    l4_exchange2_inv(x, xbuf, L);
    l4_ntt16_dif<-12>(x);
    // exchange 1 backwards on 64-bit representatives, then twiddle 1 inverse (and 1/N) at the receiver
#pragma unroll
    for (int i = 0; i < 16; i++) xbuf[L.x1r + 4 * br4(i)] = l4_to_u64(x[i]);
    WAVE_SYNC();
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) {
        u64 lo, hi;
        ff_mul_wide(xbuf[L.x1w + k2 * NTT_ROW], tw1i[k2 * 64 + L.lane], lo, hi);
        l4_from_u128(x[k2], lo, hi);
    }
    WAVE_SYNC();
    l4_ntt16_dif<-12>(x);
    // post-twist 2^(E), E = (j2 > 0 ? 96 : 0) - 6 j2 = 24 k + s, and conversion: the rotation by k is a
    // renaming, the sub-limb part s is applied to the packed word inside l4_to_i32_shl
#define POST(j2)                                                              \
    do {                                                                      \
        constexpr int E = ((j2) > 0 ? 96 : 0) - 6 * (j2);                     \
        L4 y;                                                                 \
        l4_rot<E / 24>(y, x[br4(j2)]);                                        \
        c[j2] = l4_to_i32_shl<E % 24>(y);                                     \
    } while (0)
    POST(0); POST(1); POST(2); POST(3); POST(4); POST(5); POST(6); POST(7); POST(8);
    POST(9); POST(10); POST(11); POST(12); POST(13); POST(14); POST(15);
#undef POST
}

// the same from 64-bit field elements (any representative)
FF_FN void ntt_inverse_l4_i32(u32 (&c)[16], const u64 (&in)[16], u64 *xbuf, const u64 *tw1i, const NttLane &L)
{
    L4 x[16];
#pragma unroll
    for (int r = 0; r < 16; r++) l4_from_u64(x[r], in[r]);
    ntt_inverse_l4_core(c, x, xbuf, tw1i, L);
}
