// ff24.h -- GF(P), P = 2^64 - 2^32 + 1, in a carry-free redundant form for the CDNA4 integer VALU.
//
// An element is four signed 32-bit limbs in radix 2^24 ("L4"):
//
//     value = w[0] + w[1] 2^24 + w[2] 2^48 + w[3] 2^72   (mod P),      2^96 = -1 (mod P)
//
// Why: on gfx950 the plain 32-bit v_add_u32 / v_sub_u32 / v_and / v_lshrrev / v_ashrrev issue at twice
// the rate of everything that produces or consumes a carry, compares, multiplies or is 64 bits wide
// (profiles/r01_microbench_valu_rates.txt, profiles/r02_microbench_l4.txt).  With 8 spare bits per limb
//   * addition / subtraction are 4 independent v_add_u32 / v_sub_u32 -- no carry chain, no comparison
//     against P, no VCC hazard;
//   * a twiddle 2^(24 k) is a negacyclic ROTATION of the limbs (register renaming; the sign of a
//     wrapped limb is absorbed by swapping the operands of the butterfly's subtraction);
//   * a twiddle 2^(24 k + s) adds one split per limb (shift, mask, add), which also re-normalises;
//   * limbs only have to stay inside int32: every add may grow them by one bit, and the places where
//     they are brought back under 2^24 (sub-limb shifts, the product split) are spaced so that the
//     bound |w| <= 2^30 holds everywhere (bounds are stated at each call site in ntt1024_l4.h).
// General multiplications (one table layer per transform, and the key product) still go through
// 64 x 64 -> 128-bit v_mad_u64_u32 chains: l4_to_u64 packs an element into ANY 64-bit representative
// and l4_from_u128 splits a product straight into limbs -- the 128 -> 64-bit modular reduction of
// ff.h disappears from the transform.
//
// Semantics replaced: the reference's finite-field modules (nufhe/transform/arithmetic.mako:78-161
// add/sub, :465-1045 the lsh family); results are bit-identical because every routine here is exact
// modulo P and the path ends in the canonical-range conversion of ntt.mako:402-408.
//
// The header also compiles for the host (tests/emu).
#pragma once
#include "ff.h"

struct L4 {
    u32 w[4];   // two's-complement limbs, kept in unsigned registers so that wrap-around is defined
};

#define L4_MASK 0x00FFFFFFu

FF_FN u32 l4_sar(u32 x, int s) { return (u32)((i32)x >> s); }

// bytes [B, B+3) of the 64-bit word hi:lo as a 24-bit number (one v_perm_b32 on the device)
template <int B>
FF_FN u32 l4_ext24(u32 hi, u32 lo)
{
    static_assert(B >= 1 && B <= 4, "byte offset");
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, 0x0C000000u | ((u32)(B + 2) << 16) | ((u32)(B + 1) << 8) | (u32)B);
#else
    return (u32)(((((u64)hi) << 32) | lo) >> (8 * B)) & L4_MASK;
#endif
}

// 32-bit word made of bytes (lo.b[SL], lo.b[SL+1], ..) then (hi.b0, ...): NLO bytes of lo starting at
// byte SL followed by 4 - NLO bytes of hi starting at byte 0
template <int SL, int NLO>
FF_FN u32 l4_pack(u32 hi, u32 lo)
{
    static_assert(SL + NLO == 3, "the low part ends at byte 2: limbs carry 24 significant bits");
#if defined(__HIP_DEVICE_COMPILE__)
    u32 sel = 0;
    for (int i = 0; i < 4; i++) sel |= (u32)(i < NLO ? SL + i : 4 + (i - NLO)) << (8 * i);
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    return ((lo & L4_MASK) >> (8 * SL)) | (hi << (8 * NLO));
#endif
}

FF_FN void l4_zero(L4 &r) { r.w[0] = r.w[1] = r.w[2] = r.w[3] = 0; }

FF_FN void l4_add(L4 &r, const L4 &a, const L4 &b)
{
#pragma unroll
    for (int i = 0; i < 4; i++) r.w[i] = a.w[i] + b.w[i];
}

FF_FN void l4_sub(L4 &r, const L4 &a, const L4 &b)
{
#pragma unroll
    for (int i = 0; i < 4; i++) r.w[i] = a.w[i] - b.w[i];
}

// sign and source limb of output limb p under multiplication by 2^(24 KROT), KROT in [0, 8)
FF_FN constexpr int l4_rot_src(int p, int krot) { return (p - (krot & 3)) & 3; }
FF_FN constexpr bool l4_rot_neg(int p, int krot) { return ((p < (krot & 3)) ? 1 : 0) != ((krot >> 2) & 1); }

// r = (a - b) * 2^(24 KROT): four subtractions, the rotation is free
template <int KROT>
FF_FN void l4_sub_rot(L4 &r, const L4 &a, const L4 &b)
{
    L4 t;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int q = l4_rot_src(p, KROT);
        t.w[p] = l4_rot_neg(p, KROT) ? b.w[q] - a.w[q] : a.w[q] - b.w[q];
    }
    r = t;
}

// r = a * 2^(24 KROT): renaming plus one negation per wrapped limb
template <int KROT>
FF_FN void l4_rot(L4 &r, const L4 &a)
{
    L4 t;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int q = l4_rot_src(p, KROT);
        t.w[p] = l4_rot_neg(p, KROT) ? 0u - a.w[q] : a.w[q];
    }
    r = t;
}

// r = e * 2^S, 0 < S < 24, for limbs of any size: e_i = hi_i 2^(24-S) + lo_i (floor split), so
// e_i 2^S = hi_i 2^24 + lo_i 2^S: the high parts move one limb up (the top one wraps negated).
// |r_i| < 2^24 + max|e| / 2^(24-S): the operation re-normalises.
// (a << s) + c in ONE instruction on the device (v_lshl_add_u32, 4.4 cycles against 3.75 + 3.0 for the shift and
// the add the compiler emits on its own)
FF_FN u32 l4_lshl_add(u32 a, u32 s, u32 c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    u32 d;
    asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(s), "v"(c));
    return d;
#else
    return (a << s) + c;
#endif
}
template <int S>
FF_FN u32 l4_lshl_add_c(u32 a, u32 c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    u32 d;
    asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "n"(S), "v"(c));
    return d;
#else
    return (a << S) + c;
#endif
}

// (a - b) + c for a >= b in ONE instruction on the device (v_sad_u32 = |a - b| + c)
FF_FN u32 l4_sub_add(u32 a, u32 b, u32 c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    u32 d;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
#else
    return (a - b) + c;
#endif
}

template <int S>
FF_FN void l4_shl(L4 &r, const L4 &e)
{
    static_assert(S > 0 && S < 24, "sub-limb shift");
    const u32 m = (1u << (24 - S)) - 1u;
    const u32 h0 = l4_sar(e.w[0], 24 - S), h1 = l4_sar(e.w[1], 24 - S), h2 = l4_sar(e.w[2], 24 - S),
              h3 = l4_sar(e.w[3], 24 - S);
    L4 t;
    t.w[0] = ((e.w[0] & m) << S) - h3;
    t.w[1] = l4_lshl_add_c<S>(e.w[1] & m, h0);
    t.w[2] = l4_lshl_add_c<S>(e.w[2] & m, h1);
    t.w[3] = l4_lshl_add_c<S>(e.w[3] & m, h2);
    r = t;
}

// the same with a run-time (per-lane) amount s in [0, 24): s24 = 24 - s, m = 2^(24-s) - 1.
// s = 0 is a pure normalisation (carries move up, limbs come back under 2^24).
FF_FN void l4_shl_var(L4 &r, const L4 &e, u32 s, u32 s24, u32 m)
{
    const u32 h0 = (u32)((i32)e.w[0] >> s24), h1 = (u32)((i32)e.w[1] >> s24), h2 = (u32)((i32)e.w[2] >> s24),
              h3 = (u32)((i32)e.w[3] >> s24);
    L4 t;
    t.w[0] = ((e.w[0] & m) << s) - h3;
    t.w[1] = l4_lshl_add(e.w[1] & m, s, h0);
    t.w[2] = l4_lshl_add(e.w[2] & m, s, h1);
    t.w[3] = l4_lshl_add(e.w[3] & m, s, h2);
    r = t;
}

// r = a * 2^S for any compile-time S (taken mod 192)
template <int S>
FF_FN void l4_mul_pow2(L4 &r, const L4 &a)
{
    constexpr int T = ((S % 192) + 192) % 192;
    L4 t;
    l4_rot<T / 24>(t, a);
    if constexpr (T % 24 != 0) l4_shl<T % 24>(t, t);
    r = t;
}

// Butterfly of a decimation-in-frequency pass: a' = a + b, b' = (a - b) * 2^S (S mod 192)
template <int S>
FF_FN void l4_bfly(L4 &a, L4 &b)
{
    constexpr int T = ((S % 192) + 192) % 192;
    L4 s, d;
    l4_add(s, a, b);
    l4_sub_rot<T / 24>(d, a, b);
    if constexpr (T % 24 != 0) l4_shl<T % 24>(d, d);
    a = s;
    b = d;
}

// multiplication by sqrt(2) = 2^24 - 2^72 (its square is 2^48 - 2 * 2^96 + 2^144 = 2): not used by the
// 1024-point transform, kept as the building block of 384-th roots of unity for other sizes.
FF_FN void l4_mul_sqrt2(L4 &r, const L4 &a)
{
    L4 t2, d;
    l4_rot<2>(t2, a);
    l4_sub(d, a, t2);
    l4_rot<1>(r, d);
}

// small signed integer d (|d| <= 2^9) times 2^E, E compile-time in [0, 192): ONE limb is non-zero
template <int E>
FF_FN void l4_place(L4 &r, i32 d)
{
    constexpr int T = ((E % 192) + 192) % 192;
    constexpr int k = T / 24, s = T % 24;
    static_assert(s <= 18, "the placed digit must stay inside int32");
    l4_zero(r);
    const u32 v = (u32)d << s;
    r.w[k & 3] = (k >> 2) ? 0u - v : v;
}

// ---------------------------------------------------------------------------------------------
// Conversions around the general multiplications
// ---------------------------------------------------------------------------------------------

// Offsets that make every limb positive without changing the value: sum L4_Z[i] 2^(24 i) = 0 mod P,
// L4_Z[i] >= 2^30 + 2^17 and small enough that w_i + L4_Z[i] plus the carries added in l4_to_u64 stays
// below 2^32 (found by search; the inequalities are asserted in tests/l4_checks.py).
#define L4_Z0 0x7f004083u
#define L4_Z1 0xbfbffc81u
#define L4_Z2 0x4002ff40u
#define L4_Z3 0x80000000u

// lo + 2^64 h0 -> a 64-bit representative (NOT canonical), for any 64-bit lo and h0 < 2^26:
// 2^64 = 2^32 - 1 (mod P), so s = lo + h0 (2^32 - 1); s wraps at most once and the wrapped value is below
// 2^58, so adding 2^32 - 1 for the lost 2^64 cannot wrap again.  One multiply-add with its own carry-out.
FF_FN u64 l4_fold64(u64 lo, u32 h0)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // (gfx940+: a VALU-written SGPR needs 2 wait states before a VALU reads it; the assembler does not
    // insert them inside an asm block)
    u64 s, carry;
    u32 m;
    asm("v_mad_u64_u32 %0, %1, %3, -1, %4\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32 %2, 0, -1, %1"
        : "=&v"(s), "=&s"(carry), "=v"(m)
        : "v"(h0), "v"(lo));
    return s + (u64)m;
#else
    const u64 s = lo + (u64)h0 * FF_EPS;
    return s + ((s < lo) ? FF_EPS : 0);
#endif
}

// L4 -> a 64-bit representative of the value (any of x, x + P that fits 64 bits; NOT canonical).
// Requires |w_i| <= 2^30.
//   1. u_i = w_i + Z_i > 0;
//   2. the top limb is cut to 16 bits: (u3 >> 16) 2^88 = (u3 >> 16) (2^56 - 2^24) mod P moves into
//      limbs 2 and 1, so the carry chain below cannot leave the 96-bit window;
//   3. exact base-2^24 digits by one carry chain (plain adds and shifts), packed into three words;
//   4. the top word (< 2^25) is folded by l4_fold64.
FF_FN u64 l4_to_u64(const L4 &x)
{
    const u32 u0 = x.w[0] + L4_Z0, u1 = x.w[1] + L4_Z1, u2 = x.w[2] + L4_Z2, u3 = x.w[3] + L4_Z3;
    const u32 u3h = u3 >> 16, u3l = u3 & 0xFFFFu;
    const u32 t1 = l4_sub_add(u1, u3h, u0 >> 24);          // u1 >= 2^30 > u3h
    const u32 t2 = l4_lshl_add_c<8>(u3h, u2) + (t1 >> 24);
    const u32 t3 = u3l + (t2 >> 24);                       // < 2^16 + 2^8
    const u32 p0 = l4_pack<0, 3>(t1, u0);                  // digits 0, 1 (low byte)
    const u32 p1 = l4_pack<1, 2>(t2, t1);                  // digits 1 (high 2 bytes), 2 (low 2 bytes)
    const u32 p2 = l4_pack<2, 1>(t3, t2);                  // digit 2 (top byte), 3
    return l4_fold64(((u64)p1 << 32) | p0, p2);
}

// 128-bit product hi:lo -> L4 (exact: bits 96.. re-enter negated at limb 0).
// Output: w0 in (-2^24, 2^24), w1 in (-2^8, 2^24), w2, w3 in [0, 2^24).
FF_FN void l4_from_u128(L4 &r, u64 lo, u64 hi)
{
    const u32 p0 = (u32)lo, p1 = (u32)(lo >> 32), p2 = (u32)hi, p3 = (u32)(hi >> 32);
    r.w[0] = (p0 & L4_MASK) - (p3 & L4_MASK);
    r.w[1] = l4_ext24<3>(p1, p0) - (p3 >> 24);
    r.w[2] = l4_ext24<2>(p2, p1);
    r.w[3] = p2 >> 8;
}

// 64-bit word (any representative) -> L4: w0, w1 in [0, 2^24), w2 in [0, 2^16), w3 = 0
FF_FN void l4_from_u64(L4 &r, u64 x)
{
    const u32 p0 = (u32)x, p1 = (u32)(x >> 32);
    r.w[0] = p0 & L4_MASK;
    r.w[1] = l4_ext24<3>(p1, p0);
    r.w[2] = p1 >> 16;
    r.w[3] = 0;
}

// x * t for a 64-bit factor t (table twiddle): limbs in, limbs out
FF_FN void l4_mul_u64(L4 &r, const L4 &x, u64 t)
{
    u64 lo, hi;
    ff_mul_wide(l4_to_u64(x), t, lo, hi);
    l4_from_u128(r, lo, hi);
}

// a0 * b0 + a1 * b1 [+ c] for ANY 64-bit a's and b's (c < 2^64), exactly, straight into limbs:
// the sum has at most 130 bits l0 | l1 << 32 | h0 << 64 | h1 << 96 | top << 128 (top <= 2).
// Output: w0 in (-2^24, 2^24), w1 in (-2^11, 2^24), w2, w3 in [0, 2^24).
template <bool ADD>
FF_FN void l4_dot2(L4 &r, u64 a0, u64 b0, u64 a1, u64 b1, u64 c)
{
    u64 lo0, hi0, lo1, hi1;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FF_MULWIDE_PLAIN)
    // the addend joins the first product inside its multiply-add chain (ff_mul_wide_add): a0 b0 + c is still a 128-bit number
    if constexpr (ADD) ff_mul_wide_add(a0, b0, c, lo0, hi0);
    else ff_mul_wide(a0, b0, lo0, hi0);
    ff_mul_wide(a1, b1, lo1, hi1);
    unsigned cy;
    u32 l0 = __builtin_addc((u32)lo0, (u32)lo1, 0u, &cy);
    u32 l1 = __builtin_addc((u32)(lo0 >> 32), (u32)(lo1 >> 32), cy, &cy);
    u32 h0 = __builtin_addc((u32)hi0, (u32)hi1, cy, &cy);
    u32 h1 = __builtin_addc((u32)(hi0 >> 32), (u32)(hi1 >> 32), cy, &cy);
    u32 top = cy;
#elif defined(__HIP_DEVICE_COMPILE__)
    ff_mul_wide(a0, b0, lo0, hi0);
    ff_mul_wide(a1, b1, lo1, hi1);
    unsigned cy;
    u32 l0 = __builtin_addc((u32)lo0, (u32)lo1, 0u, &cy);
    u32 l1 = __builtin_addc((u32)(lo0 >> 32), (u32)(lo1 >> 32), cy, &cy);
    u32 h0 = __builtin_addc((u32)hi0, (u32)hi1, cy, &cy);
    u32 h1 = __builtin_addc((u32)(hi0 >> 32), (u32)(hi1 >> 32), cy, &cy);
    u32 top = cy;
    if constexpr (ADD) {
        l0 = __builtin_addc(l0, (u32)c, 0u, &cy);
        l1 = __builtin_addc(l1, (u32)(c >> 32), cy, &cy);
        h0 = __builtin_addc(h0, 0u, cy, &cy);
        h1 = __builtin_addc(h1, 0u, cy, &cy);
        top += cy;
    }
#else
    ff_mul_wide(a0, b0, lo0, hi0);
    ff_mul_wide(a1, b1, lo1, hi1);
    unsigned __int128 sum = (((unsigned __int128)hi0 << 64) | lo0);
    unsigned __int128 t = sum + (((unsigned __int128)hi1 << 64) | lo1);
    u32 top = t < sum;
    if constexpr (ADD) {
        const unsigned __int128 t2 = t + c;
        top += t2 < t;
        t = t2;
    }
    const u32 l0 = (u32)t, l1 = (u32)(t >> 32), h0 = (u32)(t >> 64), h1 = (u32)(t >> 96);
#endif
    r.w[0] = (l0 & L4_MASK) - (h1 & L4_MASK);
    r.w[1] = l4_ext24<3>(l1, l0) - ((h1 >> 24) | (top << 8));
    r.w[2] = l4_ext24<2>(h0, l1);
    r.w[3] = h0 >> 8;
}

// canonical-range conversion at the end of the inverse transform (ntt.mako:402-408, ntt_cpu.py:74-80):
// x * 2^S is an integer c with |c| < 2^62 known to the caller (0 <= S < 24); returns c mod 2^32.
// Any 64-bit representative v of c is c (c >= 0, top bit clear) or c + P (top bit set), and
// P = 1 mod 2^32, so c = v.lo - v.bit63.  The sub-limb part 2^S of a twiddle is applied to the packed
// word (one 64-bit shift + one fold) instead of to the four limbs.
template <int S>
FF_FN u32 l4_to_i32_shl(const L4 &x)
{
    static_assert(S >= 0 && S < 24, "sub-limb shift");
    u64 v = l4_to_u64(x);
    if constexpr (S > 0) v = l4_fold64(v << S, (u32)(v >> (64 - S)));
    return (u32)v - (u32)(v >> 63);
}

FF_FN u32 l4_to_i32(const L4 &x) { return l4_to_i32_shl<0>(x); }
