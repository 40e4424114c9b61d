// ff.h -- arithmetic in GF(P), P = 2^64 - 2^32 + 1, for CDNA4 32-bit integer VALUs.
//
// Replaces the reference's finite-field Mako modules (nufhe/transform/arithmetic.mako:42-1045:
// add :78-119, sub :122-161, mul :197-333, lsh :465-1045).  Written from the behavioural spec
// (canonical representatives in [0, P); 2^64 = 2^32 - 1, 2^96 = -1, 2^192 = 1 mod P); there is
// no PTX/OpenCL flavour matrix here, only one formulation for the gfx950 integer ALU.
//
// The header also compiles with a host C++ compiler: tests/emu/ runs the wave-level kernel
// bodies lane by lane on the CPU to check them against the oracle without a GPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FF_FN __host__ __device__ __forceinline__
#define FF_HD __host__ __device__
#else
#define FF_FN static inline __attribute__((always_inline))
#define FF_HD
#endif

typedef uint64_t u64;
typedef uint32_t u32;
typedef int32_t i32;

#define FF_P 0xFFFFFFFF00000001ULL
#define FF_EPS 0xFFFFFFFFULL /* 2^64 mod P */

// a, b canonical (< P) -> canonical
FF_FN u64 ff_sub(u64 a, u64 b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // 32-bit borrow chain: the borrow of the subtraction itself selects the correction
    unsigned c1, c2, c3, c4;
    const u32 d0 = __builtin_subc((u32)a, (u32)b, 0u, &c1);
    const u32 d1 = __builtin_subc((u32)(a >> 32), (u32)(b >> 32), c1, &c2);
    const u32 m = 0u - c2;                               // borrow: + P == - EPS (mod 2^64)
    const u32 r0 = __builtin_subc(d0, m, 0u, &c3);
    const u32 r1 = __builtin_subc(d1, 0u, c3, &c4);
    return ((u64)r1 << 32) | r0;
#else
    u64 d = a - b;
    return (a < b) ? d - FF_EPS : d;   // borrow: + P == - EPS (mod 2^64)
#endif
}

// a, b canonical -> canonical: on carry-out or s >= P subtract P, i.e. add EPS modulo 2^64
// (one v_lshl_add_u64, two 64-bit compares, one select, one v_lshl_add_u64 on gfx950)
FF_FN u64 ff_add(u64 a, u64 b)
{
    const u64 s = a + b;
    const u32 m = ((s < a) | (s >= FF_P)) ? 0xFFFFFFFFu : 0u;
    return s + (u64)m;
}

FF_FN u64 ff_neg(u64 a) { return ff_sub(0, a); }

// any 64-bit value -> canonical
FF_FN u64 ff_canon(u64 x) { return x >= FF_P ? x - FF_P : x; }

// w * (2^32 - 1), canonical for any 32-bit w
FF_FN u64 ff_times_eps(u32 w) { return ((u64)w << 32) - w; }

// lo + 2^64 * h0 -> canonical, for ANY 64-bit lo and 32-bit h0.
// lo + h0 * eps <= (2^64 - 1) + (2^32 - 1)^2 = 2P - 2, so ONE conditional subtraction of P (= adding
// eps modulo 2^64) finishes the job: on gfx950 a v_mad_u64_u32, two 64-bit compares, a select and
// a 64-bit add.
FF_FN u64 ff_reduce96(u64 lo, u32 h0)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // One v_mad_u64_u32 gives s = lo + h0 * eps AND its carry-out (the compiler neither folds the
    // addend into the multiply-add nor uses its carry); mask = carry | (s >= P), as 0 / 0xFFFFFFFF.
    // Hazards (gfx940+: VALU-written SGPR read by a VALU needs 2 wait states) do not arise: the
    // VALU-written masks are only read by the scalar OR.
    u64 s, carry;
    u32 m;
    asm("v_mad_u64_u32 %0, %1, %3, -1, %4\n\t"
        "v_cmp_le_u64 vcc, %5, %0\n\t"
        "s_or_b64 vcc, vcc, %1\n\t"
        "v_cndmask_b32 %2, 0, -1, vcc"
        : "=&v"(s), "=&s"(carry), "=v"(m)
        : "v"(h0), "v"(lo), "s"((u64)FF_P)
        : "vcc", "scc");
    return s + (u64)m;
#else
    const u64 s = lo + (u64)h0 * FF_EPS;
    const u32 m = ((s < lo) | (s >= FF_P)) ? 0xFFFFFFFFu : 0u;
    return s + (u64)m;
#endif
}

// lo + 2^64 * (h0 + 2^32 h1) -> canonical    (2^64 = 2^32 - 1, 2^96 = -1)
FF_FN u64 ff_reduce128(u64 lo, u32 h0, u32 h1)
{
    // lo - h1 with the borrow fixed by + P; valid for ANY 64-bit lo (h1 < 2^32: cannot underflow again)
    return ff_reduce96(ff_sub(lo, (u64)h1), h0);
}

// 64 x 64 -> 128 as one chain of four 32 x 32 + 64 multiply-adds (v_mad_u64_u32); the low word is
// assembled from the chain's partial sums instead of a second, separate 64-bit multiply
FF_FN void ff_mul_wide(u64 a, u64 b, u64 &lo, u64 &hi)
{
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FF_MULWIDE_PLAIN)
    // (round 4) The third multiply-add takes the WHOLE second partial sum as its 64-bit addend -- no zero-extension move --
    // and the overflow of that addition comes back as the instruction's carry-out c, which becomes the top word of the last
    // addend:   u = a0 b0;  t = a1 b0 + u.hi;  w = a0 b1 + t  (mod 2^64, carry c);  hi = a1 b1 + (w.hi + 2^32 c);
    // lo = w.lo : u.lo.   4 multiply-adds + 2 moves + 1 select instead of 4 + 4 + a 64-bit add: 224 products per
    // blind-rotate iteration, VALU 12,014 -> 11,565, K1 39.3 -> 38.4 ms.  (gfx940+: the VALU-written carry needs 2 wait
    // states before a VALU reads it; nothing inserts them inside an asm block, hence the s_nop.)
    const u64 u = (u64)a0 * b0;
    const u64 t = (u64)a1 * b0 + (u >> 32);
    u64 w, carry;
    u32 c;
    asm("v_mad_u64_u32 %0, %1, %3, %4, %5\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32 %2, 0, 1, %1"
        : "=&v"(w), "=&s"(carry), "=v"(c)
        : "v"(a0), "v"(b1), "v"(t));
    hi = (u64)a1 * b1 + (((u64)c << 32) | (u32)(w >> 32));
    lo = (w << 32) | (u32)u;
#else
    const u64 u = (u64)a0 * b0;
    const u64 v = (u64)a1 * b0 + (u >> 32);
    const u64 w = (u64)a0 * b1 + (u32)v;
    hi = (u64)a1 * b1 + (v >> 32) + (w >> 32);
    lo = (w << 32) | (u32)u;
#endif
}

// a * b + c for ANY 64-bit a, b, c, exactly: (2^64 - 1)^2 + 2^64 - 1 < 2^128, so the sum has no fifth word.  On the device
// the addend rides in the FIRST multiply-add and its overflow joins the second partial sum as the top word of that
// addend (a1 b0 + u.hi + 2^32 k < 2^64: if k = 1 the wrapped u is below 2^64 - 2^33): one select instead of the four
// add-with-carry instructions of a separate 128-bit addition.
FF_FN void ff_mul_wide_add(u64 a, u64 b, u64 c, u64 &lo, u64 &hi)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FF_MULWIDE_PLAIN)
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 u, k, w, carry;
    u32 kc, cc;
    asm("v_mad_u64_u32 %0, %1, %3, %4, %5\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32 %2, 0, 1, %1"
        : "=&v"(u), "=&s"(k), "=v"(kc)
        : "v"(a0), "v"(b0), "v"(c));
    const u64 t = (u64)a1 * b0 + (((u64)kc << 32) | (u32)(u >> 32));
    asm("v_mad_u64_u32 %0, %1, %3, %4, %5\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32 %2, 0, 1, %1"
        : "=&v"(w), "=&s"(carry), "=v"(cc)
        : "v"(a0), "v"(b1), "v"(t));
    hi = (u64)a1 * b1 + (((u64)cc << 32) | (u32)(w >> 32));
    lo = (w << 32) | (u32)u;
#else
    u64 l, h;
    ff_mul_wide(a, b, l, h);
    lo = l + c;
    hi = h + (lo < l ? 1 : 0);
#endif
}

// a0 * b0 + a1 * b1 [+ c] for canonical operands with ONE reduction: the two 128-bit products (each
// < 2^128 - 2^97) and the optional canonical addend c are summed exactly in 129 bits
// lo + 2^64 (h0 + 2^32 h1) + 2^128 top, and 2^128 = -2^32 (mod P), so the top bit joins h1 as a
// 33-bit subtrahend.  Saves a reduction and a modular addition per extra product.
template <bool ADD>
FF_FN u64 ff_dot2(u64 a0, u64 b0, u64 a1, u64 b1, u64 c)
{
    u64 lo0, hi0, lo1, hi1;
    ff_mul_wide(a0, b0, lo0, hi0);
    ff_mul_wide(a1, b1, lo1, hi1);
#if defined(__HIP_DEVICE_COMPILE__)
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
    const u64 lo = ((u64)l1 << 32) | l0;
#else
    unsigned __int128 sum = (((unsigned __int128)hi0 << 64) | lo0);
    unsigned __int128 t = sum + (((unsigned __int128)hi1 << 64) | lo1);
    u32 top = t < sum;
    if constexpr (ADD) {
        const unsigned __int128 t2 = t + c;
        top += t2 < t;
        t = t2;
    }
    const u64 lo = (u64)t;
    const u32 h0 = (u32)(t >> 64), h1 = (u32)(t >> 96);
#endif
    return ff_reduce96(ff_sub(lo, ((u64)top << 32) | h1), h0);
}

// canonical * canonical -> canonical
FF_FN u64 ff_mul(u64 a, u64 b)
{
    u64 lo, hi;
    ff_mul_wide(a, b, lo, hi);
    return ff_reduce128(lo, (u32)hi, (u32)(hi >> 32));
}

// signed 32-bit integer -> canonical field element (ntt.mako:395-399)
FF_FN u64 ff_from_i32(i32 x)
{
    u64 v = (u64)(int64_t)x;
    return x < 0 ? v + FF_P : v;
}

// canonical field element -> i32: values above P/2 are negative, truncated to 32 bits
// (ntt.mako:402-408, ntt_cpu.py:74-80)
FF_FN i32 ff_to_i32(u64 v) { return (i32)((u32)v - (u32)(v > FF_P / 2)); }

// (w0 + w1 2^32 + w2 2^64) * 2^(32 Q), Q in {0,1,2}; w2 < 2^31.  -> canonical
template <int Q>
FF_FN u64 ff_place96(u32 w0, u32 w1, u32 w2)
{
    if constexpr (Q == 0) {          // (w1:w0) + w2 * eps
        return ff_reduce96(((u64)w1 << 32) | w0, w2);
    } else if constexpr (Q == 1) {   // w0 2^32 + w1 eps - w2
        return ff_reduce96(ff_sub((u64)w0 << 32, (u64)w2), w1);
    } else {               // w0 eps - (w2:w1)
        return ff_sub(ff_times_eps(w0), ((u64)w2 << 32) | w1);
    }
}

// canonical x times 2^S, S compile-time in [0, 96)
template <int S>
FF_FN u64 ff_mul_pow2_lt96(u64 x)
{
    static_assert(S >= 0 && S < 96, "shift out of range");
    if constexpr (S == 0) return x;
    constexpr int R = S % 32, Q = S / 32;
    const u32 lo = (u32)x, hi = (u32)(x >> 32);
    u32 w0, w1, w2;
    if constexpr (R == 0) {
        w0 = lo; w1 = hi; w2 = 0;
    } else {
        w0 = lo << R;
        w1 = (hi << R) | (lo >> ((32 - R) & 31));
        w2 = hi >> ((32 - R) & 31);
    }
    return ff_place96<Q>(w0, w1, w2);
}

// canonical x times 2^S mod P, S compile-time, any integer (reduced mod 192; 2^96 = -1)
// (the reference's "lsh" family, arithmetic.mako:465-1045)
template <int S>
FF_FN u64 ff_mul_pow2(u64 x)
{
    constexpr int T = ((S % 192) + 192) % 192;
    if constexpr (T >= 96) return ff_neg(ff_mul_pow2_lt96<T - 96>(x));
    else return ff_mul_pow2_lt96<T>(x);
}

// (a - b) * 2^S for canonical a, b; a twiddle 2^S with S >= 96 (mod 192) is -2^(S-96): the sign is
// absorbed by swapping the operands of the subtraction instead of negating the product.
template <int S>
FF_FN u64 ff_submul_pow2(u64 a, u64 b)
{
    constexpr int T = ((S % 192) + 192) % 192;
    if constexpr (T >= 96) return ff_mul_pow2_lt96<T - 96>(ff_sub(b, a));
    else return ff_mul_pow2_lt96<T>(ff_sub(a, b));
}

// small signed integer d in [-2^9, 2^9) (a gadget digit) times 2^S -> canonical.
// S <= 54: the product fits a signed 64-bit word, so the residue is t or t + P.
// 64 <= S <= 84: d 2^S = (d 2^(S-64)) 2^64 = u eps with u eps = (u << 32) - u still inside a signed
// 64-bit word (|u| < 2^(S-55)), again t or t + P.
template <int S>
FF_FN u64 ff_small_times_pow2(i32 d)
{
    static_assert((S >= 0 && S <= 54) || (S >= 64 && S <= 84), "shift out of range");
    if constexpr (S <= 54) {
        const u64 t = (u64)((int64_t)d << S);
        return d < 0 ? t + FF_P : t;
    } else {
        const int64_t u = (int64_t)d << (S - 64);
        const u64 t = (u64)((u << 32) - u);
        return d < 0 ? t + FF_P : t;
    }
}

// canonical x times 2^r, r a run-time (per-lane) amount in [0, 31]
FF_FN u64 ff_mul_pow2_var(u64 x, u32 r)
{
    const u32 hi = (u32)(x >> 32);
    const u64 sh = x << r;                      // low 64 bits of x * 2^r
    const u32 w2 = (hi >> 1) >> (31 - r);       // bits shifted out (r = 0 safe)
    return ff_reduce96(sh, w2);
}
