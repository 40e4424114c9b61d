// l4_hook.h -- TEST HOOK: one dispatcher over the primitives of ff24.h / ntt1024_l4.h, compiled into
// the device library (nufhe_l4_op) and into the CPU emulator (tests/emu), so that the same Python test
// drives the host build and the gfx950 build (v_perm_b32 selectors, the carry-out asm of l4_to_u64)
// against big-integer arithmetic.  Arrays are u32[count][4].
//   0  out = (lo, hi, 0, 0) of l4_to_u64(a)
//   1  out = l4_from_u128(lo = a[0..1], hi = a[2..3])
//   2  out = l4_mul_pow2<shift>(a), shift in [0, 192)
//   3  out = a + b, out2 = (a - b) * 2^shift   (l4_bfly, shift a multiple of 6 here: 0, 6, .. 186)
//   4  out = l4_mul_u64(a, t = b[0..1])
//   5  out = l4_dot2<true>(a[0..1], a[2..3], b[0..1], b[2..3], c[0..1])
//   6  out = (l4_to_i32(a), 0, 0, 0)
//   7  out = l4_from_u64(a[0..1])
//   8  out = a * 2^(BASE + STEP lo), lo = b[0] & 3: the 24 instantiations of the twiddle-2 layers,
//      shift = 12 * dir + 3 * hi + (q - 1), dir 0 forward / 1 inverse
//   9  out = l4_place<6 * shift>((i32)a[0]), shift in [0, 16)
//  10  out = (l4_to_i32_shl<shift>(a), 0, 0, 0), shift in {0, 6, 12, 18}: (a * 2^shift) as an int32
#pragma once
#include "ff24.h"
#include "ntt1024_l4.h"

template <int S>
struct L4ShiftDispatch {
    FF_HD static void run(L4 &r, const L4 &a, int s)
    {
        if (s == S) l4_mul_pow2<S>(r, a);
        else L4ShiftDispatch<S - 1>::run(r, a, s);
    }
};
template <>
struct L4ShiftDispatch<-1> {
    FF_HD static void run(L4 &r, const L4 &a, int) { r = a; }
};

template <int S6>
struct L4BflyDispatch {
    FF_HD static void run(L4 &a, L4 &b, int s)
    {
        if (s == 6 * S6) l4_bfly<6 * S6>(a, b);
        else L4BflyDispatch<S6 - 1>::run(a, b, s);
    }
};
template <>
struct L4BflyDispatch<-1> {
    FF_HD static void run(L4 &, L4 &, int) {}
};

template <int J2>
struct L4PlaceDispatch {
    FF_HD static void run(L4 &r, i32 d, int j2)
    {
        if (j2 == J2) l4_place<6 * J2>(r, d);
        else L4PlaceDispatch<J2 - 1>::run(r, d, j2);
    }
};
template <>
struct L4PlaceDispatch<-1> {
    FF_HD static void run(L4 &r, i32, int) { l4_zero(r); }
};

FF_FN void l4_hook_lane_tw(L4 &x, u32 lo, int which)
{
#define CASE(dir, hi, q)                                                                        \
    if (which == 12 * (dir) + 3 * (hi) + ((q)-1)) {                                              \
        if ((dir) == 0) l4_mul_pow2_lane<12 * (q) * (hi), 3 * (q)>(x, lo);                        \
        else l4_mul_pow2_lane<-12 * (q) * (hi), -3 * (q)>(x, lo);                                 \
    }
#define ROW(dir, hi) CASE(dir, hi, 1) CASE(dir, hi, 2) CASE(dir, hi, 3)
    ROW(0, 0) ROW(0, 1) ROW(0, 2) ROW(0, 3) ROW(1, 0) ROW(1, 1) ROW(1, 2) ROW(1, 3)
#undef ROW
#undef CASE
}

FF_FN void l4_hook(u32 *out, u32 *out2, const u32 *a, const u32 *b, const u32 *c, int op, int shift)
{
    L4 A, B, R;
    for (int i = 0; i < 4; i++) { A.w[i] = a[i]; B.w[i] = b ? b[i] : 0; }
    l4_zero(R);
    const u64 a01 = ((u64)a[1] << 32) | a[0], a23 = ((u64)a[3] << 32) | a[2];
    const u64 b01 = b ? (((u64)b[1] << 32) | b[0]) : 0, b23 = b ? (((u64)b[3] << 32) | b[2]) : 0;
    switch (op) {
    case 0: { const u64 v = l4_to_u64(A); R.w[0] = (u32)v; R.w[1] = (u32)(v >> 32); break; }
    case 1: l4_from_u128(R, a01, a23); break;
    case 2: L4ShiftDispatch<191>::run(R, A, shift); break;
    case 3: L4BflyDispatch<31>::run(A, B, shift); R = A; for (int i = 0; i < 4; i++) out2[i] = B.w[i]; break;
    case 4: l4_mul_u64(R, A, b01); break;
    case 5: l4_dot2<true>(R, a01, a23, b01, b23, ((u64)c[1] << 32) | c[0]); break;
    case 6: R.w[0] = l4_to_i32(A); break;
    case 7: l4_from_u64(R, a01); break;
    case 8: R = A; l4_hook_lane_tw(R, b[0] & 3u, shift); break;
    case 9: L4PlaceDispatch<15>::run(R, (i32)a[0], shift); break;
    case 10:
        R.w[0] = shift == 6 ? l4_to_i32_shl<6>(A) : shift == 12 ? l4_to_i32_shl<12>(A)
                 : shift == 18 ? l4_to_i32_shl<18>(A) : l4_to_i32_shl<0>(A);
        break;
    }
    for (int i = 0; i < 4; i++) out[i] = R.w[i];
}
