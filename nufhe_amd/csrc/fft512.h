// fft512.h -- fp64 negacyclic "folded" FFT of a degree-1024 torus polynomial: 512 complex points,
// one wavefront per polynomial, 8 complex values per lane.
//
// Replaces the reference's fft512 device module (nufhe/transform/fft.mako:18-355: 64 threads x 8
// values, three radix-8 passes, 8 barriers per transform).  Same transform as fft_transform_ref
// (nufhe/transform/fft.py:27-51, doc/source/implementation_details.rst:25-64):
//
//   forward: X_k = sum_{j<512} (a_j - i a_{j+512}) w^j W^(jk),  w = exp(-i pi/1024), W = w^4
//   inverse: a_j = round Re(y_j), a_{j+512} = round -Im(y_j),  y_j = conj(w^j)/512 sum_k X_k conj(W)^(jk)
//
// Factorisation 512 = 8 x 8 x 8 with j = j1 + 64 j2 (lane j1, register j2), k = k2 + 8 (c + 8 d):
//   pass 1 (in lane, j2 -> k2): inputs pre-multiplied by the compile-time constants w^(64 j2), 8-point DFT
//   twiddle 1 (table, 512 entries): w^(j1 (1 + 4 k2))   -- carries the per-lane part w^j1 of the fold twist
//   exchange 1 (LDS): lane (k2, a), register b, j1 = a + 8 b
//   pass 2 (in lane, b -> c): 8-point DFT;  twiddle 2 (table, 64 entries): V^(a c), V = W^8
//   exchange 2 (LDS): lane (k2, c), register a
//   pass 3 (in lane, a -> d): 8-point DFT
// The result stays in this wave layout (lane L = 8 k2 + c, register d); the FFT-domain bootstrapping
// key is stored in the same layout.  The inverse runs the passes backwards with conjugated twiddles.
// Coefficient ownership matches ntt1024.h: lane l holds coefficients l + 64 r, r = 0..15
// (r < 8: real parts, r >= 8: imaginary parts), so the blind-rotate body is shared.
//
// Exchange buffer: 8 rows x 72 complex (16-byte slots); the 8-slot row padding and the (a + c) & 7
// rotation make every ds_write_b128 / ds_read_b128 conflict-free.
#pragma once
#include <math.h>
#include <stdint.h>

#include "ff.h"
#include "ntt1024.h"   // WAVE_SYNC

struct alignas(16) cplx {
    double re, im;
};

#define FFT_ROW 72
#define FFT_XBUF_ELEMS (8 * FFT_ROW)      /* 576 complex = 9216 bytes per wave */
#define FFT_TW1_ELEMS 512                 /* [k2][j1] */
#define FFT_TW2_ELEMS 64                  /* [c][a]   */

FF_FN cplx c_add(cplx a, cplx b) { return cplx{a.re + b.re, a.im + b.im}; }
FF_FN cplx c_sub(cplx a, cplx b) { return cplx{a.re - b.re, a.im - b.im}; }
FF_FN cplx c_mul(cplx a, cplx b) { return cplx{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
FF_FN cplx c_mul_conj(cplx a, cplx b) { return cplx{a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im}; }  // a * conj(b)
FF_FN cplx c_mul_conj_negim(cplx a, cplx b) { return cplx{a.re * b.re + a.im * b.im, a.re * b.im - a.im * b.re}; }  // conj(a * conj(b))
// acc += a * b as four fused multiply-adds chained through the accumulator.  Written out because the compiler may
// contract a * b - c * d into mul + fma but may not re-associate the sum with the accumulator: `acc += a * b` compiles
// to mul, fma, add per component (6 instructions per complex term instead of 4; round 4: 414 -> 286 in the product
// phase of k_bootstrap_fft).
FF_FN void c_fma_acc(cplx &acc, cplx a, cplx b)
{
    acc.re = fma(a.re, b.re, acc.re);
    acc.re = fma(-a.im, b.im, acc.re);
    acc.im = fma(a.re, b.im, acc.im);
    acc.im = fma(a.im, b.re, acc.im);
}
FF_FN cplx c_mul_mi(cplx a) { return cplx{a.im, -a.re}; }   // a * (-i)
FF_FN cplx c_mul_pi(cplx a) { return cplx{-a.im, a.re}; }   // a * (+i)

FF_FN constexpr int br3(int i) { return ((i & 1) << 2) | (i & 2) | ((i & 4) >> 2); }

// natural-order frequency index held by (lane, reg) after fft_forward
FF_FN constexpr int fft_freq_index(int lane, int reg) { return (lane >> 3) + 8 * ((lane & 7) + 8 * reg); }

// element offset of (lane, reg) inside one FFT-domain key polynomial (wave layout): [reg][lane]
FF_FN constexpr int bkf_elem_offset(int lane, int reg) { return reg * 64 + lane; }

struct FftLane {
    int lane;
    int x1w;   // + k2 * ROW
    int x1r;   // + 8 b
    int x2a;   // lane viewed as (k2, a): k2 * ROW;  + 8 c + ((a + c) & 7)
    int a;     // lane & 7
};

FF_FN FftLane fft_lane_init(int lane)
{
    FftLane L;
    L.lane = lane;
    L.a = lane & 7;
    L.x1w = lane;
    L.x1r = (lane >> 3) * FFT_ROW + (lane & 7);
    L.x2a = (lane >> 3) * FFT_ROW;
    return L;
}

// 8-point DFT, decimation in frequency, natural-order input, bit-reversed output (x[i] = X[br3(i)]).
// INV = false: kernel exp(-2 pi i jk/8); INV = true: exp(+2 pi i jk/8) (unnormalised).
template <bool INV>
FF_FN void dft8(cplx (&x)[8])
{
    const double s = 0.70710678118654752440;
    // stage 0
    cplx u0 = c_add(x[0], x[4]), v0 = c_sub(x[0], x[4]);
    cplx u1 = c_add(x[1], x[5]), v1 = c_sub(x[1], x[5]);
    cplx u2 = c_add(x[2], x[6]), v2 = c_sub(x[2], x[6]);
    cplx u3 = c_add(x[3], x[7]), v3 = c_sub(x[3], x[7]);
    // twiddles E^j on v_j, E = exp(-+ 2 pi i/8).  E^1 and E^3 carry the factor s = 1/sqrt 2: it is NOT applied to
    // v1, v3 (2 multiplications each) but folded into the last stage, whose additions become fused multiply-adds
    // x = q +- s q' (same instruction count there, one rounding less per output)
    if (!INV) {
        v1 = cplx{v1.re + v1.im, v1.im - v1.re};                 // * (1 - i)        [* s pending]
        v2 = c_mul_mi(v2);
        v3 = cplx{v3.im - v3.re, -(v3.re + v3.im)};              // * (-1 - i)       [* s pending]
    } else {
        v1 = cplx{v1.re - v1.im, v1.re + v1.im};                 // * (1 + i)        [* s pending]
        v2 = c_mul_pi(v2);
        v3 = cplx{-(v3.re + v3.im), v3.re - v3.im};              // * (-1 + i)       [* s pending]
    }
    // stage 1
    cplx p0 = c_add(u0, u2), p2 = c_sub(u0, u2);
    cplx p1 = c_add(u1, u3), p3 = INV ? c_mul_pi(c_sub(u1, u3)) : c_mul_mi(c_sub(u1, u3));
    cplx q0 = c_add(v0, v2), q2 = c_sub(v0, v2);
    cplx q1 = c_add(v1, v3), q3 = INV ? c_mul_pi(c_sub(v1, v3)) : c_mul_mi(c_sub(v1, v3));     // [* s pending]
    // stage 2
    x[0] = c_add(p0, p1); x[1] = c_sub(p0, p1);
    x[2] = c_add(p2, p3); x[3] = c_sub(p2, p3);
    x[4] = cplx{fma(s, q1.re, q0.re), fma(s, q1.im, q0.im)};  x[5] = cplx{fma(-s, q1.re, q0.re), fma(-s, q1.im, q0.im)};
    x[6] = cplx{fma(s, q3.re, q2.re), fma(s, q3.im, q2.im)};  x[7] = cplx{fma(-s, q3.re, q2.re), fma(-s, q3.im, q2.im)};
}

// compile-time constants g^j2 = exp(-i pi j2/16), j2 = 0..7
#define FFT_G_RE(j) ((j) == 0 ? 1.0 : (j) == 1 ? 0.98078528040323044913 : (j) == 2 ? 0.92387953251128675613 : \
                     (j) == 3 ? 0.83146961230254523708 : (j) == 4 ? 0.70710678118654752440 :                  \
                     (j) == 5 ? 0.55557023301960222474 : (j) == 6 ? 0.38268343236508977173 : 0.19509032201612826785)
#define FFT_G_IM(j) ((j) == 0 ? -0.0 : (j) == 1 ? -0.19509032201612826785 : (j) == 2 ? -0.38268343236508977173 : \
                     (j) == 3 ? -0.55557023301960222474 : (j) == 4 ? -0.70710678118654752440 :                  \
                     (j) == 5 ? -0.83146961230254523708 : (j) == 6 ? -0.92387953251128675613 : -0.98078528040323044913)

// Pass 1 of the forward transform: the pre-twist g^j = exp(-i pi j / 16) = c_j (1 - i t_j), t_j = tan(pi j / 16), and the
// 8-point DFT in one, with the cosines NEVER multiplied out ("scaled" butterflies): each input is only rotated by its
// tangent (2 fused multiply-adds instead of the 4 instructions of a complex product) and carries c_j as a pending real
// scale; wherever two values with pending scales a, b meet in a butterfly, x + y becomes fma(b / a, y, x) with pending
// scale a -- the same instruction count as the plain add -- and the last stage leaves pending scale c_0 = 1.
// 14 fp64 instructions fewer per transform than pre-twist + dft8<false> (same result up to rounding: the scale ratios are
// correctly rounded constants).  Output order as dft8: x[i] = X[br3(i)].
#define FFT_TAN(j) ((j) == 1 ? 0.19891236737965800691 : (j) == 2 ? 0.41421356237309504880 : (j) == 3 ? 0.66817863791929891999 : \
                    (j) == 5 ? 1.49660576266548901760 : (j) == 6 ? 2.41421356237309504880 : 5.02733949212584810451)
FF_FN void dft8_pretwisted_fwd(cplx (&x)[8])
{
    const double s = 0.70710678118654752440;
    constexpr double c1 = FFT_G_RE(1), c2 = FFT_G_RE(2), c3 = FFT_G_RE(3), c4 = FFT_G_RE(4), c5 = FFT_G_RE(5), c6 = FFT_G_RE(6),
                     c7 = FFT_G_RE(7);
    // tangent rotations: x_j <- x_j (1 - i t_j)   [pending c_j];  t_4 = 1
    auto rot = [](cplx v, double t) { return cplx{fma(t, v.im, v.re), fma(-t, v.re, v.im)}; };
    x[1] = rot(x[1], FFT_TAN(1)); x[2] = rot(x[2], FFT_TAN(2)); x[3] = rot(x[3], FFT_TAN(3));
    x[4] = cplx{x[4].re + x[4].im, x[4].im - x[4].re};
    x[5] = rot(x[5], FFT_TAN(5)); x[6] = rot(x[6], FFT_TAN(6)); x[7] = rot(x[7], FFT_TAN(7));
    auto add = [](cplx a, double r, cplx b) { return cplx{fma(r, b.re, a.re), fma(r, b.im, a.im)}; };     // a + r b
    auto sub = [](cplx a, double r, cplx b) { return cplx{fma(-r, b.re, a.re), fma(-r, b.im, a.im)}; };   // a - r b
    // stage 0: pairs (j, j + 4); pending after it: u_j, v_j carry c_j (c_0 = 1)
    constexpr double r0 = c4, r1 = c5 / c1, r2 = c6 / c2, r3 = c7 / c3;
    cplx u0 = add(x[0], r0, x[4]), v0 = sub(x[0], r0, x[4]);
    cplx u1 = add(x[1], r1, x[5]), v1 = sub(x[1], r1, x[5]);
    cplx u2 = add(x[2], r2, x[6]), v2 = sub(x[2], r2, x[6]);
    cplx u3 = add(x[3], r3, x[7]), v3 = sub(x[3], r3, x[7]);
    // twiddles E^j of the 8-point DFT on v_j, as in dft8<false> (the 1/sqrt 2 of E^1, E^3 stays pending too)
    v1 = cplx{v1.re + v1.im, v1.im - v1.re};                 // * (1 - i)        [pending c_1 s]
    v2 = c_mul_mi(v2);                                       //                  [pending c_2]
    v3 = cplx{v3.im - v3.re, -(v3.re + v3.im)};              // * (-1 - i)       [pending c_3 s]
    // stage 1
    constexpr double r4 = c2, r5 = c3 / c1;
    cplx p0 = add(u0, r4, u2), p2 = sub(u0, r4, u2);
    cplx p1 = add(u1, r5, u3), p3 = c_mul_mi(sub(u1, r5, u3));            // [pending c_1]
    cplx q0 = add(v0, r4, v2), q2 = sub(v0, r4, v2);
    cplx q1 = add(v1, r5, v3), q3 = c_mul_mi(sub(v1, r5, v3));            // [pending c_1 s]
    // stage 2
    constexpr double r6 = c1;
    const double r7 = c1 * s;
    x[0] = add(p0, r6, p1); x[1] = sub(p0, r6, p1);
    x[2] = add(p2, r6, p3); x[3] = sub(p2, r6, p3);
    x[4] = add(q0, r7, q1); x[5] = sub(q0, r7, q1);
    x[6] = add(q2, r7, q3); x[7] = sub(q2, r7, q3);
}

// Forward transform of NX independent polynomials at once (their passes are interleaved between
// the wave-level sync points, so one LDS round trip serves NX transforms).
//   in : x[i][j2] = (a_j, -a_{j+512}) for j = lane + 64 j2   (i.e. a_j - i a_{j+512})
//   out: x[i][d]  = X_k, k = fft_freq_index(lane, d)
// xbuf[i]: private exchange buffer of transform i (FFT_XBUF_ELEMS complex each)
// twiddle 2 of this lane: from the 64-entry table, or from 8 registers loaded once (the 7 values a lane needs never
// change: V^(a c) for its a = lane & 7)
struct FftTw2Regs {
    cplx v[8];
};
FF_FN cplx fft_tw2(const cplx *tw2, int c, const FftLane &L) { return tw2[c * 8 + L.a]; }
FF_FN cplx fft_tw2(const FftTw2Regs &t, int c, const FftLane &) { return t.v[c]; }
FF_FN void fft_tw2_load(FftTw2Regs &t, const cplx *tw2, const FftLane &L)
{
#pragma unroll
    for (int c = 0; c < 8; c++) t.v[c] = tw2[c * 8 + L.a];
}

struct FftNoHook {
    FF_HD inline void operator()(int) const {}
};

// hook(0), hook(1): called once behind the writes of exchange 1 / exchange 2 -- where the wave would otherwise only wait
// for LDS (the quad kernel of the exact engine requests bootstrapping-key words there)
template <int NX, class TW2, class Hook = FftNoHook>
FF_FN void fft_forward_n(cplx (&x)[NX][8], cplx *const (&xbuf)[NX], const cplx *tw1, const TW2 &tw2,
                         const FftLane &L, Hook &&hook = Hook())
{
#pragma unroll
    for (int t = 0; t < NX; t++) dft8_pretwisted_fwd(x[t]);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int k2 = br3(i);
        const cplx w = tw1[k2 * 64 + L.lane];           // one table read serves all NX transforms
#pragma unroll
        for (int t = 0; t < NX; t++) xbuf[t][L.x1w + k2 * FFT_ROW] = c_mul(x[t][i], w);
    }
    hook(0);
    WAVE_SYNC();
#pragma unroll
    for (int t = 0; t < NX; t++)
#pragma unroll
        for (int b = 0; b < 8; b++) x[t][b] = xbuf[t][L.x1r + 8 * b];
    WAVE_SYNC();
#pragma unroll
    for (int t = 0; t < NX; t++) {
        dft8<false>(x[t]);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int c = br3(i);
            const cplx v = c == 0 ? x[t][i] : c_mul(x[t][i], fft_tw2(tw2, c, L));
            xbuf[t][L.x2a + 8 * c + ((L.a + c) & 7)] = v;
        }
    }
    hook(1);
    WAVE_SYNC();
    // now this lane is (k2, c = lane & 7); read a = 0..7
#pragma unroll
    for (int t = 0; t < NX; t++)
#pragma unroll
        for (int a = 0; a < 8; a++) x[t][a] = xbuf[t][L.x2a + 8 * L.a + ((a + L.a) & 7)];
    WAVE_SYNC();
#pragma unroll
    for (int t = 0; t < NX; t++) {
        dft8<false>(x[t]);
        cplx y[8];
#pragma unroll
        for (int i = 0; i < 8; i++) y[br3(i)] = x[t][i];
#pragma unroll
        for (int d = 0; d < 8; d++) x[t][d] = y[d];
    }
}

// Inverse transform of NX polynomials at once (includes 1/512 and the conj(w^j) untwist).
//   in : x[i][d]  = X_k, k = fft_freq_index(lane, d)
//   out: x[i][j2] = y_j for j = lane + 64 j2:  a_j = Re y_j,  a_{j+512} = -Im y_j  (before rounding)
template <int NX, class TW2>
FF_FN void fft_inverse_n(cplx (&x)[NX][8], cplx *const (&xbuf)[NX], const cplx *tw1, const TW2 &tw2,
                         const FftLane &L)
{
#pragma unroll
    for (int t = 0; t < NX; t++) {
        dft8<true>(x[t]);                                   // pass 3 inverse: d -> a
#pragma unroll
        for (int i = 0; i < 8; i++) {                       // lane = (k2, c): write a
            const int a = br3(i);
            xbuf[t][L.x2a + 8 * L.a + ((a + L.a) & 7)] = x[t][i];
        }
    }
    WAVE_SYNC();
#pragma unroll
    for (int t = 0; t < NX; t++)
#pragma unroll
        for (int c = 0; c < 8; c++) {                       // lane = (k2, a): read c, * conj(V^(a c))
            const cplx v = xbuf[t][L.x2a + 8 * c + ((L.a + c) & 7)];
            x[t][c] = c == 0 ? v : c_mul_conj(v, fft_tw2(tw2, c, L));
        }
    WAVE_SYNC();
#pragma unroll
    for (int t = 0; t < NX; t++) {
        dft8<true>(x[t]);                                   // pass 2 inverse: c -> b
#pragma unroll
        for (int i = 0; i < 8; i++) xbuf[t][L.x1r + 8 * br3(i)] = x[t][i];
    }
    WAVE_SYNC();
#pragma unroll
    for (int k2 = 0; k2 < 8; k2++) {                        // lane = j1: read k2, * conj(tw1)
        const cplx w = tw1[k2 * 64 + L.lane];               // one table read serves all NX transforms
#pragma unroll
        for (int t = 0; t < NX; t++) x[t][k2] = c_mul_conj(xbuf[t][L.x1w + k2 * FFT_ROW], w);
    }
    WAVE_SYNC();
    const double sc = 1.0 / 512.0;
#pragma unroll
    for (int t = 0; t < NX; t++) {
        dft8<true>(x[t]);                                   // pass 1 inverse: k2 -> j2
        cplx y[8];
#pragma unroll
        for (int i = 0; i < 8; i++) y[br3(i)] = x[t][i];
        x[t][0] = cplx{y[0].re * sc, y[0].im * sc};
#pragma unroll
        for (int j = 1; j < 8; j++) x[t][j] = c_mul_conj(y[j], cplx{FFT_G_RE(j) * sc, FFT_G_IM(j) * sc});
    }
}

// ---------------------------------------------------------------------------------------------
// Two INVERSE transforms half a phase apart.  fft_inverse_n<2> runs its two transforms in lock step: both compute,
// both write, both read -- and at every exchange the wave has nothing to issue until the first read returns behind
// all 16 writes (ds_write_b128: ~13 cycles each, MI355X_MICROARCH.md LDS table).  Here transform A's exchange is in
// flight while transform B computes and vice versa (same instructions, same values: only the order of independent
// work changes).  FFT_PIN() keeps the compiler from sinking the reads it has just issued below the other transform's
// arithmetic.  Measured on the wave kernel (profiles/r04_fft_experiments.txt): inverse pair staggered + shared table
// reads -1.8 %; the forward pair staggered costs 40 bytes of scratch for -0.7 % and is not used.
// ---------------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
#define FFT_PIN() asm volatile("" ::: "memory")
#else
#define FFT_PIN() ((void)0)
#endif

// NEGIM: the imaginary parts come out negated (what the coefficient layout wants: a_{j+512} = -Im y_j) -- the sign moves
// into the operands of the last multiplication instead of costing the rounding an integer instruction per value.
template <bool NEGIM = false, class TW2>
FF_FN void fft_inverse_2s(cplx (&x)[2][8], cplx *const (&xbuf)[2], const cplx *tw1, const TW2 &tw2, const FftLane &L)
{
    auto p3_write = [&](int t) {
        dft8<true>(x[t]);                                   // pass 3 inverse: d -> a
#pragma unroll
        for (int i = 0; i < 8; i++) {                       // lane = (k2, c): write a
            const int a = br3(i);
            xbuf[t][L.x2a + 8 * L.a + ((a + L.a) & 7)] = x[t][i];
        }
    };
    auto x2_read = [&](int t) {
#pragma unroll
        for (int c = 0; c < 8; c++) x[t][c] = xbuf[t][L.x2a + 8 * c + ((L.a + c) & 7)];     // lane = (k2, a): read c
    };
    auto p2_write = [&](int t) {
#pragma unroll
        for (int c = 1; c < 8; c++) x[t][c] = c_mul_conj(x[t][c], fft_tw2(tw2, c, L));      // * conj(V^(a c))
        dft8<true>(x[t]);                                   // pass 2 inverse: c -> b
#pragma unroll
        for (int i = 0; i < 8; i++) xbuf[t][L.x1r + 8 * br3(i)] = x[t][i];
    };
    auto x1_read = [&](int t) {
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++) x[t][k2] = xbuf[t][L.x1w + k2 * FFT_ROW];            // lane = j1: read k2
    };
    auto p1 = [&](int t) {
        const double sc = 1.0 / 512.0;
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++) x[t][k2] = c_mul_conj(x[t][k2], tw1[k2 * 64 + L.lane]);
        dft8<true>(x[t]);                                   // pass 1 inverse: k2 -> j2
        cplx y[8];
#pragma unroll
        for (int i = 0; i < 8; i++) y[br3(i)] = x[t][i];
        x[t][0] = cplx{y[0].re * sc, y[0].im * (NEGIM ? -sc : sc)};
#pragma unroll
        for (int j = 1; j < 8; j++) {
            const cplx g = cplx{FFT_G_RE(j) * sc, FFT_G_IM(j) * sc};
            x[t][j] = NEGIM ? c_mul_conj_negim(y[j], g) : c_mul_conj(y[j], g);
        }
    };
    p3_write(0);
    WAVE_SYNC();
    x2_read(0);
    FFT_PIN();
    p3_write(1);
    WAVE_SYNC();
    x2_read(1);
    FFT_PIN();
    p2_write(0);
    WAVE_SYNC();
    x1_read(0);
    FFT_PIN();
    p2_write(1);
    WAVE_SYNC();
    x1_read(1);
    FFT_PIN();
    p1(0);
    p1(1);
    WAVE_SYNC();
}

// single-polynomial wrappers
FF_FN void fft_forward(cplx (&x)[8], cplx *xbuf, const cplx *tw1, const cplx *tw2, const FftLane &L)
{
    cplx (&xs)[1][8] = reinterpret_cast<cplx (&)[1][8]>(x);
    cplx *const bufs[1] = {xbuf};
    fft_forward_n<1>(xs, bufs, tw1, tw2, L);
}

FF_FN void fft_inverse(cplx (&x)[8], cplx *xbuf, const cplx *tw1, const cplx *tw2, const FftLane &L)
{
    cplx (&xs)[1][8] = reinterpret_cast<cplx (&)[1][8]>(x);
    cplx *const bufs[1] = {xbuf};
    fft_inverse_n<1>(xs, bufs, tw1, tw2, L);
}

#if defined(NUFHE_EMU)
// tests/emu only: distance of the values handed to the final rounding from the nearest integer
// (0.5 = a rounding decision could flip) and their largest magnitude
extern double g_emu_fft_max_frac, g_emu_fft_max_abs;
#endif

// round-to-nearest-even, then the low 32 bits of the two's-complement integer: the reference's
// round -> int64 -> truncate to int32 (fft.mako:272-277), for every |v| <= 2^52 -- the largest magnitude the
// blind rotation can produce (4 digit polynomials x 1024 terms x 2^9 x 2^31).  The magnitude
// takes the magic add (|v| + 2^52 lies in [2^52, 2^53], where the spacing of doubles is exactly 1 and the low
// mantissa bits ARE the integer; a signed v + 1.5 * 2^52 would leave that binade from |v| = 2^51 on), the sign
// is put back on the 32-bit result (round-half-even is symmetric).
FF_FN u32 fft_round_to_u32(double v)
{
#if defined(NUFHE_EMU)
    {
        const double f = fabs(v - nearbyint(v));
        if (f > g_emu_fft_max_frac) g_emu_fft_max_frac = f;
        if (fabs(v) > g_emu_fft_max_abs) g_emu_fft_max_abs = fabs(v);
    }
#endif
    const double magic = 4503599627370496.0;   // 2^52
    union { double d; u64 u; } c, s;
    c.d = fabs(v) + magic;
    s.d = v;
    const u32 r = (u32)c.u;
    const u32 m = (u32)((i32)(u32)(s.u >> 32) >> 31);     // 0 or 0xFFFFFFFF
    return (r ^ m) - m;
}

// acc + round(v) modulo 2^32: the same rounding with the sign fix folded into the accumulation --
// (r ^ m) - m + acc = (r ^ m) + (acc - m), one subtraction and one v_xad_u32 after the sign mask.
FF_FN u32 fft_round_add_u32(u32 acc, double v)
{
#if defined(NUFHE_EMU)
    return acc + fft_round_to_u32(v);
#else
    const double magic = 4503599627370496.0;   // 2^52
    union { double d; u64 u; } c, s;
    c.d = fabs(v) + magic;
    s.d = v;
    const u32 r = (u32)c.u;
    const u32 m = (u32)((i32)(u32)(s.u >> 32) >> 31);     // 0 or 0xFFFFFFFF
#if defined(__HIP_DEVICE_COMPILE__)
    // written out: left alone the compiler turns acc - m into acc + (sign >> 31), a second shift of the same word, and
    // keeps xor and add apart
    u32 t, d;
    asm("v_sub_u32 %0, %1, %2" : "=v"(t) : "v"(acc), "v"(m));
    asm("v_xad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(r), "v"(m), "v"(t));
    return d;
#else
    return (r ^ m) + (acc - m);
#endif
#endif
}

// host-side table construction (w = exp(-i pi / 1024))
static inline void fft_make_tables(cplx *tw1, cplx *tw2)
{
    const long double pi = 3.14159265358979323846264338327950288L;
    for (int k2 = 0; k2 < 8; k2++)
        for (int j1 = 0; j1 < 64; j1++) {
            const long double ang = -pi * (long double)(j1 * (1 + 4 * k2)) / 1024.0L;
            tw1[k2 * 64 + j1] = cplx{(double)cosl(ang), (double)sinl(ang)};
        }
    for (int c = 0; c < 8; c++)
        for (int a = 0; a < 8; a++) {
            const long double ang = -2.0L * pi * (long double)(a * c) / 64.0L;
            tw2[c * 8 + a] = cplx{(double)cosl(ang), (double)sinl(ang)};
        }
}
