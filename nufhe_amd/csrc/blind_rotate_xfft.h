// blind_rotate_xfft.h -- the EXACT fp64 engine for NTT-parameter keys ("exact-fft", nufhe_cloudkey_set_engine).
//
// The external product of the NTT path (nufhe/tgsw_cpu.py:82-106 with the NTT references) is nothing but the exact
// integer negacyclic convolution  sum_{m,d} digit_d(T_m) (*) BK_row[m][d][mo]  reduced modulo 2^32; the prime field is
// only the reference's means of computing it without rounding.  ANY exact method gives the same words.  This one uses
// the fp64 folded FFT-512 of fft512.h -- the widest multiplier per issue slot this chip has -- and makes it exact by
// bounding the magnitudes: every key coefficient is split into two balanced 16-bit halves
//
//     K = K_lo + 2^16 K_hi,   K_lo in [-2^15, 2^15),   K_hi in [-2^15, 2^15]
//
// whose products with the gadget digits (|d| <= 512, 4 x 1024 terms) are bounded by 2^36 instead of 2^52.  The
// worst-case fp64 error of forward transforms + multiply-accumulate + inverse transform at that magnitude is 0.037
// (212 u sqrt(512) sum_j |d_j|_2 |K_j|_2 with u = 2^-53 and the norms at their maxima), so both
// roundings deliver the exact integers for EVERY key and EVERY accumulator, and
//
//     result = round(lo) + (round(hi) << 16)   (mod 2^32)
//
// is bit-identical to the NTT kernel's output by construction (tests: every word of 4096-bit NAND / MUX gates against
// the oracle, the reference-made golden external product, all-extreme adversarial inputs).
//
// Cost per blind-rotate step: the 4 forward transforms are shared by both halves; the multiply-accumulate runs against
// two key planes and 4 inverse transforms replace 2:  8 transforms + 2 x 128 complex MACs against 6 + 128 of the plain
// FFT kernel, ~2.9 k VALU instructions per bit-iteration against the 11.3 k of the limb-form NTT kernel.
//
// Registers are the scarce resource (2 waves per SIMD = 256 per wave): 4 x 8 complex sums are 128 of them, a transform
// pair in flight another ~115.  The accumulator (32 words per lane) therefore does NOT stay in registers during the
// product: it is parked in a per-bit global buffer (8 b128 stores at the start of a step, 8 loads at its end, L2
// resident) -- the LDS has no room for it beside the two exchange buffers (kernels.hip: 8 waves x 18 KiB + tables).
#pragma once
#include "blind_rotate_fft.h"

#define BKX_POLY_ELEMS (2 * BKF_POLY_ELEMS)     /* complex per key polynomial: [half][reg 8][lane 64] */
#define BKX_ROW_ELEMS (8 * BKX_POLY_ELEMS)      /* [m][d][mo][half][reg][lane] */

struct BrXfftLds {
    cplx *xbufA;         // FFT_XBUF_ELEMS complex; bara in its row padding (brf_bara_slot)
    cplx *xbufB;         // FFT_XBUF_ELEMS complex; the accumulator mirror aliases its first 8 KiB
    const cplx *tw1;     // [512]
    const cplx *tw2;     // [64]
    BrPace pace;
    u32 *park;           // GLOBAL memory, this bit's 32 x 64 words: the accumulator during the external product
};

FF_FN BrFftLds brx_as_fft_lds(const BrXfftLds &l)
{
    BrFftLds f;
    f.xbufA = l.xbufA; f.xbufB = l.xbufB; f.park = nullptr; f.tw1 = l.tw1; f.tw2 = l.tw2; f.pace = l.pace;
    return f;
}

#if defined(NUFHE_EMU)
extern double g_emu_xfft_max_frac, g_emu_xfft_max_abs;
#endif

// low 32 bits of round-to-nearest-even(v) for |v| < 2^51: v + 1.5 * 2^52 stays in the binade [2^52, 2^53) where doubles
// ARE the integers, and the low mantissa word is the two's-complement residue.  (The plain FFT path cannot use the signed
// constant -- its sums reach 2^52, fft_round_to_u32 -- here |v| <= 2^36.)
FF_FN u32 xfft_round_lo32(double v)
{
#if defined(NUFHE_EMU)
    {
        const double f = fabs(v - nearbyint(v));
        if (f > g_emu_xfft_max_frac) g_emu_xfft_max_frac = f;
        if (fabs(v) > g_emu_xfft_max_abs) g_emu_xfft_max_abs = fabs(v);
    }
#endif
    union { double d; u64 u; } c;
    c.d = v + 6755399441055744.0;     // 1.5 * 2^52
    return (u32)c.u;
}

// MAC of the two transformed digit polynomials (d = 0, 1) of input polynomial m against both halves of
// BK_row[m][d][mo], mo = 0, 1: sum[h][mo][r] += x[d][r] * key[m][d][mo][h][r].  The 64 key loads of an m (1 KiB per load
// and wave, consecutive in memory) are software-pipelined in groups as in brf_mac_pair.
#ifndef BRX_KEY_GROUP
#define BRX_KEY_GROUP 2
#endif
#ifndef BRX_KEY_DEPTH
#define BRX_KEY_DEPTH 2
#endif
template <int D = BRX_KEY_DEPTH>
FF_FN void brx_mac_pair(cplx (&sum)[2][2][8], const cplx (&x)[2][8], const cplx *row, int m, int lane)
{
    constexpr int G = BRX_KEY_GROUP, NG = 64 / G;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef u32 brx_u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(row + (long)m * 4 * BKX_POLY_ELEMS), (short)0, 4 * BKX_POLY_ELEMS * (int)sizeof(cplx), 0x00020000);
    const u32 voff = (u32)lane * (u32)sizeof(cplx);
    auto load = [&](int gi) {
        const int byte = gi * 64 * (int)sizeof(cplx);
        union { brx_u32x4 w; cplx c; } u;
        u.w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + (u32)(byte & 4095), byte & ~4095, 0);
        return u.c;
    };
#else
    const cplx *base = row + (long)m * 4 * BKX_POLY_ELEMS + lane;
    auto load = [&](int gi) { return base[gi * 64]; };
#endif
    cplx q[D + 1][G];
#pragma unroll
    for (int j = 0; j < D; j++)
#pragma unroll
        for (int i = 0; i < G; i++) q[j][i] = load(j * G + i);
#pragma unroll
    for (int g = 0; g < NG; g++) {
        if (g + D < NG) {
#pragma unroll
            for (int i = 0; i < G; i++) q[D][i] = load((g + D) * G + i);
            BR_ISSUE_FENCE();
        }
#pragma unroll
        for (int i = 0; i < G; i++) {
            const int gi = g * G + i;                       // = ((d * 2 + mo) * 2 + h) * 8 + r
            const int r = gi & 7, h = (gi >> 3) & 1, mo = (gi >> 4) & 1, d = gi >> 5;
            c_fma_acc(sum[h][mo][r], x[d][r], q[0][i]);
        }
#pragma unroll
        for (int j = 0; j < D; j++)
#pragma unroll
            for (int i = 0; i < G; i++) q[j][i] = q[j + 1][i];
    }
}

// sum[h][mo][r] = (a_{lane + 64 r}, a_{lane + 64 r + 512}) BEFORE rounding of  sum_{m,d} digit_d(T_m) (*) K_h[m][d][mo]
template <class TW2>
FF_FN void brx_external_product_sums(cplx (&sum)[2][2][8], const u32 (&T)[2][16], const cplx *row, const BrXfftLds &lds,
                                     const TW2 &tw2, const FftLane &L)
{
    cplx *const bufs[2] = {lds.xbufA, lds.xbufB};
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int mo = 0; mo < 2; mo++)
#pragma unroll
            for (int r = 0; r < 8; r++) sum[h][mo][r] = cplx{0.0, 0.0};
#pragma unroll
    for (int m = 0; m < 2; m++) {
        cplx x[2][8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            x[0][r] = cplx{(double)br_digit<0>(T[m][r]), -(double)br_digit<0>(T[m][r + 8])};   // a_j - i a_{j+512}
            x[1][r] = cplx{(double)br_digit<1>(T[m][r]), -(double)br_digit<1>(T[m][r + 8])};
        }
        fft_forward_n<2>(x, bufs, lds.tw1, tw2, L);
        brx_mac_pair(sum, x, row, m, L.lane);
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" ::: "memory");      // keep the key loads of the next polynomial below this point (register pressure)
#endif
    }
}

// res[mo][r] = coefficient lane + 64 r of  sum_{m,d} digit_d(T_m) (*) BK_row[m][d][mo]  mod 2^32, exactly
template <class TW2>
FF_FN void brx_external_product(u32 (&res)[2][16], const u32 (&T)[2][16], const cplx *row, const BrXfftLds &lds,
                                const TW2 &tw2, const FftLane &L)
{
    cplx *const bufs[2] = {lds.xbufA, lds.xbufB};
    cplx sum[2][2][8];
    brx_external_product_sums(sum, T, row, lds, tw2, L);
    fft_inverse_2s<true>(sum[0], bufs, lds.tw1, tw2, L);
#pragma unroll
    for (int mo = 0; mo < 2; mo++)
#pragma unroll
        for (int r = 0; r < 8; r++) {
            res[mo][r] = xfft_round_lo32(sum[0][mo][r].re);
            res[mo][r + 8] = xfft_round_lo32(sum[0][mo][r].im);
        }
    fft_inverse_2s<true>(sum[1], bufs, lds.tw1, tw2, L);
#pragma unroll
    for (int mo = 0; mo < 2; mo++)
#pragma unroll
        for (int r = 0; r < 8; r++) {
            res[mo][r] += xfft_round_lo32(sum[1][mo][r].re) << 16;
            res[mo][r + 8] += xfft_round_lo32(sum[1][mo][r].im) << 16;
        }
}

// the accumulator of this lane <-> its global parking slots: word (m, r) of lane l at park[((m * 16 + r) / 4) * 256 + l * 4 + r % 4]
// (one b128 per lane and instruction, 1 KiB contiguous per wave)
struct alignas(16) brx_u4 {
    u32 x, y, z, w;
};

FF_FN void brx_park_store(u32 *park, const u32 (&acc)[2][16], int lane)
{
    brx_u4 *p = (brx_u4 *)park + lane;
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int q = 0; q < 4; q++) p[(m * 4 + q) * 64] = brx_u4{acc[m][4 * q], acc[m][4 * q + 1], acc[m][4 * q + 2], acc[m][4 * q + 3]};
}

FF_FN void brx_park_load(u32 (&acc)[2][16], const u32 *park, int lane)
{
    const brx_u4 *p = (const brx_u4 *)park + lane;
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const brx_u4 v = p[(m * 4 + q) * 64];
            acc[m][4 * q] = v.x; acc[m][4 * q + 1] = v.y; acc[m][4 * q + 2] = v.z; acc[m][4 * q + 3] = v.w;
        }
}

FF_FN void brx_init_acc(u32 (&acc)[2][16], u32 barb, i32 mu, const BrXfftLds &lds, int lane)
{
    i32 *mirror = (i32 *)lds.xbufB;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const u32 j = (u32)(lane + 64 * r);
        const u32 t = (j + barb) & 2047u;
        acc[0][r] = 0;
        acc[1][r] = (t < 1024u) ? (u32)mu : 0u - (u32)mu;
        mirror[j] = 0;
        mirror[1024 + j] = (i32)acc[1][r];
    }
    WAVE_SYNC();
}

// One blind-rotate step: ACC += BK_row (.) ((X^a - 1) ACC)  (bootstrap.py:96-109), exact
template <class TW2>
FF_FN void brx_step(u32 (&acc)[2][16], u32 a, const cplx *row, const BrXfftLds &lds, const TW2 &tw2, const FftLane &L)
{
    const int lane = L.lane;
    i32 *mirror = (i32 *)lds.xbufB;
    cplx *const bufs[2] = {lds.xbufA, lds.xbufB};
    // (X^a - 1) ACC as in brf_step
    const u32 base = (u32)lane - a;
    u32 V[2][16], sm[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const u32 t = base + 64u * (u32)r;
        V[0][r] = (u32)mirror[t & 1023u];
        V[1][r] = (u32)mirror[1024 + (t & 1023u)];
        sm[r] = (u32)((i32)(t << 21) >> 31);
    }
    BRF_SCHED_FENCE();
    u32 T[2][16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const u32 nsm = 0u - sm[r];
#pragma unroll
        for (int m = 0; m < 2; m++) T[m][r] = (V[m][r] ^ sm[r]) + (nsm - acc[m][r]);
    }
    WAVE_SYNC();    // every lane has read the mirror before buffer B is reused for exchanges
    brx_park_store(lds.park, acc, lane);
    cplx sum[2][2][8];
    brx_external_product_sums(sum, T, row, lds, tw2, L);
    fft_inverse_2s<true>(sum[0], bufs, lds.tw1, tw2, L);
    brx_park_load(acc, lds.park, lane);
#pragma unroll
    for (int mo = 0; mo < 2; mo++)
#pragma unroll
        for (int r = 0; r < 8; r++) {
            acc[mo][r] += xfft_round_lo32(sum[0][mo][r].re);
            acc[mo][r + 8] += xfft_round_lo32(sum[0][mo][r].im);
        }
    fft_inverse_2s<true>(sum[1], bufs, lds.tw1, tw2, L);
#pragma unroll
    for (int mo = 0; mo < 2; mo++)
#pragma unroll
        for (int r = 0; r < 8; r++) {
            acc[mo][r] += xfft_round_lo32(sum[1][mo][r].re) << 16;
            acc[mo][r + 8] += xfft_round_lo32(sum[1][mo][r].im) << 16;
        }
#pragma unroll
    for (int mo = 0; mo < 2; mo++)
#pragma unroll
        for (int r = 0; r < 16; r++) mirror[mo * 1024 + lane + 64 * r] = (i32)acc[mo][r];
    WAVE_SYNC();
}

FF_FN void brx_blind_rotate(u32 (&acc)[2][16], const cplx *bk, int n, u32 barb, i32 mu, const BrXfftLds &lds,
                            const FftLane &L)
{
    brx_init_acc(acc, barb, mu, lds, L.lane);
    const BrFftLds f = brx_as_fft_lds(lds);
    for (int i = 0; i < n; i++) {
        br_pace(lds.pace, (u32)i);
        const u32 a = WAVE_UNIFORM((u32)*brf_bara_slot(f, i));
        if (a == 0) continue;
        brx_step(acc, a, bk + (long)i * BKX_ROW_ELEMS, lds, lds.tw2, L);
    }
    br_pace_done(lds.pace);
}

// Key preparation: one TGSW polynomial (int32 coefficients) -> its two balanced halves, each as the (a_j, -a_{j+512})
// input of the forward transform (kernels.hip k_bkx_from_coeffs).  K = lo + 2^16 hi with lo = sign-extended low half.
FF_FN void xfft_split(i32 k, i32 &lo, i32 &hi)
{
    lo = (i32)(int16_t)(u32)k;
    hi = (i32)(((long long)k - (long long)lo) >> 16);      // in [-2^15, 2^15]
}
