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

// The ~27 lane-dependent LDS addresses of the exchanges are loop invariants that the compiler keeps in registers across
// the product phases (and partly spills).  Re-deriving the lane record from an opaque copy of the lane index in front of
// every transform group makes them phase-local: ~30 more integer instructions per group, ~25 registers free while the
// key loads are in flight (256 VGPRs + 56 B scratch -> 252, none).
FF_FN FftLane brx_fresh_lane(const FftLane &L)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int lane = L.lane;
    asm volatile("" : "+v"(lane));
    return fft_lane_init(lane);
#else
    return L;
#endif
}
#ifndef BRX_KEY_AUX
#define BRX_KEY_AUX 0        /* cache policy bits of the key loads */
#endif
#ifndef BRX_PARK_AUX
#define BRX_PARK_AUX 17      /* cache policy bits of the parking loads / stores (bit 0 sc0, bit 1 nt, bit 4 sc1): sc0 sc1 = past the
                                vector L1, which the key stream owns (20.7 -> 19.3 ms) */
#endif
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
        u.w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + (u32)(byte & 4095), byte & ~4095, BRX_KEY_AUX);
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
                                     const TW2 &tw2, const FftLane &L BR_PROBE_ARG)
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
        fft_forward_n<2>(x, bufs, lds.tw1, tw2, brx_fresh_lane(L));
        BR_PROBE_MARK(1 + 2 * m);
        brx_mac_pair(sum, x, row, m, L.lane);
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" ::: "memory");      // keep the key loads of the next polynomial below this point (register pressure)
#endif
        BR_PROBE_MARK(2 + 2 * m);
    }
}

// res[mo][r] = coefficient lane + 64 r of  sum_{m,d} digit_d(T_m) (*) BK_row[m][d][mo]  mod 2^32, exactly
template <class TW2>
FF_FN void brx_external_product(u32 (&res)[2][16], const u32 (&T)[2][16], const cplx *row, const BrXfftLds &lds,
                                const TW2 &tw2, const FftLane &L)
{
    cplx *const bufs[2] = {lds.xbufA, lds.xbufB};
    cplx sum[2][2][8];
#if defined(BR_PROBE) && defined(__HIP_DEVICE_COMPILE__)
    BrProbe probe_ = {};
#endif
    brx_external_product_sums(sum, T, row, lds, tw2, L BR_PROBE_PASS);
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

// (device: buffer stores / loads -- the wave-uniform base lives in scalar registers, the lane contributes one 32-bit
// offset; with per-lane 64-bit pointers the compiler kept five address pairs alive across the whole product and spilled
// them: a scratch reload in front of every other parking store / load, 3 k ticks per iteration.)  `park` MUST be wave-uniform.
FF_FN void brx_park_store(u32 *park, const u32 (&acc)[2][16], int lane)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef u32 brx_u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)park, (short)0, 2048 * (int)sizeof(u32), 0x00020000);
    // offsets: the lane part in a register the compiler cannot see through (so that the constant part stays an
    // immediate of the instruction instead of being hoisted into eight loop-invariant registers), no scalar offset
    // operand -- a scalar offset register that is rewritten right behind the instruction that reads it
    // (s_movk s2 / buffer_load .. s2 / s_movk s2 ...) gave wrong words whenever two waves shared a SIMD
    u32 voff = (u32)lane * 16u;
    asm volatile("" : "+v"(voff));
    const u32 voff_hi = voff + 4096u;
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const brx_u32x4 v = {acc[m][4 * q], acc[m][4 * q + 1], acc[m][4 * q + 2], acc[m][4 * q + 3]};
            __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (m ? voff_hi : voff) + (u32)(q * 1024), 0, BRX_PARK_AUX);
        }
#else
    brx_u4 *p = (brx_u4 *)park + lane;
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int q = 0; q < 4; q++) p[(m * 4 + q) * 64] = brx_u4{acc[m][4 * q], acc[m][4 * q + 1], acc[m][4 * q + 2], acc[m][4 * q + 3]};
#endif
}

FF_FN void brx_park_load(u32 (&acc)[2][16], const u32 *park, int lane)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef u32 brx_u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)park, (short)0, 2048 * (int)sizeof(u32), 0x00020000);
    u32 voff = (u32)lane * 16u;
    asm volatile("" : "+v"(voff));
    const u32 voff_hi = voff + 4096u;
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const brx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (m ? voff_hi : voff) + (u32)(q * 1024), 0, BRX_PARK_AUX);
            acc[m][4 * q] = v.x; acc[m][4 * q + 1] = v.y; acc[m][4 * q + 2] = v.z; acc[m][4 * q + 3] = v.w;
        }
#else
    const brx_u4 *p = (const brx_u4 *)park + lane;
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const brx_u4 v = p[(m * 4 + q) * 64];
            acc[m][4 * q] = v.x; acc[m][4 * q + 1] = v.y; acc[m][4 * q + 2] = v.z; acc[m][4 * q + 3] = v.w;
        }
#endif
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
FF_FN void brx_step(u32 (&acc)[2][16], u32 a, const cplx *row, const BrXfftLds &lds, const TW2 &tw2, const FftLane &L BR_PROBE_ARG)
{
    BR_PROBE_BEGIN();
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
    BR_PROBE_MARK(0);
    brx_external_product_sums(sum, T, row, lds, tw2, L BR_PROBE_PASS);
    const FftLane Li = brx_fresh_lane(L);
    fft_inverse_2s<true>(sum[0], bufs, lds.tw1, tw2, Li);
    // the rounded low halves wait in 32 registers; the parked accumulator is requested BEHIND the second inverse pair
    // (requested earlier -- behind the first pair, in front of the second, or in front of its last pass, whole or half --
    // the 32 words do not fit beside the transform: 56-192 B of scratch, 19.9-23.6 ms; profiles/r06_xfft_experiments.txt 4c, 11)
    u32 res[2][16];
#pragma unroll
    for (int mo = 0; mo < 2; mo++)
#pragma unroll
        for (int r = 0; r < 8; r++) {
            res[mo][r] = xfft_round_lo32(sum[0][mo][r].re);
            res[mo][r + 8] = xfft_round_lo32(sum[0][mo][r].im);
        }
    BR_PROBE_MARK(5);
    fft_inverse_2s<true>(sum[1], bufs, lds.tw1, tw2, Li);
    BR_PROBE_MARK(6);
    brx_park_load(acc, lds.park, lane);
#pragma unroll
    for (int mo = 0; mo < 2; mo++)
#pragma unroll
        for (int r = 0; r < 8; r++) {
            acc[mo][r] += res[mo][r] + (xfft_round_lo32(sum[1][mo][r].re) << 16);
            acc[mo][r + 8] += res[mo][r + 8] + (xfft_round_lo32(sum[1][mo][r].im) << 16);
        }
#pragma unroll
    for (int mo = 0; mo < 2; mo++)
#pragma unroll
        for (int r = 0; r < 16; r++) mirror[mo * 1024 + lane + 64 * r] = (i32)acc[mo][r];
    WAVE_SYNC();
    BR_PROBE_MARK(7);
}

FF_FN void brx_blind_rotate(u32 (&acc)[2][16], const cplx *bk, int n, u32 barb, i32 mu, const BrXfftLds &lds,
                            const FftLane &L)
{
    brx_init_acc(acc, barb, mu, lds, L.lane);
    const BrFftLds f = brx_as_fft_lds(lds);
#if defined(BR_PROBE) && defined(__HIP_DEVICE_COMPILE__)
    BrProbe probe_ = {};
    const long long probe_w0 = clock64(), probe_r0 = wall_clock64();
#endif
    for (int i = 0; i < n; i++) {
        br_pace(lds.pace, (u32)i);
        const u32 a = WAVE_UNIFORM((u32)*brf_bara_slot(f, i));
        if (a == 0) continue;        // (X^0 - 1) ACC = 0: the external product adds nothing
        brx_step(acc, a, bk + (long)i * BKX_ROW_ELEMS, lds, lds.tw2, L BR_PROBE_PASS);
    }
    br_pace_done(lds.pace);
#if defined(BR_PROBE) && defined(__HIP_DEVICE_COMPILE__)
    if (L.lane == 0) {
        for (int i = 0; i < 8; i++) atomicAdd(&g_br_probe[i], (unsigned long long)probe_.t[i]);
        atomicAdd(&g_br_probe[14], (unsigned long long)(clock64() - probe_w0));
        atomicAdd(&g_br_probe[13], 1ull);
        atomicAdd(&g_br_probe[12], (unsigned long long)(wall_clock64() - probe_r0));   // 100 MHz
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// tlwe_mask_size = K = 2 on the exact engine (cf. brfk_* in blind_rotate_fft.h: the accumulator int32[K+1][1024] lives
// only in LDS, one wave per SIMD, 512 registers per wave).  TGSW rows have (K+1) * 2 * (K+1) = 18 polynomials, each with
// two halves: [m][d][mo][half][reg][lane].  Six digit polynomials per sum: the error bound of DESIGN.md section 7 grows
// from four terms to six, 0.037 -> 0.055, still an order below the 0.5 at which a rounding could change.
// ---------------------------------------------------------------------------------------------------------------------
template <int K>
FF_FN void brxk_mac_pair(cplx (&sum)[2][K + 1][8], const cplx (&x)[2][8], const cplx *row, int m, int lane)
{
    // load index gi = ((d * (K+1) + mo) * 2 + h) * 8 + r, consecutive in memory; groups of 4, one group ahead
    constexpr int NG = 2 * (K + 1) * 2 * 2;
    const cplx *base = row + (long)(m * 2) * (K + 1) * BKX_POLY_ELEMS + lane;
    auto addr = [&](int g, int i) { return base + (g * 4 + i) * 64; };
    cplx k[4], n[4];
#pragma unroll
    for (int i = 0; i < 4; i++) k[i] = *addr(0, i);
#pragma unroll
    for (int g = 0; g < NG; g++) {
        if (g + 1 < NG) {
#pragma unroll
            for (int i = 0; i < 4; i++) n[i] = *addr(g + 1, i);
            BR_ISSUE_FENCE();
        }
        const int h = (g >> 1) & 1, dm = g >> 2, d = dm / (K + 1), mo = dm % (K + 1);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = 4 * (g & 1) + i;
            c_fma_acc(sum[h][mo][r], x[d][r], k[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) k[i] = n[i];
    }
}

// emit(mo, r, v): v = coefficient lane + 64 r (mod 2^32, exact) of sum_{m,d} digit_d(T_m) (*) BK_row[m][d][mo];
// tsrc(m, T) fills T[r] = coefficient lane + 64 r of input polynomial m; all sources are read before the first emit
template <int K, class TSource, class Emit>
FF_FN void brxk_external_product(TSource &&tsrc, Emit &&emit, const cplx *row, const BrFftLdsK &lds, const FftLane &L)
{
    static_assert(K == 2, "pairing of the inverse transforms below is written for three polynomials");
    cplx *const bufs[2] = {lds.xbufA, lds.xbufB};
    cplx sum[2][K + 1][8];
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int mo = 0; mo <= K; mo++)
#pragma unroll
            for (int r = 0; r < 8; r++) sum[h][mo][r] = cplx{0.0, 0.0};
#pragma unroll
    for (int m = 0; m <= K; m++) {
        u32 T[16];
        tsrc(m, T);
        cplx x[2][8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            x[0][r] = cplx{(double)br_digit<0>(T[r]), -(double)br_digit<0>(T[r + 8])};   // a_j - i a_{j+512}
            x[1][r] = cplx{(double)br_digit<1>(T[r]), -(double)br_digit<1>(T[r + 8])};
        }
        fft_forward_n<2>(x, bufs, lds.tw1, lds.tw2, L);
        brxk_mac_pair<K>(sum, x, row, m, L.lane);
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" ::: "memory");
#endif
    }
    // six inverse transforms as three staggered pairs: (lo 0, lo 1), (lo 2, hi 2), (hi 0, hi 1)
    u32 res[K + 1][16];
    {
        fft_inverse_2s<true>(reinterpret_cast<cplx (&)[2][8]>(sum[0]), bufs, lds.tw1, lds.tw2, L);       // lo 0, lo 1
#pragma unroll
        for (int mo = 0; mo < 2; mo++)
#pragma unroll
            for (int r = 0; r < 8; r++) {
                res[mo][r] = xfft_round_lo32(sum[0][mo][r].re);
                res[mo][r + 8] = xfft_round_lo32(sum[0][mo][r].im);
            }
        cplx pair[2][8];
#pragma unroll
        for (int r = 0; r < 8; r++) { pair[0][r] = sum[0][2][r]; pair[1][r] = sum[1][2][r]; }
        fft_inverse_2s<true>(pair, bufs, lds.tw1, lds.tw2, L);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            res[2][r] = xfft_round_lo32(pair[0][r].re) + (xfft_round_lo32(pair[1][r].re) << 16);
            res[2][r + 8] = xfft_round_lo32(pair[0][r].im) + (xfft_round_lo32(pair[1][r].im) << 16);
        }
        fft_inverse_2s<true>(reinterpret_cast<cplx (&)[2][8]>(sum[1]), bufs, lds.tw1, lds.tw2, L);       // hi 0, hi 1
#pragma unroll
        for (int mo = 0; mo < 2; mo++)
#pragma unroll
            for (int r = 0; r < 8; r++) {
                res[mo][r] += xfft_round_lo32(sum[1][mo][r].re) << 16;
                res[mo][r + 8] += xfft_round_lo32(sum[1][mo][r].im) << 16;
            }
    }
#pragma unroll
    for (int mo = 0; mo <= K; mo++)
#pragma unroll
        for (int r = 0; r < 16; r++) emit(mo, r, res[mo][r]);
}

template <int K>
FF_FN void brxk_step(u32 a, const cplx *row, const BrFftLdsK &lds, const FftLane &L)
{
    const int lane = L.lane;
    brxk_external_product<K>(
        [&](int m, u32 (&T)[16]) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const u32 j = (u32)(lane + 64 * r);
                const u32 t = (j - a) & 2047u;
                const u32 v = (u32)lds.acc[m * 1024 + (t & 1023u)];
                const u32 self = (u32)lds.acc[m * 1024 + j];
                T[r] = ((t & 1024u) ? 0u - v : v) - self;               // polynomials_cpu.py:46-58
            }
        },
        [&](int mo, int r, u32 v) { lds.acc[mo * 1024 + lane + 64 * r] += (i32)v; }, row, lds, L);
    WAVE_SYNC();
}

// prologue + blind rotation + load of the accumulator into registers for br_extract<K> (cf. brfk_bootstrap_body)
template <int K>
FF_FN void brxk_bootstrap_body(u32 (&acc)[K + 1][16], const BrSource &s0, const BrSource &s1, i32 c0, long bit,
                               const cplx *bk, int n, i32 mu, const BrFftLdsK &lds, const FftLane &L)
{
    const int lane = L.lane;
    for (int i = lane; i < n; i += 64) {
        u32 v = 0;
        if (s0.p) v += (u32)s0.p * (u32)s0.a[bit * s0.a_stride + i];
        if (s1.p) v += (u32)s1.p * (u32)s1.a[bit * s1.a_stride + i];
        *brfk_bara_slot(lds, i) = (uint16_t)br_modswitch(v);
    }
    u32 vb = (u32)c0;
    if (s0.p) vb += (u32)s0.p * (u32)s0.b[bit * s0.b_stride];
    if (s1.p) vb += (u32)s1.p * (u32)s1.b[bit * s1.b_stride];
    const u32 barb = br_modswitch(vb);
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const u32 j = (u32)(lane + 64 * r);
        const u32 t = (j + barb) & 2047u;
#pragma unroll
        for (int m = 0; m < K; m++) lds.acc[m * 1024 + j] = 0;
        lds.acc[K * 1024 + j] = (t < 1024u) ? mu : (i32)(0u - (u32)mu);
    }
    WAVE_SYNC();
    for (int i = 0; i < n; i++) {
        const u32 a = WAVE_UNIFORM((u32)*brfk_bara_slot(lds, i));
        if (a == 0) continue;
        brxk_step<K>(a, bk + (long)i * BK_ROW_POLYS(K) * BKX_POLY_ELEMS, lds, L);
    }
#pragma unroll
    for (int m = 0; m <= K; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[m][r] = (u32)lds.acc[m * 1024 + lane + 64 * r];
}

// ---------------------------------------------------------------------------------------------------------------------
// Quad variant of the exact engine (k = 1, small batches): FOUR wavefronts per bit, one transform each way per wave.
//   forward side : wave w = 2 m + d decomposes digit d of (X^a - 1) ACC_m and transforms it -> X_w, left in its exchange
//                  buffer (idle until the inverse transform) for the three other waves
//   product side : wave w = 2 mo + h sums  X_q (.) BK_row[q >> 1][q & 1][mo][half h]  over q = 0..3 in the order of the
//                  one-wave kernel (the same fp64 values), transforms back, rounds, and adds  lo  (h = 0)  or  hi << 16
//                  (h = 1) into ACC_mo with LDS atomics (integer additions commute: the result does not depend on
//                  which wave comes first)
// Three team barriers per step: X visible | every X read (buffers free for the inverse; not needed when the inverse
// transform has an exchange buffer of its own: SPLIT, one team per work-group) | ACC complete.  No registers
// to speak of (8 complex sums), so the 32 key words of a step are requested in front of and inside the forward transform
// and have arrived when the products begin.  46 KiB of LDS per team (82 KiB with the second buffers); two teams per
// work-group at most: a third would fit the LDS but not the registers (245 of the 168 it would leave per wave).
// ---------------------------------------------------------------------------------------------------------------------
struct BrXfftQuadLds {
    cplx *xbuf;              // this wave's exchange buffer (FFT_XBUF_ELEMS complex)
    cplx *xbuf_inv;          // SPLIT: a second one for the inverse transform (then barrier 2 is not needed); else = xbuf
    const cplx *xbuf_team;   // the team's 2 (K + 1) buffers, FFT_XBUF_ELEMS apart: wave q's transformed digit polynomial
    i32 *acc;                // [K + 1][1024], shared by the team
    uint16_t *bara;          // [BR_MAX_LWE], shared
    const cplx *tw1;
    const cplx *tw2;
};

#if defined(__HIP_DEVICE_COMPILE__)
#define BRXQ_LDS_ADD(p, v) ((void)__hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
#else
#define BRXQ_LDS_ADD(p, v) ((void)(*(p) += (v)))
#endif

// tlwe_mask_size = K: NQ = 2 (K + 1) waves per bit ((K + 1) input polynomials x 2 digits = (K + 1) outputs x 2 key halves),
// key rows [m][d][mo][half][reg][lane]; six terms per sum for K = 2 (error bound 0.055, see brxk_* above)
template <int W, int K, bool SPLIT, class TeamSync>
FF_FN void brxq_step(u32 a, const cplx *row, const BrXfftQuadLds &lds, const FftLane &L, TeamSync &&team_sync BR_PROBE_ARG)
{
    BR_PROBE_BEGIN();
    constexpr int NQ = 2 * (K + 1);
    constexpr int M = W >> 1, D = W & 1;          // forward side
    constexpr int MO = W >> 1, H = W & 1;         // product side
    const int lane = L.lane;
    // key[q][r] = BK_row[q >> 1][q & 1][MO][half H], register r of this lane
    cplx key[NQ][8];
#if defined(__HIP_DEVICE_COMPILE__)
    typedef u32 brx_u32x4 __attribute__((ext_vector_type(4)));
    u32 voff = (u32)lane * (u32)sizeof(cplx);
    asm volatile("" : "+v"(voff));
    const u32 voff_hi = voff + 4096u;
    auto load_key = [&](int q) {
        // one descriptor per polynomial half (8 KiB): every offset is an instruction immediate, no scalar offset operand
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(row + (long)(q * (K + 1) + MO) * BKX_POLY_ELEMS + H * BKF_POLY_ELEMS), (short)0,
            BKF_POLY_ELEMS * (int)sizeof(cplx), 0x00020000);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            union { brx_u32x4 w; cplx c; } u;
            u.w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (r < 4 ? voff : voff_hi) + (u32)((r & 3) * 1024), 0, BRX_KEY_AUX);
            key[q][r] = u.c;
        }
    };
#else
    auto load_key = [&](int q) {
        const cplx *p = row + (long)(q * (K + 1) + MO) * BKX_POLY_ELEMS + H * BKF_POLY_ELEMS + lane;
#pragma unroll
        for (int r = 0; r < 8; r++) key[q][r] = p[r * 64];
    };
#endif
    // The 32 KiB of key words a wave needs per step (k = 1) are requested in four pieces of 8 KiB: in front of the rotation,
    // behind the two exchange writes of the forward transform, and behind the transform.  All at once in front of the
    // rotation the four waves of a team queue 128 KiB at the CU's one L2 port (64 bytes per clock) and the issue of
    // the requests itself stalls for ~2 k cycles (tools/probe_xfft.py).  (k = 2: pieces 4 and 5 two products ahead.)
    load_key(0);
    BR_ISSUE_FENCE();
    cplx x[1][8];
    {
        u32 T[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const u32 j = (u32)(lane + 64 * r);
            const u32 t = (j - a) & 2047u;
            const u32 v = (u32)lds.acc[M * 1024 + (t & 1023u)];
            T[r] = ((t & 1024u) ? 0u - v : v) - (u32)lds.acc[M * 1024 + j];     // (X^a - 1) ACC_M, polynomials_cpu.py:46-58
        }
#pragma unroll
        for (int r = 0; r < 8; r++) x[0][r] = cplx{(double)br_digit<D>(T[r]), -(double)br_digit<D>(T[r + 8])};   // a_j - i a_{j+512}
    }
    cplx *const buf1[1] = {lds.xbuf};
    BR_PROBE_MARK(0);
    fft_forward_n<1>(x, buf1, lds.tw1, lds.tw2, L, [&](int stage) {
        BR_ISSUE_FENCE();       // (fences on both sides: the requests stay where they are written)
        load_key(1 + stage);
        BR_ISSUE_FENCE();
    });
    BR_PROBE_MARK(1);
    WAVE_SYNC();        // every lane is done with the exchange buffer
#pragma unroll
    for (int r = 0; r < 8; r++) lds.xbuf[r * 64 + lane] = x[0][r];
    BR_ISSUE_FENCE();
    load_key(3);
    team_sync();        // (1) the transformed digit polynomials are visible; every wave has read ACC
    BR_PROBE_MARK(2);
    cplx sum[1][8];
#pragma unroll
    for (int r = 0; r < 8; r++) sum[0][r] = cplx{0.0, 0.0};
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        if (q >= 1 && q + 3 < NQ) load_key(q + 3);      // (k = 2; behind the fence of product q - 1: key[q - 1] is dead)
        cplx xq[8];
#pragma unroll
        for (int r = 0; r < 8; r++)     // (k = 2 re-reads its own transform: 32 registers less across the barrier)
            xq[r] = (q == W && K == 1) ? x[0][r] : lds.xbuf_team[q * FFT_XBUF_ELEMS + r * 64 + lane];
#pragma unroll
        for (int r = 0; r < 8; r++) c_fma_acc(sum[0][r], xq[r], key[q][r]);
        BR_ISSUE_FENCE();       // (one polynomial's X words in flight at a time: left alone all reads are hoisted, 96+ registers)
#if defined(__HIP_DEVICE_COMPILE__)
        // the products are finished HERE: left alone the scheduler sinks half of them below the barrier and keeps their
        // operands (key and X words) alive across it -- 50 registers spilled
#pragma unroll
        for (int r = 0; r < 8; r++) asm volatile("" : "+v"(sum[0][r].re), "+v"(sum[0][r].im));
#endif
    }
    BR_PROBE_MARK(3);
    if (!SPLIT) team_sync();        // (2) every wave has read every X: the exchange buffers are free again
    BR_PROBE_MARK(4);
    cplx *const buf2[1] = {SPLIT ? lds.xbuf_inv : lds.xbuf};
    fft_inverse_n<1>(sum, buf2, lds.tw1, lds.tw2, L);
    BR_PROBE_MARK(5);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const u32 re = xfft_round_lo32(sum[0][r].re), im = xfft_round_lo32(-sum[0][r].im);
        BRXQ_LDS_ADD((u32 *)lds.acc + MO * 1024 + lane + 64 * r, H ? re << 16 : re);
        BRXQ_LDS_ADD((u32 *)lds.acc + MO * 1024 + lane + 64 * (r + 8), H ? im << 16 : im);
    }
    BR_PROBE_MARK(6);
    team_sync();        // (3) ACC complete
    BR_PROBE_MARK(7);
}

template <int W, int K, bool SPLIT, class TeamSync>
FF_FN void brxq_bootstrap(i32 *out_a, i32 *out_b, const BrSource &s0, const BrSource &s1, i32 c0, long bit,
                          const cplx *bk, int n, i32 mu, const BrXfftQuadLds &lds, const FftLane &L, TeamSync &&team_sync)
{
    constexpr int THREADS = 64 * 2 * (K + 1);
    const int tid = 64 * W + L.lane;
    for (int i = tid; i < n; i += THREADS) {
        u32 v = 0;
        if (s0.p) v += (u32)s0.p * (u32)s0.a[bit * s0.a_stride + i];
        if (s1.p) v += (u32)s1.p * (u32)s1.a[bit * s1.a_stride + i];
        lds.bara[i] = (uint16_t)br_modswitch(v);
    }
    u32 vb = (u32)c0;
    if (s0.p) vb += (u32)s0.p * (u32)s0.b[bit * s0.b_stride];
    if (s1.p) vb += (u32)s1.p * (u32)s1.b[bit * s1.b_stride];
    const u32 barb = br_modswitch(vb);
    for (int j = tid; j < 1024; j += THREADS) {
        const u32 t = ((u32)j + barb) & 2047u;
#pragma unroll
        for (int m = 0; m < K; m++) lds.acc[m * 1024 + j] = 0;
        lds.acc[K * 1024 + j] = (t < 1024u) ? mu : (i32)(0u - (u32)mu);
    }
    WAVE_SYNC();
    team_sync();
#if defined(BR_PROBE) && defined(__HIP_DEVICE_COMPILE__)
    BrProbe probe_ = {};
    const long long probe_w0 = clock64(), probe_r0 = wall_clock64();
#endif
    for (int i = 0; i < n; i++) {
        const u32 a = WAVE_UNIFORM((u32)lds.bara[i]);
        if (a == 0) continue;        // (all waves of the team read the same word)
        brxq_step<W, K, SPLIT>(a, bk + (long)i * BK_ROW_POLYS(K) * BKX_POLY_ELEMS, lds, L, team_sync BR_PROBE_PASS);
    }
#if defined(BR_PROBE) && defined(__HIP_DEVICE_COMPILE__)
    if (L.lane == 0) {
        for (int i = 0; i < 8; i++) atomicAdd(&g_br_probe[i], (unsigned long long)probe_.t[i]);
        atomicAdd(&g_br_probe[14], (unsigned long long)(clock64() - probe_w0));
        atomicAdd(&g_br_probe[13], 1ull);
        atomicAdd(&g_br_probe[12], (unsigned long long)(wall_clock64() - probe_r0));   // 100 MHz
    }
#endif
    // sample extraction (tlwe_cpu.py:55-58)
    for (int j = tid; j < K * 1024; j += THREADS) {
        const int m = j >> 10, jj = j & 1023;
        const u32 v = (u32)lds.acc[j];
        out_a[m * 1024 + ((1024 - jj) & 1023)] = (i32)(jj == 0 ? v : 0u - v);
    }
    if (tid == 0) *out_b = lds.acc[K * 1024];
}

// Key preparation: one TGSW polynomial (int32 coefficients) -> its two balanced halves, each as the (a_j, -a_{j+512})
// input of the forward transform (kernels.hip k_bkx_from_coeffs).  K = lo + 2^16 hi with lo = sign-extended low half.
FF_FN void xfft_split(i32 k, i32 &lo, i32 &hi)
{
    lo = (i32)(int16_t)(u32)k;
    hi = (i32)(((long long)k - (long long)lo) >> 16);      // in [-2^15, 2^15]
}
