// blind_rotate_fft.h -- the fused bootstrap body with the fp64 folded-FFT transform
// (BASELINE config 5; reference: nufhe/blind_rotate.mako:18-226 rendered with
// polynomial_transform_fft, nufhe/transform/fft.mako).  Same structure, same wave-per-bit mapping
// and same coefficient ownership as blind_rotate.h; only the external product differs:
//   digits -> 4 forward FFT-512 -> complex multiply-accumulate against the FFT-domain key row
//   -> 2 inverse FFT-512 -> round to nearest -> int32          (tgsw_cpu.py:82-106 with FFT refs)
#pragma once
#include "blind_rotate.h"
#include "fft512.h"

#define BKF_POLY_ELEMS 512                 /* complex per key polynomial */
#define BKF_ROW_ELEMS (8 * BKF_POLY_ELEMS)

// Per-wave LDS (18432 bytes): two FFT exchange buffers A | B of 9216 bytes each.
//  * the accumulator mirror int32[2][1024] (8 KiB) is ALIASED onto buffer B: it is only read at
//    the start of a step (rotated reads of both polynomials into registers) and rewritten at its
//    end; in between B is an exchange buffer;
//  * bara (u16 x 512) lives in the 8 x 128 bytes of row padding of buffer A, which the exchanges
//    never touch.
//  * BRF_PARK accumulator words per lane are parked in LDS during the external product (the kernel sits at the 256-VGPR limit of 2 waves per SIMD and would otherwise spill to
//    scratch memory): 4 per lane in the row padding of buffer B (free whenever B is an exchange
//    buffer), the rest in a small extra region.
#ifndef BRF_PARK
#define BRF_PARK 7
#endif
#define BRF_PARK_EXTRA_BYTES ((BRF_PARK > 4 ? BRF_PARK - 4 : 0) * 64 * 4)
struct BrFftLds {
    cplx *xbufA;         // FFT_XBUF_ELEMS complex
    cplx *xbufB;         // FFT_XBUF_ELEMS complex; acc mirror aliases its first 8 KiB
    u32 *park;           // [(BRF_PARK - 4) * 64]
    const cplx *tw1;     // [512]
    const cplx *tw2;     // [64]
    BrPace pace;         // blind_rotate.h
};

// nothing moves across (instruction scheduling): used to issue a batch of LDS reads before the first use
#if defined(__HIP_DEVICE_COMPILE__)
#define BRF_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define BRF_SCHED_FENCE() ((void)0)
#endif

// LDS slot of parked value j (0 .. BRF_PARK-1) of this lane
FF_FN u32 *brf_park_slot(const BrFftLds &lds, int j, int lane)
{
    const int idx = j * 64 + lane;
    if (j < 4) return (u32 *)((unsigned char *)lds.xbufB + (idx >> 5) * (FFT_ROW * 16) + 1024) + (idx & 31);
    return lds.park + (idx - 256);
}

FF_FN i32 *brf_acc_mirror(const BrFftLds &lds) { return (i32 *)lds.xbufB; }

FF_FN uint16_t *brf_bara_slot(const BrFftLds &lds, int i)
{
    // u16 index i -> padding of row (i >> 6): bytes [row * 1152 + 1024, row * 1152 + 1152)
    return (uint16_t *)((unsigned char *)lds.xbufA + (i >> 6) * (FFT_ROW * 16) + 1024) + (i & 63);
}

// linear combination + mod-switch of the n mask coefficients into LDS, returns barb (cf. br_prologue)
FF_FN u32 brf_prologue(const BrSource &s0, const BrSource &s1, i32 c0, long bit, int n, const BrFftLds &lds,
                       int lane)
{
    for (int i = lane; i < n; i += 64) {
        u32 v = 0;
        if (s0.p) v += (u32)s0.p * (u32)s0.a[bit * s0.a_stride + i];
        if (s1.p) v += (u32)s1.p * (u32)s1.a[bit * s1.a_stride + i];
        *brf_bara_slot(lds, i) = (uint16_t)br_modswitch(v);
    }
    u32 vb = (u32)c0;
    if (s0.p) vb += (u32)s0.p * (u32)s0.b[bit * s0.b_stride];
    if (s1.p) vb += (u32)s1.p * (u32)s1.b[bit * s1.b_stride];
    return br_modswitch(vb);
}

// MAC of the two transformed digit polynomials (d = 0, 1) of input polynomial m against
// BK_row[m][d][mo], mo = 0, 1 (tgsw_cpu.py:63-77).  The 32 key loads (16 bytes per lane each) are
// software-pipelined in groups of BRF_KEY_GROUP: the next group is in flight while the current one is
// multiplied -- left to itself the compiler hoists all 32 loads (128 VGPRs) above the arithmetic and
// spills part of the accumulator to scratch.
#ifndef BRF_KEY_GROUP
#define BRF_KEY_GROUP 2
#endif
#ifndef BRF_KEY_DEPTH
#define BRF_KEY_DEPTH 2      /* wave kernel: groups in flight ahead of the one being multiplied: 2 = 8 more VGPRs, 255 in all, no
                                scratch, K1 12.59 -> 12.30 ms; 3 spills (72 bytes), group 4 x depth 2 is 12.20 ms with 52 bytes
                                of scratch (profiles/r03_fft_key_prefetch_variants.txt) */
#endif
template <int D = 1>
FF_FN void brf_mac_pair(cplx (&sum)[2][8], const cplx (&x)[2][8], const cplx *row, int m, int lane)
{
    // group g: polynomial (m, d, mo) = g / GROUPS_PER_POLY, registers GROUP * (g % GROUPS_PER_POLY) ...
    constexpr int GPP = 8 / BRF_KEY_GROUP, NG = 4 * GPP;
#if defined(__HIP_DEVICE_COMPILE__)
    // buffer loads: the wave-uniform part of the address (row, polynomial, 4 KiB window) lives in scalar registers and is
    // added by the scalar unit, the lane contributes a 32-bit offset -- a per-lane 64-bit pointer costs a v_add_co / v_addc
    // pair for every 4 KiB the 12-bit immediate offset cannot reach (30 VALU instructions per iteration)
    typedef u32 brf_u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(row + (long)(m * 2) * 2 * BKF_POLY_ELEMS), (short)0, 4 * BKF_POLY_ELEMS * (int)sizeof(cplx), 0x00020000);
    const u32 voff = (u32)lane * (u32)sizeof(cplx);
    auto load = [&](int g, int i) {
        const int poly = g / GPP, r = BRF_KEY_GROUP * (g % GPP) + i;    // poly = d * 2 + mo
        const int byte = (poly * BKF_POLY_ELEMS + r * 64) * (int)sizeof(cplx);
        union { brf_u32x4 w; cplx c; } u;
        u.w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + (u32)(byte & 4095), byte & ~4095, 0);
        return u.c;
    };
#else
    const cplx *base = row + (long)(m * 2) * 2 * BKF_POLY_ELEMS + lane;
    auto load = [&](int g, int i) {
        const int poly = g / GPP, r = BRF_KEY_GROUP * (g % GPP) + i;    // poly = d * 2 + mo
        return base[poly * BKF_POLY_ELEMS + r * 64];
    };
#endif
    cplx q[D + 1][BRF_KEY_GROUP];          // ring of pending groups: q[j] belongs to group g + j
#pragma unroll
    for (int j = 0; j < D; j++)
#pragma unroll
        for (int i = 0; i < BRF_KEY_GROUP; i++) q[j][i] = load(j, i);
#pragma unroll
    for (int g = 0; g < NG; g++) {
        if (g + D < NG) {
#pragma unroll
            for (int i = 0; i < BRF_KEY_GROUP; i++) q[D][i] = load(g + D, i);
            BR_ISSUE_FENCE();
        }
        const int d = (g / GPP) >> 1, mo = (g / GPP) & 1;
#pragma unroll
        for (int i = 0; i < BRF_KEY_GROUP; i++) {
            const int r = BRF_KEY_GROUP * (g % GPP) + i;
            c_fma_acc(sum[mo][r], x[d][r], q[0][i]);
        }
#pragma unroll
        for (int j = 0; j < D; j++)
#pragma unroll
            for (int i = 0; i < BRF_KEY_GROUP; i++) q[j][i] = q[j + 1][i];
    }
}

// sum[mo][r] = (a_{lane + 64 r}, -a_{lane + 64 r + 512}) BEFORE rounding of  sum_{m,d} digit_d(T_m) (*) BK_row[m][d][mo];
// T[m][r] = coefficient lane + 64 r of input polynomial m.  The two digit polynomials of each m are
// transformed together, and so are the two output polynomials (fft_*_n<2>).
template <class TW2>
FF_FN void brf_external_product_sums(cplx (&sum)[2][8], const u32 (&T)[2][16], const cplx *row, const BrFftLds &lds,
                                     const TW2 &tw2, const FftLane &L BR_PROBE_ARG)
{
    cplx *const bufs[2] = {lds.xbufA, lds.xbufB};
#pragma unroll
    for (int r = 0; r < 8; r++) { sum[0][r] = cplx{0.0, 0.0}; sum[1][r] = cplx{0.0, 0.0}; }
#pragma unroll
    for (int m = 0; m < 2; m++) {
        cplx x[2][8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            x[0][r] = cplx{(double)br_digit<0>(T[m][r]), -(double)br_digit<0>(T[m][r + 8])};   // a_j - i a_{j+512}
            x[1][r] = cplx{(double)br_digit<1>(T[m][r]), -(double)br_digit<1>(T[m][r + 8])};
        }
        fft_forward_n<2>(x, bufs, lds.tw1, tw2, L);
        BR_PROBE_MARK(2 * m);
        brf_mac_pair<BRF_KEY_DEPTH>(sum, x, row, m, L.lane);       // (the 2-waves-per-bit kernel keeps depth 1: no registers to spare)
#if defined(__HIP_DEVICE_COMPILE__)
        // keep the key loads of the next polynomial below this point (register pressure)
        asm volatile("" ::: "memory");
#endif
        BR_PROBE_MARK(2 * m + 1);
    }
    fft_inverse_2s<true>(sum, bufs, lds.tw1, tw2, L);
    BR_PROBE_MARK(4);
}

// res[mo][r] = coefficient lane + 64 r of  sum_{m,d} digit_d(T_m) (*) BK_row[m][d][mo]  (tgsw_cpu.py:82-106)
template <class TW2>
FF_FN void brf_external_product(u32 (&res)[2][16], const u32 (&T)[2][16], const cplx *row, const BrFftLds &lds,
                                const TW2 &tw2, const FftLane &L BR_PROBE_ARG)
{
    cplx sum[2][8];
    brf_external_product_sums(sum, T, row, lds, tw2, L BR_PROBE_PASS);
#pragma unroll
    for (int mo = 0; mo < 2; mo++)
#pragma unroll
        for (int r = 0; r < 8; r++) {
            res[mo][r] = fft_round_to_u32(sum[mo][r].re);
            res[mo][r + 8] = fft_round_to_u32(sum[mo][r].im);
        }
}

FF_FN void brf_init_acc(u32 (&acc)[2][16], u32 barb, i32 mu, const BrFftLds &lds, int lane)
{
    i32 *mirror = brf_acc_mirror(lds);
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

// One blind-rotate step: ACC += BK_row (.) ((X^a - 1) ACC)  (bootstrap.py:96-109)
template <class TW2>
FF_FN void brf_step(u32 (&acc)[2][16], u32 a, const cplx *row, const BrFftLds &lds, const TW2 &tw2, const FftLane &L BR_PROBE_ARG)
{
    BR_PROBE_BEGIN();
    const int lane = L.lane;
    i32 *mirror = brf_acc_mirror(lds);
    // (X^a - 1) ACC, polynomials_cpu.py:46-58: coefficient j comes from j - a mod 2048, negated when that wraps (bit 10).
    // All 16 paired reads are issued before the first one is used (left alone the compiler waits for each in turn:
    // 16 LDS round trips in a row); sign mask s = 0 / -1: +-v - acc = (v ^ s) + (-s - acc), one v_xad_u32.
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
    // park BRF_PARK accumulator words in LDS for the duration of the product (they are only needed
    // again for the update below)
#pragma unroll
    for (int r = 0; r < BRF_PARK; r++) *brf_park_slot(lds, r, lane) = acc[0][r];
    cplx sum[2][8];
    BR_PROBE_MARK(5);
    brf_external_product_sums(sum, T, row, lds, tw2, L BR_PROBE_PASS);
#pragma unroll
    for (int r = 0; r < BRF_PARK; r++) acc[0][r] = *brf_park_slot(lds, r, lane);
#pragma unroll
    for (int mo = 0; mo < 2; mo++)
#pragma unroll
        for (int r = 0; r < 8; r++) {
            acc[mo][r] = fft_round_add_u32(acc[mo][r], sum[mo][r].re);           // rounding and accumulation in one
            acc[mo][r + 8] = fft_round_add_u32(acc[mo][r + 8], sum[mo][r].im);   // (imaginary parts arrive negated)
        }
#pragma unroll
    for (int mo = 0; mo < 2; mo++)
#pragma unroll
        for (int r = 0; r < 16; r++) mirror[mo * 1024 + lane + 64 * r] = (i32)acc[mo][r];
    WAVE_SYNC();
    BR_PROBE_MARK(6);
}

FF_FN void brf_blind_rotate(u32 (&acc)[2][16], const cplx *bk, int n, u32 barb, i32 mu, const BrFftLds &lds,
                            const FftLane &L)
{
    brf_init_acc(acc, barb, mu, lds, L.lane);
    FftTw2Regs tw2;           // 28 registers the kernel has to spare; saves the 35 LDS reads per iteration (K1 -2 %)
    fft_tw2_load(tw2, lds.tw2, L);
#if defined(BR_PROBE) && defined(__HIP_DEVICE_COMPILE__)
    BrProbe probe_ = {};
    const long long probe_w0 = clock64(), probe_r0 = wall_clock64();
#endif
    for (int i = 0; i < n; i++) {
        br_pace(lds.pace, (u32)i);
        const u32 a = WAVE_UNIFORM((u32)*brf_bara_slot(lds, i));
        if (a == 0) continue;
        brf_step(acc, a, bk + (long)i * BKF_ROW_ELEMS, lds, tw2, L BR_PROBE_PASS);
    }
    br_pace_done(lds.pace);
#if defined(BR_PROBE) && defined(__HIP_DEVICE_COMPILE__)
    if (L.lane == 0) {
        for (int i = 0; i < 8; i++) atomicAdd(&g_br_probe[i], (unsigned long long)probe_.t[i]);
        atomicAdd(&g_br_probe[14], (unsigned long long)(clock64() - probe_w0));
        atomicAdd(&g_br_probe[13], 1ull);
        atomicAdd(&g_br_probe[12], (unsigned long long)(wall_clock64() - probe_r0));
        const unsigned gw = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & 8191u;
        g_br_probe_life[2 * gw] = (unsigned)probe_r0;
        g_br_probe_life[2 * gw + 1] = (unsigned)wall_clock64();
    }
#endif
}

// ------------------------------------------------------------------------------------------
// tlwe_mask_size = K > 1 with the FFT transform (the reference runs this pair through its multi-kernel
// driver, nufhe/bootstrap.py:96-142 with polynomial_transform_fft; test/test_gates.py:88-100).
// Same body as above with K + 1 polynomials: TGSW rows have (K+1) * 2 * (K+1) polynomials, K + 1 pairs of
// forward transforms and K + 1 inverse transforms per step.  The accumulator lives ONLY in its LDS
// mirror int32[K+1][1024] (not aliased: the registers are needed for the (K+1) x 8 complex sums), so a
// wave needs 2 x 9216 + (K+1) x 4096 bytes of LDS; the kernel runs 4 waves per CU (one per SIMD, see kernels.hip).
// ------------------------------------------------------------------------------------------
struct BrFftLdsK {
    cplx *xbufA;         // FFT_XBUF_ELEMS complex; bara lives in its row padding (brf_bara_slot)
    cplx *xbufB;         // FFT_XBUF_ELEMS complex
    i32 *acc;            // [K+1][1024]
    const cplx *tw1;
    const cplx *tw2;
};

FF_FN uint16_t *brfk_bara_slot(const BrFftLdsK &lds, int i)
{
    return (uint16_t *)((unsigned char *)lds.xbufA + (i >> 6) * (FFT_ROW * 16) + 1024) + (i & 63);
}

// sum[mo] += x[d] (*) BK_row[m][d][mo] for d = 0, 1 and mo = 0..K; key loads pipelined in groups of 4
template <int K>
FF_FN void brfk_mac_pair(cplx (&sum)[K + 1][8], const cplx (&x)[2][8], const cplx *row, int m, int lane)
{
    constexpr int NG = 2 * (K + 1) * 2;            // groups of 4 registers: (d, mo, half)
    const cplx *base = row + (long)(m * 2) * (K + 1) * BKF_POLY_ELEMS + lane;
    auto addr = [&](int g, int i) { return base + (g >> 1) * BKF_POLY_ELEMS + (4 * (g & 1) + i) * 64; };   // g >> 1 = d (K+1) + mo
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
        const int d = (g >> 1) / (K + 1), mo = (g >> 1) % (K + 1);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = 4 * (g & 1) + i;
            c_fma_acc(sum[mo][r], x[d][r], k[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) k[i] = n[i];
    }
}

// emit(mo, r, v): v = coefficient lane + 64 r (mod 2^32) of sum_{m,d} digit_d(T_m) (*) BK_row[m][d][mo];
// tsrc(m, T) fills T[r] = coefficient lane + 64 r of input polynomial m; all sources are read before
// the first emit.
template <int K, class TSource, class Emit>
FF_FN void brfk_external_product(TSource &&tsrc, Emit &&emit, const cplx *row, const BrFftLdsK &lds, const FftLane &L)
{
    cplx *const bufs[2] = {lds.xbufA, lds.xbufB};
    cplx sum[K + 1][8];
#pragma unroll
    for (int mo = 0; mo <= K; mo++)
#pragma unroll
        for (int r = 0; r < 8; r++) sum[mo][r] = cplx{0.0, 0.0};
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
        brfk_mac_pair<K>(sum, x, row, m, L.lane);
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" ::: "memory");
#endif
    }
    static_assert(K == 2, "pairing of the inverse transforms below is written for three polynomials");
    {
        cplx (&pair)[2][8] = reinterpret_cast<cplx (&)[2][8]>(sum[0]);
        fft_inverse_n<2>(pair, bufs, lds.tw1, lds.tw2, L);
        fft_inverse(sum[2], lds.xbufA, lds.tw1, lds.tw2, L);
    }
#pragma unroll
    for (int mo = 0; mo <= K; mo++)
#pragma unroll
        for (int r = 0; r < 8; r++) {
            emit(mo, r, fft_round_to_u32(sum[mo][r].re));
            emit(mo, r + 8, fft_round_to_u32(-sum[mo][r].im));
        }
}

// One blind-rotate step on the LDS-resident accumulator: ACC += BK_row (.) ((X^a - 1) ACC)
template <int K>
FF_FN void brfk_step(u32 a, const cplx *row, const BrFftLdsK &lds, const FftLane &L)
{
    const int lane = L.lane;
    brfk_external_product<K>(
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

// prologue + blind rotation + load of the accumulator into registers for br_extract<K>
template <int K>
FF_FN void brfk_bootstrap_body(u32 (&acc)[K + 1][16], const BrSource &s0, const BrSource &s1, i32 c0, long bit,
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
    // ACC = (0, ..., 0, X^(2N - barb) * mu)  (bootstrap.py:176-182)
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
        brfk_step<K>(a, bk + (long)i * BK_ROW_POLYS(K) * BKF_POLY_ELEMS, lds, L);
    }
#pragma unroll
    for (int m = 0; m <= K; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[m][r] = (u32)lds.acc[m * 1024 + lane + 64 * r];
}

// ------------------------------------------------------------------------------------------
// Team variant of the FFT path (small batches): four wavefronts share one bit exactly as in
// blind_rotate.h (brt_*): wave w = 2 m + d transforms digit d of polynomial m and multiplies it with
// BK_row[m][d][0..1]; the partial sums meet in LDS; waves 0/1 add them in the fixed order w = 0..3,
// transform back, round and update the shared accumulator.  The summation order differs from the
// one-wave kernel (fp64 addition is not associative), which is covered by the path's stated
// tolerance; on every tested input the rounded results are identical.
// ------------------------------------------------------------------------------------------
#define BRFT_PART_ELEMS (BRT_WAVES * 2 * BKF_POLY_ELEMS)   /* complex: partial sums [wave][mo][reg][lane] */

struct BrFftTeamLds {
    cplx *xbuf;          // this wave's exchange buffer (FFT_XBUF_ELEMS complex)
    i32 *acc;            // [2][1024] accumulator shared by the team
    uint16_t *bara;      // [BR_MAX_LWE]
    cplx *part;          // [BRFT_PART_ELEMS]
    const cplx *tw1;
    const cplx *tw2;
};

template <class TeamSync>
FF_FN void brft_step(u32 a, const cplx *row, const BrFftTeamLds &lds, const FftLane &L, int w, TeamSync &&team_sync)
{
    const int lane = L.lane;
    const int m = w >> 1;
    const int sh = (w & 1) ? 12 : 22;      // digit d = w & 1
    double dg[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const u32 j = (u32)(lane + 64 * r);
        const u32 t = (j - a) & 2047u;
        const u32 v = (u32)lds.acc[m * 1024 + (t & 1023u)];
        const u32 T = ((t & 1024u) ? 0u - v : v) - (u32)lds.acc[m * 1024 + j];
        dg[r] = (double)((((i32)(T + TGSW_OFFSET) >> sh) & 1023) - 512);
    }
    cplx x[8];
#pragma unroll
    for (int r = 0; r < 8; r++) x[r] = cplx{dg[r], -dg[r + 8]};                // a_j - i a_{j+512}
    fft_forward(x, lds.xbuf, lds.tw1, lds.tw2, L);
    const cplx *poly = row + (long)w * 2 * BKF_POLY_ELEMS + lane;
#pragma unroll
    for (int mo = 0; mo < 2; mo++)
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const cplx k = poly[mo * BKF_POLY_ELEMS + r * 64];
            cplx p;
            p.re = x[r].re * k.re - x[r].im * k.im;
            p.im = x[r].re * k.im + x[r].im * k.re;
            lds.part[((w * 2 + mo) * 8 + r) * 64 + lane] = p;
        }
    team_sync();
    if (w < 2) {
        cplx sum[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            cplx acc = lds.part[((0 * 2 + w) * 8 + r) * 64 + lane];
#pragma unroll
            for (int src = 1; src < BRT_WAVES; src++) {
                const cplx p = lds.part[((src * 2 + w) * 8 + r) * 64 + lane];
                acc.re += p.re;
                acc.im += p.im;
            }
            sum[r] = acc;
        }
        fft_inverse(sum, lds.xbuf, lds.tw1, lds.tw2, L);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            lds.acc[w * 1024 + lane + 64 * r] += (i32)fft_round_to_u32(sum[r].re);
            lds.acc[w * 1024 + lane + 64 * (r + 8)] += (i32)fft_round_to_u32(-sum[r].im);
        }
    }
    team_sync();
}

template <class TeamSync>
FF_FN void brft_bootstrap(i32 *out_a, i32 *out_b, const BrSource &s0, const BrSource &s1, i32 c0, long bit,
                          const cplx *bk, int n, i32 mu, const BrFftTeamLds &lds, const FftLane &L, int w,
                          TeamSync &&team_sync)
{
    const int tid = 64 * w + L.lane;
    for (int i = tid; i < n; i += 64 * BRT_WAVES) {
        u32 v = 0;
        if (s0.p) v += (u32)s0.p * (u32)s0.a[bit * s0.a_stride + i];
        if (s1.p) v += (u32)s1.p * (u32)s1.a[bit * s1.a_stride + i];
        lds.bara[i] = (uint16_t)br_modswitch(v);
    }
    u32 vb = (u32)c0;
    if (s0.p) vb += (u32)s0.p * (u32)s0.b[bit * s0.b_stride];
    if (s1.p) vb += (u32)s1.p * (u32)s1.b[bit * s1.b_stride];
    const u32 barb = br_modswitch(vb);
    for (int j = tid; j < 1024; j += 64 * BRT_WAVES) {
        const u32 t = ((u32)j + barb) & 2047u;
        lds.acc[j] = 0;
        lds.acc[1024 + j] = (t < 1024u) ? mu : (i32)(0u - (u32)mu);
    }
    team_sync();
    for (int i = 0; i < n; i++) {
        const u32 a = WAVE_UNIFORM((u32)lds.bara[i]);
        if (a == 0) continue;
        brft_step(a, bk + (long)i * BKF_ROW_ELEMS, lds, L, w, team_sync);
    }
    for (int j = tid; j < 1024; j += 64 * BRT_WAVES) {
        const u32 v = (u32)lds.acc[j];
        out_a[(1024 - j) & 1023] = (i32)(j == 0 ? v : 0u - v);
    }
    if (tid == 0) *out_b = lds.acc[1024];
}

// ------------------------------------------------------------------------------------------
// Team variant for tlwe_mask_size = K > 1 with the FFT transform (small batches): K + 1 wavefronts per bit, as in
// blind_rotate.h (brtk_*): wave m owns input polynomial m -- its two digit transforms (interleaved, fft_forward_n<2>)
// and their products with BK_row[m][0..1][0..K] --, the (K+1) x (K+1) partial sums meet in LDS, wave mo adds the
// K + 1 partial sums of output polynomial mo in the fixed order m = 0..K, transforms back, rounds and updates
// ACC[mo].  2 forward + 1 inverse transform on the critical path instead of 2 (K+1) + (K+1).  The summation order
// differs from the one-wave kernel (fp64 addition is not associative): covered by the path's stated tolerance, and
// the rounded results are identical on every tested input.
// ------------------------------------------------------------------------------------------
#define BRFTK_PART_ELEMS(K) (((K) + 1) * ((K) + 1) * BKF_POLY_ELEMS)   /* complex: partial sums [wave m][mo][reg][lane] */

struct BrFftTeamLdsK {
    cplx *xbufA;         // this wave's two exchange buffers (FFT_XBUF_ELEMS complex each)
    cplx *xbufB;
    i32 *acc;            // [K+1][1024], shared by the team
    uint16_t *bara;      // [BR_MAX_LWE], shared
    cplx *part;          // [BRFTK_PART_ELEMS(K)], shared
    const cplx *tw1;
    const cplx *tw2;
};

template <int K, class TeamSync>
FF_FN void brftk_step(u32 a, const cplx *row, const BrFftTeamLdsK &lds, const FftLane &L, int w, TeamSync &&team_sync)
{
    const int lane = L.lane;
    cplx *const bufs[2] = {lds.xbufA, lds.xbufB};
    {
        u32 T[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const u32 j = (u32)(lane + 64 * r);
            const u32 t = (j - a) & 2047u;
            const u32 v = (u32)lds.acc[w * 1024 + (t & 1023u)];
            T[r] = ((t & 1024u) ? 0u - v : v) - (u32)lds.acc[w * 1024 + j];     // (X^a - 1) ACC_w
        }
        cplx x[2][8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            x[0][r] = cplx{(double)br_digit<0>(T[r]), -(double)br_digit<0>(T[r + 8])};   // a_j - i a_{j+512}
            x[1][r] = cplx{(double)br_digit<1>(T[r]), -(double)br_digit<1>(T[r + 8])};
        }
        fft_forward_n<2>(x, bufs, lds.tw1, lds.tw2, L);
        cplx ps[K + 1][8];
#pragma unroll
        for (int mo = 0; mo <= K; mo++)
#pragma unroll
            for (int r = 0; r < 8; r++) ps[mo][r] = cplx{0.0, 0.0};
        brfk_mac_pair<K>(ps, x, row, w, lane);
#pragma unroll
        for (int mo = 0; mo <= K; mo++)
#pragma unroll
            for (int r = 0; r < 8; r++) lds.part[((w * (K + 1) + mo) * 8 + r) * 64 + lane] = ps[mo][r];
    }
    team_sync();
    cplx sum[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        cplx acc = lds.part[((0 * (K + 1) + w) * 8 + r) * 64 + lane];
#pragma unroll
        for (int src = 1; src <= K; src++) {
            const cplx p = lds.part[((src * (K + 1) + w) * 8 + r) * 64 + lane];
            acc.re += p.re;
            acc.im += p.im;
        }
        sum[r] = acc;
    }
    fft_inverse(sum, lds.xbufA, lds.tw1, lds.tw2, L);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        lds.acc[w * 1024 + lane + 64 * r] += (i32)fft_round_to_u32(sum[r].re);
        lds.acc[w * 1024 + lane + 64 * (r + 8)] += (i32)fft_round_to_u32(-sum[r].im);
    }
    team_sync();
}

// Whole bootstrap body of the (K+1)-wave FFT team for one bit; out_a has K * 1024 entries.
template <int K, class TeamSync>
FF_FN void brftk_bootstrap(i32 *out_a, i32 *out_b, const BrSource &s0, const BrSource &s1, i32 c0, long bit,
                           const cplx *bk, int n, i32 mu, const BrFftTeamLdsK &lds, const FftLane &L, int w,
                           TeamSync &&team_sync)
{
    constexpr int THREADS = 64 * (K + 1);
    const int tid = 64 * w + L.lane;
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
    // ACC = (0, ..., 0, X^(2N - barb) * mu)  (bootstrap.py:176-182)
    for (int j = tid; j < 1024; j += THREADS) {
        const u32 t = ((u32)j + barb) & 2047u;
#pragma unroll
        for (int m = 0; m < K; m++) lds.acc[m * 1024 + j] = 0;
        lds.acc[K * 1024 + j] = (t < 1024u) ? mu : (i32)(0u - (u32)mu);
    }
    team_sync();
    for (int i = 0; i < n; i++) {
        const u32 a = WAVE_UNIFORM((u32)lds.bara[i]);
        if (a == 0) continue;
        brftk_step<K>(a, bk + (long)i * BK_ROW_POLYS(K) * BKF_POLY_ELEMS, lds, L, w, team_sync);
    }
    // sample extraction (tlwe_cpu.py:55-58)
    for (int j = tid; j < K * 1024; j += THREADS) {
        const int m = j >> 10, jj = j & 1023;
        const u32 v = (u32)lds.acc[j];
        out_a[m * 1024 + ((1024 - jj) & 1023)] = (i32)(jj == 0 ? v : 0u - v);
    }
    if (tid == 0) *out_b = lds.acc[K * 1024];
}

// ------------------------------------------------------------------------------------------
// Pair variant of the FFT path (k = 1, medium batches: CUs < bits <= 3 x CUs), the counterpart of blind_rotate.h
// (brp_*): two wavefronts per bit.  Wave W decomposes both digits of polynomial W, transforms them together
// (fft_forward_n<2>) and multiplies them with its half of the row: partial sums of both output polynomials.  The partial
// sum of the OTHER wave's output crosses through this wave's first exchange buffer (idle between the products and the
// inverse transform), pair barrier, add, pair barrier, then wave W transforms output W back, rounds and updates ACC[W],
// which only it reads and writes during the loop.  The two partial sums are added as (m = 0) + (m = 1) whichever wave
// does it (fp64 addition commutes), a different association from the one-wave kernel's running sum: covered by the
// path's stated tolerance, identical after rounding on every tested input.
// ------------------------------------------------------------------------------------------
struct BrFftPairLds {
    cplx *xbufA;             // this wave's two exchange buffers (FFT_XBUF_ELEMS complex each)
    cplx *xbufB;
    const cplx *xbufA_other; // the other wave's first buffer
    i32 *acc;                // [2][1024], shared by the pair
    uint16_t *bara;          // [BR_MAX_LWE], shared
    const cplx *tw1;
    const cplx *tw2;
    BrPace pace;
};

template <int W, class TW2, class PairSync>
FF_FN void brfp_step(u32 a, const cplx *row, const BrFftPairLds &lds, const TW2 &tw2, const FftLane &L, PairSync &&pair_sync)
{
    const int lane = L.lane;
    cplx *const bufs[2] = {lds.xbufA, lds.xbufB};
    cplx sum[8];
    {
        u32 T[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const u32 j = (u32)(lane + 64 * r);
            const u32 t = (j - a) & 2047u;
            const u32 v = (u32)lds.acc[W * 1024 + (t & 1023u)];
            T[r] = ((t & 1024u) ? 0u - v : v) - (u32)lds.acc[W * 1024 + j];     // (X^a - 1) ACC_W
        }
        cplx x[2][8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            x[0][r] = cplx{(double)br_digit<0>(T[r]), -(double)br_digit<0>(T[r + 8])};   // a_j - i a_{j+512}
            x[1][r] = cplx{(double)br_digit<1>(T[r]), -(double)br_digit<1>(T[r + 8])};
        }
        fft_forward_n<2>(x, bufs, lds.tw1, tw2, L);
        cplx ps[2][8];
#pragma unroll
        for (int r = 0; r < 8; r++) { ps[0][r] = cplx{0.0, 0.0}; ps[1][r] = cplx{0.0, 0.0}; }
        brf_mac_pair(ps, x, row, W, lane);
        WAVE_SYNC();     // every lane is done with the exchange buffers
#pragma unroll
        for (int r = 0; r < 8; r++) {
            lds.xbufA[r * 64 + lane] = ps[1 - W][r];
            sum[r] = ps[W][r];
        }
    }
    pair_sync();
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const cplx o = lds.xbufA_other[r * 64 + lane];
        sum[r].re += o.re;
        sum[r].im += o.im;
    }
    pair_sync();
    {
        cplx (&one)[1][8] = reinterpret_cast<cplx (&)[1][8]>(sum);
        cplx *const buf1[1] = {lds.xbufA};
        fft_inverse_n<1>(one, buf1, lds.tw1, tw2, L);
    }
#pragma unroll
    for (int r = 0; r < 8; r++) {
        lds.acc[W * 1024 + lane + 64 * r] += (i32)fft_round_to_u32(sum[r].re);
        lds.acc[W * 1024 + lane + 64 * (r + 8)] += (i32)fft_round_to_u32(-sum[r].im);
    }
    WAVE_SYNC();
}

template <int W, class PairSync>
FF_FN void brfp_bootstrap(i32 *out_a, i32 *out_b, const BrSource &s0, const BrSource &s1, i32 c0, long bit,
                          const cplx *bk, int n, i32 mu, const BrFftPairLds &lds, const FftLane &L, PairSync &&pair_sync)
{
    const int tid = 64 * W + L.lane;
    for (int i = tid; i < n; i += 128) {
        u32 v = 0;
        if (s0.p) v += (u32)s0.p * (u32)s0.a[bit * s0.a_stride + i];
        if (s1.p) v += (u32)s1.p * (u32)s1.a[bit * s1.a_stride + i];
        lds.bara[i] = (uint16_t)br_modswitch(v);
    }
    u32 vb = (u32)c0;
    if (s0.p) vb += (u32)s0.p * (u32)s0.b[bit * s0.b_stride];
    if (s1.p) vb += (u32)s1.p * (u32)s1.b[bit * s1.b_stride];
    const u32 barb = br_modswitch(vb);
    for (int j = tid; j < 1024; j += 128) {
        const u32 t = ((u32)j + barb) & 2047u;
        lds.acc[j] = 0;
        lds.acc[1024 + j] = (t < 1024u) ? mu : (i32)(0u - (u32)mu);
    }
    FftTw2Regs tw2;
    fft_tw2_load(tw2, lds.tw2, L);
    WAVE_SYNC();
    pair_sync();
    for (int i = 0; i < n; i++) {
        br_pace(lds.pace, (u32)i);
        const u32 a = WAVE_UNIFORM((u32)lds.bara[i]);
        if (a == 0) continue;
        brfp_step<W>(a, bk + (long)i * BKF_ROW_ELEMS, lds, tw2, L, pair_sync);
    }
    br_pace_done(lds.pace);
    pair_sync();
    for (int j = tid; j < 1024; j += 128) {
        const u32 v = (u32)lds.acc[j];
        out_a[(1024 - j) & 1023] = (i32)(j == 0 ? v : 0u - v);
    }
    if (tid == 0) *out_b = lds.acc[1024];
}

// ------------------------------------------------------------------------------------------
// Ring variant of the FFT path for tlwe_mask_size = K > 1 (the counterpart of blind_rotate.h, brr_*): K + 1 wavefronts
// per bit, wave w owns input polynomial w (its two interleaved digit transforms and their products with
// BK_row[w][0..1][0..K]) and output polynomial w.  No partial-sum buffer: in phase p = 1..K wave w writes its partial sum
// of output (w + p) mod (K + 1) into its first exchange buffer, team barrier, adds the one wave (w - p) mod (K + 1)
// wrote for it, team barrier.  67 KiB of LDS per team for K = 2, two teams per CU.  fp64 sums: own share first, then the
// received ones in phase order -- another association than the other kernels', covered by the path's tolerance and
// identical after rounding on every tested input.
// ------------------------------------------------------------------------------------------
struct BrFftRingLds {
    cplx *xbuf_team;         // K + 1 pairs of exchange buffers (2 x FFT_XBUF_ELEMS complex per wave); wave w: pair w
    i32 *acc;                // [K+1][1024], shared by the team
    uint16_t *bara;          // [BR_MAX_LWE], shared
    const cplx *tw1;
    const cplx *tw2;
    BrPace pace;
};

// out = x[0] (*) poly0 + x[1] (*) poly1 (one output polynomial; key loads in groups of 4, one group ahead)
FF_FN void brf_mac_one(cplx (&out)[8], const cplx (&x)[2][8], const cplx *poly0, const cplx *poly1, int lane)
{
    auto addr = [&](int g, int i) { return ((g >> 1) ? poly1 : poly0) + lane + (4 * (g & 1) + i) * 64; };   // g = 2 d + half
    cplx k[4], n[4];
#pragma unroll
    for (int r = 0; r < 8; r++) out[r] = cplx{0.0, 0.0};
#pragma unroll
    for (int i = 0; i < 4; i++) k[i] = *addr(0, i);
#pragma unroll
    for (int g = 0; g < 4; g++) {
        if (g + 1 < 4) {
#pragma unroll
            for (int i = 0; i < 4; i++) n[i] = *addr(g + 1, i);
            BR_ISSUE_FENCE();
        }
        const int d = g >> 1;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = 4 * (g & 1) + i;
            c_fma_acc(out[r], x[d][r], k[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) k[i] = n[i];
    }
}

template <int K, class TW2, class TeamSync>
FF_FN void brfr_step(u32 a, const cplx *row, const BrFftRingLds &lds, const TW2 &tw2, const FftLane &L, int w,
                     TeamSync &&team_sync)
{
    const int lane = L.lane;
    cplx *xbufA = lds.xbuf_team + (long)w * 2 * FFT_XBUF_ELEMS;
    cplx *const bufs[2] = {xbufA, xbufA + FFT_XBUF_ELEMS};
    cplx x[2][8];
    {
        u32 T[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const u32 j = (u32)(lane + 64 * r);
            const u32 t = (j - a) & 2047u;
            const u32 v = (u32)lds.acc[w * 1024 + (t & 1023u)];
            T[r] = ((t & 1024u) ? 0u - v : v) - (u32)lds.acc[w * 1024 + j];     // (X^a - 1) ACC_w
        }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            x[0][r] = cplx{(double)br_digit<0>(T[r]), -(double)br_digit<0>(T[r + 8])};   // a_j - i a_{j+512}
            x[1][r] = cplx{(double)br_digit<1>(T[r]), -(double)br_digit<1>(T[r + 8])};
        }
    }
    fft_forward_n<2>(x, bufs, lds.tw1, tw2, L);
    // polynomial (w, d, mo) of the row: ((w * 2 + d) * (K + 1) + mo)
    const cplx *rw = row + (long)w * 2 * (K + 1) * BKF_POLY_ELEMS;
    auto partial = [&](cplx (&out)[8], int mo) {
        brf_mac_one(out, x, rw + (long)mo * BKF_POLY_ELEMS, rw + (long)(K + 1 + mo) * BKF_POLY_ELEMS, lane);
    };
    cplx sum[8];
    WAVE_SYNC();                 // every lane is done with the exchange buffers
#pragma unroll 1
    for (int p = 1; p <= K; p++) {
        int to = w + p, from = w - p;
        if (to > K) to -= K + 1;
        if (from < 0) from += K + 1;
        {
            cplx ps[8];
            partial(ps, to);
#pragma unroll
            for (int r = 0; r < 8; r++) xbufA[r * 64 + lane] = ps[r];
        }
        if (p == 1) partial(sum, w);      // own share, while the others finish theirs
        team_sync();
        const cplx *src = lds.xbuf_team + (long)from * 2 * FFT_XBUF_ELEMS;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const cplx o = src[r * 64 + lane];
            sum[r].re += o.re;
            sum[r].im += o.im;
        }
        team_sync();
    }
    {
        cplx (&one)[1][8] = reinterpret_cast<cplx (&)[1][8]>(sum);
        cplx *const buf1[1] = {xbufA};
        fft_inverse_n<1>(one, buf1, lds.tw1, tw2, L);
    }
#pragma unroll
    for (int r = 0; r < 8; r++) {
        lds.acc[w * 1024 + lane + 64 * r] += (i32)fft_round_to_u32(sum[r].re);
        lds.acc[w * 1024 + lane + 64 * (r + 8)] += (i32)fft_round_to_u32(-sum[r].im);
    }
    WAVE_SYNC();
}

template <int K, class TeamSync>
FF_FN void brfr_bootstrap(i32 *out_a, i32 *out_b, const BrSource &s0, const BrSource &s1, i32 c0, long bit,
                          const cplx *bk, int n, i32 mu, const BrFftRingLds &lds, const FftLane &L, int w,
                          TeamSync &&team_sync)
{
    constexpr int THREADS = 64 * (K + 1);
    const int tid = 64 * w + L.lane;
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
    FftTw2Regs tw2;
    fft_tw2_load(tw2, lds.tw2, L);
    WAVE_SYNC();
    team_sync();
    for (int i = 0; i < n; i++) {
        br_pace(lds.pace, (u32)i);
        const u32 a = WAVE_UNIFORM((u32)lds.bara[i]);
        if (a == 0) continue;
        brfr_step<K>(a, bk + (long)i * BK_ROW_POLYS(K) * BKF_POLY_ELEMS, lds, tw2, L, w, team_sync);
    }
    br_pace_done(lds.pace);
    team_sync();
    for (int j = tid; j < K * 1024; j += THREADS) {
        const int m = j >> 10, jj = j & 1023;
        const u32 v = (u32)lds.acc[j];
        out_a[m * 1024 + ((1024 - jj) & 1023)] = (i32)(jj == 0 ? v : 0u - v);
    }
    if (tid == 0) *out_b = lds.acc[K * 1024];
}

// ------------------------------------------------------------------------------------------
// Quad variant of the FFT path (k = 1, batches up to 1 x CUs bits: the latency case), the counterpart of brxq_* in
// blind_rotate_xfft.h: FOUR wavefronts per bit, the work-group is the team (s_barrier).
//   forward side : wave w = 2 m + d transforms digit d of (X^a - 1) ACC_m and leaves it in its first exchange buffer
//   product side : wave w = 2 mo + p sums  X_q (.) BK_row[q >> 1][q & 1][mo]  over q = 2 p, 2 p + 1;  the waves p = 1 hand
//                  their partial sums over through their second buffer, the waves p = 0 add them, transform back, round and
//                  update ACC_mo (which only they write)
// fp64 sums as (q0 + q1) + (q2 + q3): another association than the one-wave kernel's running sum -- covered by the
// path's stated tolerance, identical after rounding on every tested input (as brfp_*, brfr_*).
// Key words: 16 per wave and step, requested in front of the rotation and behind the first exchange write of the forward
// transform.  Three barriers per step: X visible | partial sums visible | ACC complete.
// ------------------------------------------------------------------------------------------
struct BrFftQuadLds {
    cplx *xbuf;              // this wave's exchange buffer of the forward transform; then its transformed digit polynomial
    cplx *xbuf_inv;          // its second buffer: p = 1: the partial sum it hands over; p = 0: exchanges of the inverse transform
    const cplx *xbuf_team;   // the team's 2 (K + 1) first buffers, FFT_XBUF_ELEMS apart
    const cplx *partner_inv; // second buffer of wave w ^ 1
    i32 *acc;                // [K + 1][1024], shared by the team
    uint16_t *bara;          // [BR_MAX_LWE], shared
    const cplx *tw1;
    const cplx *tw2;
};

// tlwe_mask_size = K: 2 (K + 1) waves per bit; wave 2 mo + p sums the K + 1 transformed digit polynomials q = p (K + 1) ... of
// output mo (key rows [m][d][mo][reg][lane], q = 2 m + d)
template <int W, int K, class TeamSync>
FF_FN void brfq_step(u32 a, const cplx *row, const BrFftQuadLds &lds, const FftLane &L, TeamSync &&team_sync)
{
    constexpr int NH = K + 1;                     // digit polynomials per product-side wave
    constexpr int M = W >> 1, D = W & 1;          // forward side
    constexpr int MO = W >> 1, P = W & 1;         // product side
    constexpr int Q0 = P * NH;
    const int lane = L.lane;
    cplx key[NH][8];                              // key[i][r] = BK_row[q >> 1][q & 1][MO], q = Q0 + i
#if defined(__HIP_DEVICE_COMPILE__)
    typedef u32 brf_u32x4 __attribute__((ext_vector_type(4)));
    u32 voff = (u32)lane * (u32)sizeof(cplx);
    asm volatile("" : "+v"(voff));
    const u32 voff_hi = voff + 4096u;
    auto load_key = [&](int i) {
        // one descriptor per polynomial (8 KiB): every offset is an instruction immediate, no scalar offset operand
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(row + (long)((Q0 + i) * (K + 1) + MO) * BKF_POLY_ELEMS), (short)0, BKF_POLY_ELEMS * (int)sizeof(cplx), 0x00020000);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            union { brf_u32x4 w; cplx c; } u;
            u.w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (r < 4 ? voff : voff_hi) + (u32)((r & 3) * 1024), 0, 0);
            key[i][r] = u.c;
        }
    };
#else
    auto load_key = [&](int i) {
        const cplx *p = row + (long)((Q0 + i) * (K + 1) + MO) * BKF_POLY_ELEMS + lane;
#pragma unroll
        for (int r = 0; r < 8; r++) key[i][r] = p[r * 64];
    };
#endif
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
    fft_forward_n<1>(x, buf1, lds.tw1, lds.tw2, L, [&](int stage) {
        if (1 + stage < NH) {
            BR_ISSUE_FENCE();       // (fences on both sides: the requests stay where they are written)
            load_key(1 + stage);
            BR_ISSUE_FENCE();
        }
    });
    WAVE_SYNC();        // every lane is done with the exchange buffer
#pragma unroll
    for (int r = 0; r < 8; r++) lds.xbuf[r * 64 + lane] = x[0][r];
    team_sync();        // (1) the transformed digit polynomials are visible; every wave has read ACC
    cplx sum[1][8];
#pragma unroll
    for (int r = 0; r < 8; r++) sum[0][r] = cplx{0.0, 0.0};
#pragma unroll
    for (int i = 0; i < NH; i++) {
        cplx xq[8];
#pragma unroll
        for (int r = 0; r < 8; r++) xq[r] = (Q0 + i == W) ? x[0][r] : lds.xbuf_team[(Q0 + i) * FFT_XBUF_ELEMS + r * 64 + lane];
#pragma unroll
        for (int r = 0; r < 8; r++) c_fma_acc(sum[0][r], xq[r], key[i][r]);
#if defined(__HIP_DEVICE_COMPILE__)
        // the products are finished HERE (cf. brxq_step: left alone the scheduler sinks them below the barrier)
#pragma unroll
        for (int r = 0; r < 8; r++) asm volatile("" : "+v"(sum[0][r].re), "+v"(sum[0][r].im));
#endif
    }
    if (P == 1) {
#pragma unroll
        for (int r = 0; r < 8; r++) lds.xbuf_inv[r * 64 + lane] = sum[0][r];
        team_sync();    // (2) partial sums visible
        team_sync();    // (3) ACC complete
        return;
    }
    team_sync();        // (2)
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const cplx o = lds.partner_inv[r * 64 + lane];
        sum[0][r].re += o.re;
        sum[0][r].im += o.im;
    }
    WAVE_SYNC();
    cplx *const buf2[1] = {lds.xbuf_inv};
    fft_inverse_n<1>(sum, buf2, lds.tw1, lds.tw2, L);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        lds.acc[MO * 1024 + lane + 64 * r] += (i32)fft_round_to_u32(sum[0][r].re);
        lds.acc[MO * 1024 + lane + 64 * (r + 8)] += (i32)fft_round_to_u32(-sum[0][r].im);
    }
    team_sync();        // (3)
}

template <int W, int K, class TeamSync>
FF_FN void brfq_bootstrap(i32 *out_a, i32 *out_b, const BrSource &s0, const BrSource &s1, i32 c0, long bit,
                          const cplx *bk, int n, i32 mu, const BrFftQuadLds &lds, const FftLane &L, TeamSync &&team_sync)
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
    for (int i = 0; i < n; i++) {
        const u32 a = WAVE_UNIFORM((u32)lds.bara[i]);
        if (a == 0) continue;        // (all waves of the team read the same word)
        brfq_step<W, K>(a, bk + (long)i * BK_ROW_POLYS(K) * BKF_POLY_ELEMS, lds, L, team_sync);
    }
    // sample extraction (tlwe_cpu.py:55-58)
    for (int j = tid; j < K * 1024; j += THREADS) {
        const int m = j >> 10, jj = j & 1023;
        const u32 v = (u32)lds.acc[j];
        out_a[m * 1024 + ((1024 - jj) & 1023)] = (i32)(jj == 0 ? v : 0u - v);
    }
    if (tid == 0) *out_b = lds.acc[K * 1024];
}
