// blind_rotate.h -- fused bootstrap body: mod-switch + test-vector init + n x (rotate, gadget
// decomposition, (k+1) l forward NTTs, multiply-accumulate against one bootstrapping-key row,
// (k+1) inverse NTTs) + sample extraction, for ONE ciphertext bit executed by ONE wavefront.
//
// Replaces the reference's fused kernel (nufhe/blind_rotate.mako:18-226, launched from
// nufhe/blind_rotate.py:156-180: one 512+-thread work-group per bit with 3 barriers per transform)
// and the driver logic around it (nufhe/bootstrap.py:154-229).  Behaviour (bit-exact):
//   bara_i = round(a_i * 2N / 2^32), barb likewise          (numeric_functions_cpu.py:23-37)
//   ACC = (0, X^(2N - barb) * [mu, ..., mu])                (bootstrap.py:176-182)
//   for i < n: ACC += BK_i (.) ((X^bara_i - 1) * ACC)       (bootstrap.py:96-109, tgsw_cpu.py:82-106)
//   extract: a'_(mN) = A_m,0, a'_(mN+j) = -A_m,(N-j), b' = B_0   (tlwe_cpu.py:41-60)
//
// Templated on K = tlwe_mask_size (1 = the default parameters; 2 = the reference's second tested
// setting, test/test_gates.py:96-100, for which the reference's own fused kernel is unavailable,
// blind_rotate.py:53-58).  l = 2, Bg = 2^10, N = 1024 are fixed.
//
// CDNA4 mapping: a wave owns its bit completely -- the accumulator lives in 16 (K+1) VGPRs per lane
// (coefficient j = lane + 64 r) with an LDS mirror that is only used for the data-dependent
// rotation; the digit polynomials are transformed one after the other by the same wave and
// multiplied straight out of registers against the key row (stored in the wave layout, read with
// 16-byte loads, 1 KiB per wave instruction); no s_barrier anywhere in the loop.
#pragma once
#include <type_traits>
#include "ff.h"
#include "ntt1024.h"
#include "ntt1024_l4.h"
#include "ntt512_half.h"

#define BR_N 1024
#define BR_MAX_LWE 512                    /* capacity of the per-wave bara buffer (u16 each) */
#define BK_POLY_ELEMS 1024
#define BK_ROW_POLYS(K) (((K) + 1) * 2 * ((K) + 1))           /* (k+1) l (k+1) polynomials per TGSW row */
#define BK_ROW_ELEMS_K(K) (BK_ROW_POLYS(K) * BK_POLY_ELEMS)
#define BK_ROW_ELEMS BK_ROW_ELEMS_K(1)

#define TGSW_OFFSET 0x80200000u           /* 2^31 + 2^21: tgsw.py:49-52 with l=2, Bg=2^10 */

#if defined(__HIP_DEVICE_COMPILE__)
#define WAVE_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#define BR_ISSUE_FENCE() asm volatile("" ::: "memory")   /* keeps a prefetch load above the arithmetic */
#else
#define WAVE_UNIFORM(x) (x)
#define BR_ISSUE_FENCE() ((void)0)
#endif

// element offset of (lane, reg) inside one key polynomial in the wave layout:
// [reg/2][lane][reg%2] so that a wave reads 8 x (64 lanes x 16 bytes)
FF_FN constexpr int bk_elem_offset(int lane, int reg) { return (reg >> 1) * 128 + lane * 2 + (reg & 1); }

// One source term of the linear combination that feeds the bootstrap (gates.py:110-114):
// tmp = (0, c0) + sum_s p_s * src_s
struct BrSource {
    const i32 *a;      // [bit * a_stride + i]
    const i32 *b;      // [bit * b_stride]
    long a_stride;
    long b_stride;
    i32 p;             // 0 => term absent
};

// Pacing of the two wavefronts that share a SIMD.  A work-group of 8 one-bit waves puts waves w and w + 4 on
// the same SIMD, and the issue arbiter serves the OLDER wave first: measured (tools/probe_segments.py), wave w ran
// its 500 iterations in 15.2 ms at the expense of wave w + 4, which then needed another 8.1 ms ALONE -- and a single
// wave fills only ~60 % of the SIMD's issue slots, while the work-group holds the CU until its last wave ends.
// Each wave therefore publishes its iteration count in LDS and lowers its own priority (s_setprio) while it is
// ahead of its partner: both finish together and the SIMD has two runnable waves to the end.  Scheduling only --
// no effect on any result.
struct BrPace {
    u32 *mine;          // nullptr: no partner on this SIMD (fewer than 5 waves in the work-group, emulator)
    const u32 *other;
};

FF_FN void br_pace(const BrPace &p, u32 done)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (p.mine == nullptr) return;
    // the pacing words live in LDS: address-space-3 accesses (ds_write_b32 / ds_read_b32, one lgkmcnt wait).  Through
    // generic volatile pointers the compiler emitted flat_store / flat_load with system-scope cache flags and a
    // `s_waitcnt vmcnt(0)` after each -- two serialised round trips at the top of every iteration (round 4).
    typedef __attribute__((address_space(3))) volatile u32 lds_vu32;
    *(lds_vu32 *)p.mine = done;
    const u32 o = WAVE_UNIFORM(*(lds_vu32 *)p.other);
    if (done > o) __builtin_amdgcn_s_setprio(0);
    else __builtin_amdgcn_s_setprio(2);
#else
    (void)p; (void)done;
#endif
}

FF_FN void br_pace_done(const BrPace &p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (p.mine == nullptr) return;
    typedef __attribute__((address_space(3))) volatile u32 lds_vu32;
    *(lds_vu32 *)p.mine = 0xFFFFFFFFu;
    __builtin_amdgcn_s_setprio(0);
#else
    (void)p;
#endif
}

struct BrLds {
    u64 *xbuf;         // NTT exchange buffer, NTT_XBUF_ELEMS u64
    i32 *acc;          // [K+1][1024] accumulator mirror for rotated reads
    uint16_t *bara;    // [BR_MAX_LWE]
    const u64 *tw1x;   // [1024] forward table in the layout of ntt_make_tw1x (limb-form transform)
    const u64 *tw1i;   // [1024]
    BrPace pace;
};

// mod-switch, Torus32ToPhase with mspace 2N = 2048 (numeric_functions_cpu.py:23-37)
FF_FN u32 br_modswitch(u32 x) { return (x + (1u << 20)) >> 21; }

// Prologue: linear combination + mod-switch of the n mask coefficients into LDS, returns barb
FF_FN u32 br_prologue(const BrSource &s0, const BrSource &s1, i32 c0, long bit, int n, const BrLds &lds,
                      int lane)
{
    for (int i = lane; i < n; i += 64) {
        u32 v = 0;
        if (s0.p) v += (u32)s0.p * (u32)s0.a[bit * s0.a_stride + i];
        if (s1.p) v += (u32)s1.p * (u32)s1.a[bit * s1.a_stride + i];
        lds.bara[i] = (uint16_t)br_modswitch(v);
    }
    u32 vb = (u32)c0;
    if (s0.p) vb += (u32)s0.p * (u32)s0.b[bit * s0.b_stride];
    if (s1.p) vb += (u32)s1.p * (u32)s1.b[bit * s1.b_stride];
    return br_modswitch(vb);
}

// gadget decomposition digit p (1-based) of a torus coefficient: tgsw_cpu.py:41-47
template <int DIGIT>
FF_FN i32 br_digit(u32 t)
{
    // the top digit in two instructions: the 2^31 of the offset flips the top bit of the top field, which is what
    // "mask, then subtract 512" undoes -- (t + 2^21) >> 22 arithmetically is the same number
    if constexpr (DIGIT == 0) return (i32)(t + (TGSW_OFFSET - 0x80000000u)) >> 22;
    const i32 x = (i32)(t + TGSW_OFFSET);
    return ((x >> (32 - 10 * (DIGIT + 1))) & 1023) - 512;
}

// sum[mo] += x (*) poly[mo], mo = 0..K: multiply-accumulate of one transformed digit polynomial
// against the K+1 key polynomials BK[i][m][d][:] (tgsw_cpu.py:63-77).  The 16-byte key loads are
// software-pipelined one step ahead: the load of pair t+1 is in flight while pair t is multiplied
// (the key row comes out of L2, ~2 us of exposed latency per wave-iteration otherwise).
template <int K>
FF_FN void br_mac(u64 (&sum)[K + 1][16], const u64 (&x)[16], const u64 *poly, int lane)
{
    const u64 *p = poly + lane * 2;
    u64 k0 = p[0], k1 = p[1];
#pragma unroll
    for (int t = 0; t < 8 * (K + 1); t++) {
        const int mo = t >> 3, h = t & 7;
        u64 n0 = 0, n1 = 0;
        if (t + 1 < 8 * (K + 1)) {
            const u64 *q = p + ((t + 1) >> 3) * BK_POLY_ELEMS + ((t + 1) & 7) * 128;
            n0 = q[0];
            n1 = q[1];
            BR_ISSUE_FENCE();
        }
        sum[mo][2 * h] = ff_add(sum[mo][2 * h], ff_mul(x[2 * h], k0));
        sum[mo][2 * h + 1] = ff_add(sum[mo][2 * h + 1], ff_mul(x[2 * h + 1], k1));
        k0 = n0;
        k1 = n1;
    }
}

// Paired form of the multiply-accumulate: both digit polynomials x0, x1 of one input polynomial m at
// once, sum[mo] (+)= x0 (*) BK[m][0][mo] + x1 (*) BK[m][1][mo] with one reduction per pair of products
// (ff_dot2).  FIRST: the sums are assigned, not accumulated.  poly = BK_row[m][0][0]; the d = 1
// polynomials follow K + 1 polynomials later.  Key loads run one step ahead, as in br_mac.
// BR_KEY_DEPTH key-load steps are kept in flight ahead of the multiplication that consumes them
#ifndef BR_KEY_DEPTH
#define BR_KEY_DEPTH 1
#endif

// Variant-build instrumentation (tools/build_variant.sh probe -DBR_PROBE): shader-clock ticks spent by wave 0
// of work-group 0 in the segments of one external product, summed over the blind rotation.  Never part of
// the shipped library.
#if defined(BR_PROBE) && defined(__HIPCC__)
static __device__ unsigned long long g_br_probe[16];
static __device__ unsigned int g_br_probe_life[2 * 8192];   // per wave: start, end (wall_clock64, 10 ns units, low 32 bits)
#endif
#if defined(BR_PROBE) && defined(__HIP_DEVICE_COMPILE__)
struct BrProbe { long long t[8]; long long last; };
#define BR_PROBE_ARG , BrProbe &probe_
#define BR_PROBE_PASS , probe_
#define BR_PROBE_BEGIN()                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    probe_.last = clock64();                                                                          \
    __builtin_amdgcn_sched_barrier(0)
#define BR_PROBE_MARK(i)                                                                              \
    do {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        const long long now_ = clock64();                                                             \
        probe_.t[i] += now_ - probe_.last;                                                            \
        probe_.last = now_;                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    } while (0)
#else
#define BR_PROBE_ARG
#define BR_PROBE_PASS
#define BR_PROBE_BEGIN()
#define BR_PROBE_MARK(i)
#endif

// The key words of one lane: element pair (2 lane, 2 lane + 1) of the 128-element block at `elem` (a compile-time
// offset in u64 units from a wave-uniform polynomial pointer).  With BR_KEY_BUFFER_LOADS (the translation unit of the
// one-wave-per-bit kernels) these are buffer loads: the uniform part of the address sits in scalar registers and is added
// by the scalar unit, the lane contributes one 32-bit offset; a per-lane 64-bit pointer costs a v_add_co / v_addc pair
// for every 4 KiB the 12-bit immediate offset cannot reach.  `poly` MUST be wave-uniform there.
#if defined(__HIP_DEVICE_COMPILE__) && defined(BR_KEY_BUFFER_LOADS)
struct BrKeyStream {
    __amdgpu_buffer_rsrc_t rsrc;
    u32 voff;
};
FF_FN BrKeyStream br_key_stream(const u64 *poly, int lane, int bytes)
{
    BrKeyStream s;
    s.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)poly, (short)0, bytes, 0x00020000);
    s.voff = (u32)lane * 16u;
    return s;
}
FF_FN void br_key_load2(u64 &a, u64 &b, const BrKeyStream &s, int elem)
{
    typedef u32 br_u32x4 __attribute__((ext_vector_type(4)));
    const int byte = elem * 8;
    const br_u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(s.rsrc, s.voff + (u32)(byte & 4095), byte & ~4095, 0);
    a = ((u64)w[1] << 32) | w[0];
    b = ((u64)w[3] << 32) | w[2];
}
#else
struct BrKeyStream {
    const u64 *p;
};
FF_FN BrKeyStream br_key_stream(const u64 *poly, int lane, int) { return BrKeyStream{poly + lane * 2}; }
FF_FN void br_key_load2(u64 &a, u64 &b, const BrKeyStream &s, int elem)
{
    a = s.p[elem];
    b = s.p[elem + 1];
}
#endif

template <int K, bool FIRST>
FF_FN void br_mac2(u64 (&sum)[K + 1][16], const u64 (&x0)[16], const u64 (&x1)[16], const u64 *poly, int lane)
{
    constexpr int STEPS = 8 * (K + 1), D = BR_KEY_DEPTH, PD = (K + 1) * BK_POLY_ELEMS;
    const BrKeyStream ks = br_key_stream(poly, lane, 2 * PD * 8);
    auto off = [](int t) { return (t >> 3) * BK_POLY_ELEMS + (t & 7) * 128; };
    u64 q[D + 1][4];                      // ring of pending steps: q[i] belongs to step t + i
#pragma unroll
    for (int i = 0; i < D; i++) {
        br_key_load2(q[i][0], q[i][1], ks, off(i));
        br_key_load2(q[i][2], q[i][3], ks, PD + off(i));
    }
#pragma unroll
    for (int t = 0; t < STEPS; t++) {
        const int mo = t >> 3, h = t & 7;
        if (t + D < STEPS) {
            const int o = off(t + D);
            br_key_load2(q[D][0], q[D][1], ks, o);
            br_key_load2(q[D][2], q[D][3], ks, PD + o);
            BR_ISSUE_FENCE();
        }
        sum[mo][2 * h] = ff_dot2<!FIRST>(x0[2 * h], q[0][0], x1[2 * h], q[0][2], sum[mo][2 * h]);
        sum[mo][2 * h + 1] = ff_dot2<!FIRST>(x0[2 * h + 1], q[0][1], x1[2 * h + 1], q[0][3], sum[mo][2 * h + 1]);
#pragma unroll
        for (int i = 0; i < D; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) q[i][j] = q[i + 1][j];
    }
}

// Last half of the paired multiply-accumulate for ONE output polynomial: out = prev + x0 (*) BK[m][0][mo]
// + x1 (*) BK[m][1][mo], the 130-bit sums split straight into limbs (l4_dot2) -- the input format of
// the inverse transform, so the 128 -> 64-bit modular reduction is skipped.  poly = BK_row[m][0][mo].
template <int K>
FF_FN void br_mac2_l4(L4 (&out)[16], const u64 (&prev)[16], const u64 (&x0)[16], const u64 (&x1)[16],
                      const u64 *poly, int lane)
{
    constexpr int D = BR_KEY_DEPTH, PD = (K + 1) * BK_POLY_ELEMS;
    const BrKeyStream ks = br_key_stream(poly, lane, (PD + BK_POLY_ELEMS) * 8);
    u64 q[D + 1][4];
#pragma unroll
    for (int i = 0; i < D; i++) {
        br_key_load2(q[i][0], q[i][1], ks, i * 128);
        br_key_load2(q[i][2], q[i][3], ks, PD + i * 128);
    }
#pragma unroll
    for (int h = 0; h < 8; h++) {
        if (h + D < 8) {
            const int o = (h + D) * 128;
            br_key_load2(q[D][0], q[D][1], ks, o);
            br_key_load2(q[D][2], q[D][3], ks, PD + o);
            BR_ISSUE_FENCE();
        }
        l4_dot2<true>(out[2 * h], x0[2 * h], q[0][0], x1[2 * h], q[0][2], prev[2 * h]);
        l4_dot2<true>(out[2 * h + 1], x0[2 * h + 1], q[0][1], x1[2 * h + 1], q[0][3], prev[2 * h + 1]);
#pragma unroll
        for (int i = 0; i < D; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) q[i][j] = q[i + 1][j];
    }
}

// External product of one TGSW row with a TLWE sample given coefficient-wise by `tsrc`:
//   emit(mo, r, v) receives v = coefficient lane + 64 r (mod 2^32) of
//   sum_{m, d} digit_d(T_m) (*) BK_row[m][d][mo]   (tgsw_cpu.py:82-106), one output polynomial after
//   the other, as soon as its inverse transform is done.
// tsrc(m, T) must fill T[r] = coefficient lane + 64 r of polynomial m (as uint32 torus values); all
// sources are read before the first emit.
// The transforms run on redundant 24-bit limbs (ntt1024_l4.h); the digit transforms leave 64-bit
// representatives for the key products.
// K = 1: everything unrolled (tsrc may index registers with m); the products of the second input
// polynomial are added to the reduced sums of the first and split straight into limbs for the inverse
// transforms.  K > 1: the loop over m stays rolled -- two forward transforms + the paired
// multiply-accumulate of code -- so tsrc gets a run-time m and must read its polynomial from memory
// (the LDS mirror), never from a register array.
template <int K, class TSource, class Emit>
FF_FN void br_external_product(TSource &&tsrc, Emit &&emit, const u64 *row, const BrLds &lds, const NttLane &L BR_PROBE_ARG)
{
    const int lane = L.lane;
    auto digits_forward = [&](int m, u64(&x0)[16], u64(&x1)[16]) {
        u32 T[16];
        tsrc(m, T);
        i32 dg[16];
#pragma unroll
        for (int r = 0; r < 16; r++) dg[r] = br_digit<0>(T[r]);
        ntt_forward_small_l4(x0, dg, lds.xbuf, lds.tw1x, L);
#pragma unroll
        for (int r = 0; r < 16; r++) dg[r] = br_digit<1>(T[r]);
        ntt_forward_small_l4(x1, dg, lds.xbuf, lds.tw1x, L);
    };
    auto finish = [&](int mo, u32(&c)[16]) {
        // coefficients j2 >= 1 come back negated (ntt_inverse_l4_core)
#pragma unroll
        for (int r = 0; r < 16; r++) emit(mo, r, r == 0 ? c[r] : 0u - c[r]);
    };
    if constexpr (K == 1) {
        BR_PROBE_BEGIN();
        u64 sum[2][16];
        {
            u64 x0[16], x1[16];
            digits_forward(0, x0, x1);
            BR_PROBE_MARK(0);
            br_mac2<1, true>(sum, x0, x1, row, lane);
            BR_PROBE_MARK(1);
        }
        u64 x0[16], x1[16];
        digits_forward(1, x0, x1);
        BR_PROBE_MARK(2);
        {
            L4 s[16];
            u32 c[16];
            br_mac2_l4<1>(s, sum[0], x0, x1, row + 4 * BK_POLY_ELEMS, lane);
            BR_PROBE_MARK(3);
            ntt_inverse_l4_core(c, s, lds.xbuf, lds.tw1i, L);
            finish(0, c);
            BR_PROBE_MARK(4);
        }
        {
            L4 s[16];
            u32 c[16];
            br_mac2_l4<1>(s, sum[1], x0, x1, row + 5 * BK_POLY_ELEMS, lane);
            BR_PROBE_MARK(5);
            ntt_inverse_l4_core(c, s, lds.xbuf, lds.tw1i, L);
            finish(1, c);
            BR_PROBE_MARK(6);
        }
    } else {
        u64 sum[K + 1][16];
#pragma unroll
        for (int mo = 0; mo <= K; mo++)
#pragma unroll
            for (int r = 0; r < 16; r++) sum[mo][r] = 0;
#pragma unroll 1
        for (int m = 0; m <= K; m++) {
            u64 x0[16], x1[16];
            digits_forward(m, x0, x1);
            br_mac2<K, false>(sum, x0, x1, row + (long)m * 2 * (K + 1) * BK_POLY_ELEMS, lane);
        }
        // Explicitly sequenced (not a `#pragma unroll` loop: the three inverse transforms exceed the
        // unroller's size threshold and a rolled loop would put `sum` into scratch memory).
        u32 c[16];
        ntt_inverse_l4_i32(c, sum[0], lds.xbuf, lds.tw1i, L);
        finish(0, c);
        ntt_inverse_l4_i32(c, sum[1], lds.xbuf, lds.tw1i, L);
        finish(1, c);
        if constexpr (K >= 2) {
            ntt_inverse_l4_i32(c, sum[2], lds.xbuf, lds.tw1i, L);
            finish(2, c);
        }
        static_assert(K <= 2, "add the further inverse transforms");
    }
}

// ACC = (0, ..., 0, X^(2N - barb) * mu) (bootstrap.py:176-182): body coefficient j is +mu if
// (j + barb) mod 2N < N, else -mu.  Fills the registers and the LDS mirror.
template <int K>
FF_FN void br_init_acc(u32 (&acc)[K + 1][16], u32 barb, i32 mu, const BrLds &lds, int lane)
{
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const u32 j = (u32)(lane + 64 * r);
        const u32 t = (j + barb) & 2047u;
#pragma unroll
        for (int m = 0; m < K; m++) {
            acc[m][r] = 0;
            lds.acc[m * 1024 + j] = 0;
        }
        acc[K][r] = (t < 1024u) ? (u32)mu : 0u - (u32)mu;
        lds.acc[K * 1024 + j] = (i32)acc[K][r];
    }
    WAVE_SYNC();
}

// One blind-rotate step with rotation amount a in [0, 2N) (bootstrap.py:96-109):
// ACC += BK_row (.) ((X^a - 1) ACC).  The accumulator lives in the LDS mirror during the loop (the
// registers are needed by the transforms and the (K+1) x 16 field-element sums); br_blind_rotate loads
// it into registers at the end.
template <int K>
FF_FN void br_step(u32 a, const u64 *row, const BrLds &lds, const NttLane &L BR_PROBE_ARG)
{
    const int lane = L.lane;
    br_external_product<K>(
        [&](int m, u32 (&T)[16]) {
            // T = (X^a - 1) ACC_m  (polynomials_cpu.py:46-58 with minus_one)
            const u32 base = (u32)lane - a;                      // (mod 2^32; only bits 0..10 are used)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const u32 t = base + 64u * (u32)r;               // source coefficient in bits 0..9; bit 10: it comes back negated
                const u32 v = (u32)lds.acc[m * 1024 + (t & 1023u)];
                const u32 self = (u32)lds.acc[m * 1024 + lane + 64 * r];
                const u32 sm = (u32)((i32)(t << 21) >> 31);      // 0 or 0xFFFFFFFF from bit 10 (one v_bfe_i32)
                T[r] = ((v ^ sm) - sm) - self;                   // 6 instructions per coefficient instead of 9 (round 4)
            }
        },
        [&](int mo, int r, u32 v) { lds.acc[mo * 1024 + lane + 64 * r] += (i32)v; }, row, lds, L BR_PROBE_PASS);
    WAVE_SYNC();
    BR_PROBE_MARK(7);
}

// The whole blind rotation for one bit; bara comes from the per-wave LDS buffer.
template <int K>
FF_FN void br_blind_rotate(u32 (&acc)[K + 1][16], const u64 *bk, int n, u32 barb, i32 mu, const BrLds &lds,
                           const NttLane &L)
{
    br_init_acc<K>(acc, barb, mu, lds, L.lane);
#if defined(BR_PROBE) && defined(__HIP_DEVICE_COMPILE__)
    BrProbe probe_ = {};
    const long long probe_w0 = clock64(), probe_r0 = wall_clock64();
#endif
    for (int i = 0; i < n; i++) {
        br_pace(lds.pace, (u32)i);
        const u32 a = WAVE_UNIFORM((u32)lds.bara[i]);
        if (a == 0) continue;   // (X^0 - 1) ACC = 0: the external product adds nothing
        br_step<K>(a, bk + (long)i * BK_ROW_ELEMS_K(K), lds, L BR_PROBE_PASS);
    }
    br_pace_done(lds.pace);
#if defined(BR_PROBE) && defined(__HIP_DEVICE_COMPILE__)
    if (L.lane == 0) {
        for (int i = 0; i < 8; i++) atomicAdd(&g_br_probe[i], (unsigned long long)probe_.t[i]);
        atomicAdd(&g_br_probe[14], (unsigned long long)(clock64() - probe_w0));
        atomicAdd(&g_br_probe[13], 1ull);
        atomicAdd(&g_br_probe[12], (unsigned long long)(wall_clock64() - probe_r0));   // 100 MHz
        const unsigned gw = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & 8191u;
        g_br_probe_life[2 * gw] = (unsigned)probe_r0;
        g_br_probe_life[2 * gw + 1] = (unsigned)wall_clock64();
    }
#endif
#pragma unroll
    for (int m = 0; m <= K; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[m][r] = (u32)lds.acc[m * 1024 + L.lane + 64 * r];
}

// Sample extraction straight from registers (tlwe_cpu.py:55-58); out_a has K * 1024 entries
template <int K>
FF_FN void br_extract(i32 *out_a, i32 *out_b, const u32 (&acc)[K + 1][16], int lane)
{
#pragma unroll
    for (int m = 0; m < K; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int j = lane + 64 * r;
            out_a[m * 1024 + ((1024 - j) & 1023)] = (i32)(j == 0 ? acc[m][r] : 0u - acc[m][r]);
        }
    if (lane == 0) *out_b = (i32)acc[K][0];
}

// ------------------------------------------------------------------------------------------
// Team variant (k = 1): FOUR wavefronts of one work-group share one ciphertext bit.  For batches
// that cannot fill the chip with one wave per bit (circuits: the slices of an adder / comparator
// chain are a few hundred bits wide) the latency of a gate is the 500-step dependent chain of ONE
// wave; splitting the step cuts that chain:
//   wave w = 2 m + d   decomposes digit d of polynomial m of (X^a - 1) ACC, transforms it and
//                      multiplies it with BK_i[m][d][0..1]               (1 forward transform, 2 MACs)
//   all waves          leave their two partial sums in LDS, work-group barrier
//   wave mo = 0, 1     adds the four partial sums of output polynomial mo, transforms back and
//                      updates ACC[mo] in the shared LDS mirror          (1 inverse transform)
//   work-group barrier
// i.e. 1 forward + 1 inverse transform on the critical path instead of 4 + 2.  The four partial
// sums are canonical field elements and field addition is exact, so regrouping them does not change
// a single bit of the result with respect to the one-wave path.
// ------------------------------------------------------------------------------------------
#define BRT_WAVES 4
#define BRT_PART_ELEMS (BRT_WAVES * 2 * 1024)   /* u64: partial sums [wave][mo][reg][lane] */

struct BrTeamLds {
    u64 *xbuf;         // this wave's NTT exchange buffer
    i32 *acc;          // [2][1024] accumulator, shared by the team (the only copy during the loop)
    uint16_t *bara;    // [BR_MAX_LWE], shared
    u64 *part;         // [BRT_PART_ELEMS], shared
    const u64 *tw1x;
    const u64 *tw1i;
};

// One blind-rotate step by the team; `w` is this wave's index (wave-uniform), team_sync() a barrier
// over the 4 waves.  All 4 waves must call it with the same a != 0.
template <class TeamSync>
FF_FN void brt_step(u32 a, const u64 *row, const BrTeamLds &lds, const NttLane &L, int w, TeamSync &&team_sync)
{
    const int lane = L.lane;
    const int m = w >> 1;
    const int sh = (w & 1) ? 12 : 22;      // digit d = w & 1: shift 32 - 10 (d + 1)
    i32 dg[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const u32 j = (u32)(lane + 64 * r);
        const u32 t = (j - a) & 2047u;
        const u32 v = (u32)lds.acc[m * 1024 + (t & 1023u)];
        const u32 T = ((t & 1024u) ? 0u - v : v) - (u32)lds.acc[m * 1024 + j];
        dg[r] = (((i32)(T + TGSW_OFFSET) >> sh) & 1023) - 512;
    }
    // The wave is alone on its SIMD, so nothing hides the latency of its key loads: all 32 (its two key polynomials,
    // 64 VGPRs -- the team kernel has them to spare) are issued HERE, before the forward transform, and have long
    // arrived when the products need them.
    u64 kq[2][16];
    {
        const u64 *kp = row + (long)w * 2 * BK_POLY_ELEMS + lane * 2;
#pragma unroll
        for (int mo = 0; mo < 2; mo++)
#pragma unroll
            for (int h = 0; h < 8; h++) {
                kq[mo][2 * h] = kp[mo * BK_POLY_ELEMS + h * 128];
                kq[mo][2 * h + 1] = kp[mo * BK_POLY_ELEMS + h * 128 + 1];
            }
        BR_ISSUE_FENCE();
    }
    u64 x[16];
    ntt_forward_small_l4(x, dg, lds.xbuf, lds.tw1x, L);
#pragma unroll
    for (int mo = 0; mo < 2; mo++)
#pragma unroll
        for (int r = 0; r < 16; r++) lds.part[((w * 2 + mo) * 16 + r) * 64 + lane] = ff_mul(x[r], kq[mo][r]);
    team_sync();
    if (w < 2) {
        u64 sum[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            u64 acc = lds.part[((0 * 2 + w) * 16 + r) * 64 + lane];
#pragma unroll
            for (int src = 1; src < BRT_WAVES; src++)
                acc = ff_add(acc, lds.part[((src * 2 + w) * 16 + r) * 64 + lane]);
            sum[r] = acc;
        }
        u32 c[16];
        ntt_inverse_l4_i32(c, sum, lds.xbuf, lds.tw1i, L);
#pragma unroll
        for (int r = 0; r < 16; r++) lds.acc[w * 1024 + lane + 64 * r] += (i32)(r == 0 ? c[r] : 0u - c[r]);
    }
    team_sync();
}

// Whole bootstrap body of a k = 1 team for one bit, shared by the 4-wave and the 8-wave team: mod-switch of the mask
// into LDS, ACC = (0, X^(2N - barb) mu), the blind rotation through `step(a, row_index)` (all waves, same a != 0),
// extraction.  tid in [0, 64 WAVES); prologue and extraction are spread over all threads; out_a has 1024 entries.
template <int WAVES, class Step, class TeamSync>
FF_FN void br_team_bootstrap(i32 *out_a, i32 *out_b, const BrSource &s0, const BrSource &s1, i32 c0, long bit, int n,
                             i32 mu, i32 *acc, uint16_t *bara, int tid, Step &&step, TeamSync &&team_sync)
{
    for (int i = tid; i < n; i += 64 * WAVES) {
        u32 v = 0;
        if (s0.p) v += (u32)s0.p * (u32)s0.a[bit * s0.a_stride + i];
        if (s1.p) v += (u32)s1.p * (u32)s1.a[bit * s1.a_stride + i];
        bara[i] = (uint16_t)br_modswitch(v);
    }
    u32 vb = (u32)c0;
    if (s0.p) vb += (u32)s0.p * (u32)s0.b[bit * s0.b_stride];
    if (s1.p) vb += (u32)s1.p * (u32)s1.b[bit * s1.b_stride];
    const u32 barb = br_modswitch(vb);
    // ACC = (0, X^(2N - barb) * mu)  (bootstrap.py:176-182)
    for (int j = tid; j < 1024; j += 64 * WAVES) {
        const u32 t = ((u32)j + barb) & 2047u;
        acc[j] = 0;
        acc[1024 + j] = (t < 1024u) ? mu : (i32)(0u - (u32)mu);
    }
    team_sync();
    for (int i = 0; i < n; i++) {
        const u32 a = WAVE_UNIFORM((u32)bara[i]);
        if (a == 0) continue;
        step(a, i);
    }
    // sample extraction (tlwe_cpu.py:55-58)
    for (int j = tid; j < 1024; j += 64 * WAVES) {
        const u32 v = (u32)acc[j];
        out_a[(1024 - j) & 1023] = (i32)(j == 0 ? v : 0u - v);
    }
    if (tid == 0) *out_b = acc[1024];
}

template <class TeamSync>
FF_FN void brt_bootstrap(i32 *out_a, i32 *out_b, const BrSource &s0, const BrSource &s1, i32 c0, long bit,
                         const u64 *bk, int n, i32 mu, const BrTeamLds &lds, const NttLane &L, int w,
                         TeamSync &&team_sync)
{
    br_team_bootstrap<BRT_WAVES>(out_a, out_b, s0, s1, c0, bit, n, mu, lds.acc, lds.bara, 64 * w + L.lane,
                                 [&](u32 a, int i) { brt_step(a, bk + (long)i * BK_ROW_ELEMS, lds, L, w, team_sync); },
                                 team_sync);
}

// ------------------------------------------------------------------------------------------
// Half-ring team (k = 1, the smallest batches): EIGHT wavefronts share one bit -- two per digit transform.
// X^1024 + 1 = (X^512 - i)(X^512 + i), so the external product runs independently in two rings of length 512
// (ntt512_half.h) and only the gadget decomposition (before) and the accumulator update (after) see whole polynomials:
//   wave w = 4 m + 2 d + h   decomposes digit d of polynomial m of (X^a - 1) ACC (as in brt_step), folds it into half
//                            ring h, transforms it (8 values per lane) and multiplies it with its half of BK_i[m][d][0..1]
//   barrier; wave 2 mo + h (w < 4) adds the four partial sums of output polynomial mo in half ring h and transforms back
//   barrier; the two halves of an output meet: a_lo = Y^0 + Y^1 updates ACC[mo][0..511], a_hi = i (Y^1 - Y^0) updates
//            ACC[mo][512..1023] -- 4 x 512 coefficients, spread over all eight waves (256 each); barrier.
// Critical path: one HALF forward + one HALF inverse transform (measured 0.85 of the 4-wave team's: the forward phase
// is bound by the CU's issue slots, profiles/r03c_team8_phases.txt).  Every step is exact in
// GF(P), so the result equals the other kernels bit for bit.  The key is read in the half-ring layout
// [row][m][d][mo][h][reg 8][lane 64] (the same field elements in another order: nth_freq_index, k_bk_to_half).
// ------------------------------------------------------------------------------------------
#define BRH_WAVES 8
#define BRH_PART_ELEMS (BRH_WAVES * 2 * 512)     /* u64: partial sums [wave][mo][reg][lane] */
#define BRH_JOIN_ELEMS (4 * 512)                 /* u64: inverse outputs [mo][h][reg][lane] */
#define BKH_POLY_ELEMS 512                       /* one half of a key polynomial */

struct BrHalfLds {
    u64 *xbuf;         // this wave's exchange buffer (NTH_XBUF_ELEMS)
    i32 *acc;          // [2][1024], shared by the team
    uint16_t *bara;    // [BR_MAX_LWE], shared
    u64 *part;         // [BRH_PART_ELEMS], shared
    u64 *join;         // [BRH_JOIN_ELEMS], shared
    const u64 *tables; // NTH_TABLE_ELEMS (ntt512_half.h)
};

#if defined(BR_PROBE) && defined(__HIPCC__)
static __device__ unsigned long long g_brh_probe[8];
#endif
#if defined(BR_PROBE) && defined(__HIP_DEVICE_COMPILE__)
#define BRH_MARK(i)                                                                                   \
    do {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        const long long now_ = clock64();                                                             \
        if (w == 0 && lane == 0 && blockIdx.x == 0) g_brh_probe[i] += (unsigned long long)(now_ - brh_last_);   \
        brh_last_ = now_;                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    } while (0)
#define BRH_BEGIN() long long brh_last_ = clock64()
#else
#define BRH_MARK(i) do { } while (0)
#define BRH_BEGIN() do { } while (0)
#endif

template <class TeamSync>
FF_FN void brh_step(u32 a, const u64 *row, const BrHalfLds &lds, int lane, int w, TeamSync &&team_sync)
{
    BRH_BEGIN();
    const int m = w >> 2, h = w & 1;
    const int sh = (w & 2) ? 12 : 22;      // digit d = (w >> 1) & 1: shift 32 - 10 (d + 1)
    // this wave's half of BK_i[m][d][mo], mo = 0, 1: 16 loads issued before anything else (registers to spare)
    u64 kq[2][8];
    {
        const u64 *kp = row + ((long)(w >> 1) * 2 * 2 + h) * BKH_POLY_ELEMS + lane;     // [m][d][mo = 0][h]
#pragma unroll
        for (int mo = 0; mo < 2; mo++)
#pragma unroll
            for (int r = 0; r < 8; r++) kq[mo][r] = kp[mo * 2 * BKH_POLY_ELEMS + r * 64];
        BR_ISSUE_FENCE();
    }
    i32 dg[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const u32 j = (u32)(lane + 64 * r);
        const u32 t = (j - a) & 2047u;
        const u32 v = (u32)lds.acc[m * 1024 + (t & 1023u)];
        const u32 T = ((t & 1024u) ? 0u - v : v) - (u32)lds.acc[m * 1024 + j];
        dg[r] = (((i32)(T + TGSW_OFFSET) >> sh) & 1023) - 512;
    }
    BRH_MARK(0);
    u64 x[8];
    const NthTables tb = nth_tables(lds.tables, h);
    if (h == 0) nth_forward_small<0>(x, dg, lds.xbuf, tb, lane);
    else nth_forward_small<1>(x, dg, lds.xbuf, tb, lane);
    BRH_MARK(1);
#pragma unroll
    for (int mo = 0; mo < 2; mo++)
#pragma unroll
        for (int r = 0; r < 8; r++) lds.part[((w * 2 + mo) * 8 + r) * 64 + lane] = ff_mul(x[r], kq[mo][r]);   // any 64-bit representative of x
    BRH_MARK(2);
    team_sync();
    BRH_MARK(3);
    if (w < 4) {
        const int mo = w >> 1;                                 // reducer (mo, h)
        u64 sum[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            u64 acc = lds.part[(((0 * 2 + h) * 2 + mo) * 8 + r) * 64 + lane];
#pragma unroll
            for (int src = 1; src < 4; src++) acc = ff_add(acc, lds.part[(((src * 2 + h) * 2 + mo) * 8 + r) * 64 + lane]);
            sum[r] = acc;
        }
        BRH_MARK(4);
        u64 y[8];
        if (h == 0) nth_inverse<0>(y, sum, lds.xbuf, tb, lane);
        else nth_inverse<1>(y, sum, lds.xbuf, tb, lane);
#pragma unroll
        for (int r = 0; r < 8; r++) lds.join[((mo * 2 + h) * 8 + r) * 64 + lane] = ff_canon(y[r]);
        BRH_MARK(5);
    }
    team_sync();
    BRH_MARK(6);
    {
        // wave w joins registers 4 (w >> 2) .. + 3 of output polynomial (w >> 1) & 1, coefficient half h
        const int mo = (w >> 1) & 1;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int r = 4 * (w >> 2) + q;
            const u64 y0 = lds.join[((mo * 2 + 0) * 8 + r) * 64 + lane], y1 = lds.join[((mo * 2 + 1) * 8 + r) * 64 + lane];
            // a_j' = Y^0 + Y^1 (half 0 writes it), a_(j'+512) = i (Y^1 - Y^0) (half 1); ntt.mako:402-408 for the conversion
            const u64 v = h == 0 ? ff_add(y0, y1) : ff_mul_pow2<48>(ff_sub(y1, y0));
            lds.acc[mo * 1024 + h * 512 + lane + 64 * r] += ff_to_i32(v);
        }
    }
    team_sync();
    BRH_MARK(7);
}

// Whole bootstrap body of the half-ring team for one bit: tid = 64 w + lane in [0, 512)
template <class TeamSync>
FF_FN void brh_bootstrap(i32 *out_a, i32 *out_b, const BrSource &s0, const BrSource &s1, i32 c0, long bit,
                         const u64 *bkh, int n, i32 mu, const BrHalfLds &lds, int lane, int w, TeamSync &&team_sync)
{
    br_team_bootstrap<BRH_WAVES>(out_a, out_b, s0, s1, c0, bit, n, mu, lds.acc, lds.bara, 64 * w + lane,
                                 [&](u32 a, int i) { brh_step(a, bkh + (long)i * BK_ROW_ELEMS, lds, lane, w, team_sync); },
                                 team_sync);
}

// ------------------------------------------------------------------------------------------
// Team variant for tlwe_mask_size = K > 1 (small batches): K + 1 wavefronts share one bit, wave m owns
// input polynomial m -- both of its digit transforms and the paired multiply-accumulate against
// BK_row[m][0..1][0..K] (exactly the per-m body of br_external_product) -- and leaves K + 1 partial
// sums in LDS; after the work-group barrier wave mo adds the K + 1 partial sums of OUTPUT polynomial
// mo, transforms back and updates the shared accumulator.  Critical path: 2 forward + 1 inverse
// transform instead of 2 (K+1) + (K+1).  The partial sums are canonical field elements and field
// addition is exact, so the result equals the one-wave kernel bit for bit.
// ------------------------------------------------------------------------------------------
#define BRTK_PART_ELEMS(K) (((K) + 1) * ((K) + 1) * 1024)   /* u64: partial sums [wave m][mo][reg][lane] */

template <int K, class TeamSync>
FF_FN void brtk_step(u32 a, const u64 *row, const BrTeamLds &lds, const NttLane &L, int w, TeamSync &&team_sync)
{
    const int lane = L.lane;
    u32 T[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const u32 j = (u32)(lane + 64 * r);
        const u32 t = (j - a) & 2047u;
        const u32 v = (u32)lds.acc[w * 1024 + (t & 1023u)];
        T[r] = ((t & 1024u) ? 0u - v : v) - (u32)lds.acc[w * 1024 + j];     // (X^a - 1) ACC_w
    }
    u64 x0[16], x1[16];
    i32 dg[16];
#pragma unroll
    for (int r = 0; r < 16; r++) dg[r] = br_digit<0>(T[r]);
    ntt_forward_small_l4(x0, dg, lds.xbuf, lds.tw1x, L);
#pragma unroll
    for (int r = 0; r < 16; r++) dg[r] = br_digit<1>(T[r]);
    ntt_forward_small_l4(x1, dg, lds.xbuf, lds.tw1x, L);
    u64 ps[K + 1][16];
    br_mac2<K, true>(ps, x0, x1, row + (long)w * 2 * (K + 1) * BK_POLY_ELEMS, lane);
#pragma unroll
    for (int mo = 0; mo <= K; mo++)
#pragma unroll
        for (int r = 0; r < 16; r++) lds.part[((w * (K + 1) + mo) * 16 + r) * 64 + lane] = ps[mo][r];
    team_sync();
    u64 sum[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        u64 acc = lds.part[((0 * (K + 1) + w) * 16 + r) * 64 + lane];
#pragma unroll
        for (int src = 1; src <= K; src++) acc = ff_add(acc, lds.part[((src * (K + 1) + w) * 16 + r) * 64 + lane]);
        sum[r] = acc;
    }
    u32 c[16];
    ntt_inverse_l4_i32(c, sum, lds.xbuf, lds.tw1i, L);
#pragma unroll
    for (int r = 0; r < 16; r++) lds.acc[w * 1024 + lane + 64 * r] += (i32)(r == 0 ? c[r] : 0u - c[r]);
    team_sync();
}

// Whole bootstrap body of the (K+1)-wave team for one bit; out_a has K * 1024 entries.
template <int K, class TeamSync>
FF_FN void brtk_bootstrap(i32 *out_a, i32 *out_b, const BrSource &s0, const BrSource &s1, i32 c0, long bit,
                          const u64 *bk, int n, i32 mu, const BrTeamLds &lds, const NttLane &L, int w,
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
        brtk_step<K>(a, bk + (long)i * BK_ROW_ELEMS_K(K), lds, L, w, team_sync);
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
// Pair variant (k = 1): TWO wavefronts per ciphertext bit, for batches between the team kernel and a full chip
// (2 x CUs < bits <= 4 x CUs).  With one wave per bit such a batch leaves every SIMD with a single wave, which
// fills ~60 % of its issue slots; two waves per bit halve the dependent chain of a bit and put two waves (of
// different bits) on each SIMD once the batch exceeds 2 x CUs:
//   wave W (0 or 1)  decomposes BOTH digits of polynomial W of (X^a - 1) ACC, transforms them and multiplies them
//                    with BK_i[W][0..1][0..1]: its partial sums of both output polynomials  (2 forward transforms)
//   exchange         the partial sum of the OTHER wave's output goes through this wave's exchange buffer (idle
//                    between the products and the inverse transform), pair barrier, the other wave adds it to its
//                    own partial sum, pair barrier (the buffer is free again)
//   wave W           transforms output polynomial W back and updates ACC[W]         (1 inverse transform)
// Wave W is the only reader and writer of ACC[W] during the loop.  The partial sums are canonical field elements
// and field addition is exact, so the result is bit-identical to the one-wave path.  pair_sync() is a barrier over
// the two waves (kernels.hip: a counter handshake in LDS -- a work-group holds several pairs, which skip a == 0
// steps independently, so the work-group barrier cannot be used).
// ------------------------------------------------------------------------------------------
struct BrPairLds {
    u64 *xbuf;               // this wave's exchange buffer (NTT_XBUF_ELEMS u64)
    const u64 *xbuf_other;   // the other wave's
    i32 *acc;                // [2][1024], shared by the pair
    uint16_t *bara;          // [BR_MAX_LWE], shared
    const u64 *tw1x;
    const u64 *tw1i;
    BrPace pace;             // against the wave of ANOTHER pair on the same SIMD
};

template <int W, class PairSync>
FF_FN void brp_step(u32 a, const u64 *row, const BrPairLds &lds, const NttLane &L, PairSync &&pair_sync)
{
    const int lane = L.lane;
    u32 T[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const u32 j = (u32)(lane + 64 * r);
        const u32 t = (j - a) & 2047u;
        const u32 v = (u32)lds.acc[W * 1024 + (t & 1023u)];
        T[r] = ((t & 1024u) ? 0u - v : v) - (u32)lds.acc[W * 1024 + j];     // (X^a - 1) ACC_W
    }
    u64 sum[16];
    {
        u64 x0[16], x1[16];
        i32 dg[16];
#pragma unroll
        for (int r = 0; r < 16; r++) dg[r] = br_digit<0>(T[r]);
        ntt_forward_small_l4(x0, dg, lds.xbuf, lds.tw1x, L);
#pragma unroll
        for (int r = 0; r < 16; r++) dg[r] = br_digit<1>(T[r]);
        ntt_forward_small_l4(x1, dg, lds.xbuf, lds.tw1x, L);
        u64 ps[2][16];
        br_mac2<1, true>(ps, x0, x1, row + (long)W * 4 * BK_POLY_ELEMS, lane);
        WAVE_SYNC();     // every lane is done with the exchange buffer
#pragma unroll
        for (int r = 0; r < 16; r++) {
            lds.xbuf[r * 64 + lane] = ps[1 - W][r];
            sum[r] = ps[W][r];
        }
    }
    pair_sync();
#pragma unroll
    for (int r = 0; r < 16; r++) sum[r] = ff_add(sum[r], lds.xbuf_other[r * 64 + lane]);
    pair_sync();
    u32 c[16];
    ntt_inverse_l4_i32(c, sum, lds.xbuf, lds.tw1i, L);
#pragma unroll
    for (int r = 0; r < 16; r++) lds.acc[W * 1024 + lane + 64 * r] += (i32)(r == 0 ? c[r] : 0u - c[r]);
    WAVE_SYNC();
}

// Whole bootstrap body of wave W of a pair for one bit
template <int W, class PairSync>
FF_FN void brp_bootstrap(i32 *out_a, i32 *out_b, const BrSource &s0, const BrSource &s1, i32 c0, long bit,
                         const u64 *bk, int n, i32 mu, const BrPairLds &lds, const NttLane &L, PairSync &&pair_sync)
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
    // ACC = (0, X^(2N - barb) * mu)  (bootstrap.py:176-182)
    for (int j = tid; j < 1024; j += 128) {
        const u32 t = ((u32)j + barb) & 2047u;
        lds.acc[j] = 0;
        lds.acc[1024 + j] = (t < 1024u) ? mu : (i32)(0u - (u32)mu);
    }
    WAVE_SYNC();
    pair_sync();
    for (int i = 0; i < n; i++) {
        br_pace(lds.pace, (u32)i);
        const u32 a = WAVE_UNIFORM((u32)lds.bara[i]);
        if (a == 0) continue;
        brp_step<W>(a, bk + (long)i * BK_ROW_ELEMS, lds, L, pair_sync);
    }
    br_pace_done(lds.pace);
    pair_sync();
    // sample extraction (tlwe_cpu.py:55-58)
    for (int j = tid; j < 1024; j += 128) {
        const u32 v = (u32)lds.acc[j];
        out_a[(1024 - j) & 1023] = (i32)(j == 0 ? v : 0u - v);
    }
    if (tid == 0) *out_b = lds.acc[1024];
}

// ------------------------------------------------------------------------------------------
// Team variant WITHOUT a partial-sum buffer, any K (used for K = 2): K + 1 wavefronts per bit, wave w owns input
// polynomial w (two forward transforms, its products with BK_row[w][0..1][0..K]) and output polynomial w (one inverse
// transform, ACC[w], which only it reads and writes during the loop).  The partial sum of every OTHER wave's output is
// handed over through this wave's exchange buffer, one output per phase: in phase p = 1..K wave w writes its partial
// sum of output (w + p) mod (K + 1), team barrier, reads the one meant for it from wave (w - p) mod (K + 1), team
// barrier.  Only one partial sum is in registers at a time (brtk_* holds all K + 1 and needs a 72 KiB buffer for K = 2),
// so the team needs 39 KiB of LDS and two teams share a CU.  K = 1 is k_bootstrap_pair's scheme (brp_*).  Canonical
// partial sums, exact field additions: bit-identical to the other kernels.  team_sync() is a barrier over the K + 1
// waves of the team (LDS arrival counters, see kernels.hip).
// ------------------------------------------------------------------------------------------
struct BrRingLds {
    u64 *xbuf_team;          // K + 1 exchange buffers, NTT_XBUF_ELEMS u64 each; wave w uses number w
    i32 *acc;                // [K+1][1024], shared by the team
    uint16_t *bara;          // [BR_MAX_LWE], shared
    const u64 *tw1x;
    const u64 *tw1i;
    BrPace pace;
};

// out = x0 (*) poly0 + x1 (*) poly1 (one output polynomial; key loads one step ahead)
FF_FN void br_mac2_one(u64 (&out)[16], const u64 (&x0)[16], const u64 (&x1)[16], const u64 *poly0, const u64 *poly1, int lane)
{
    const u64 *p = poly0 + lane * 2, *pd = poly1 + lane * 2;
    u64 q0 = p[0], q1 = p[1], q2 = pd[0], q3 = pd[1];
#pragma unroll
    for (int h = 0; h < 8; h++) {
        u64 n0 = 0, n1 = 0, n2 = 0, n3 = 0;
        if (h + 1 < 8) {
            n0 = p[(h + 1) * 128]; n1 = p[(h + 1) * 128 + 1]; n2 = pd[(h + 1) * 128]; n3 = pd[(h + 1) * 128 + 1];
            BR_ISSUE_FENCE();
        }
        out[2 * h] = ff_dot2<false>(x0[2 * h], q0, x1[2 * h], q2, 0);
        out[2 * h + 1] = ff_dot2<false>(x0[2 * h + 1], q1, x1[2 * h + 1], q3, 0);
        q0 = n0; q1 = n1; q2 = n2; q3 = n3;
    }
}

template <int K, class TeamSync>
FF_FN void brr_step(u32 a, const u64 *row, const BrRingLds &lds, const NttLane &L, int w, TeamSync &&team_sync)
{
    const int lane = L.lane;
    u64 *xbuf = lds.xbuf_team + w * NTT_XBUF_ELEMS;
    u64 x0[16], x1[16];
    {
        u32 T[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const u32 j = (u32)(lane + 64 * r);
            const u32 t = (j - a) & 2047u;
            const u32 v = (u32)lds.acc[w * 1024 + (t & 1023u)];
            T[r] = ((t & 1024u) ? 0u - v : v) - (u32)lds.acc[w * 1024 + j];     // (X^a - 1) ACC_w
        }
        i32 dg[16];
#pragma unroll
        for (int r = 0; r < 16; r++) dg[r] = br_digit<0>(T[r]);
        ntt_forward_small_l4(x0, dg, xbuf, lds.tw1x, L);
#pragma unroll
        for (int r = 0; r < 16; r++) dg[r] = br_digit<1>(T[r]);
        ntt_forward_small_l4(x1, dg, xbuf, lds.tw1x, L);
    }
    // polynomial (w, d, mo) of the row: ((w * 2 + d) * (K + 1) + mo)
    const u64 *rw = row + (long)w * 2 * (K + 1) * BK_POLY_ELEMS;
    auto partial = [&](u64 (&out)[16], int mo) {
        br_mac2_one(out, x0, x1, rw + (long)mo * BK_POLY_ELEMS, rw + (long)(K + 1 + mo) * BK_POLY_ELEMS, lane);
    };
    u64 sum[16];
    WAVE_SYNC();                 // every lane is done with the exchange buffer
#pragma unroll 1
    for (int p = 1; p <= K; p++) {
        int to = w + p, from = w - p;
        if (to > K) to -= K + 1;
        if (from < 0) from += K + 1;
        {
            u64 ps[16];
            partial(ps, to);
#pragma unroll
            for (int r = 0; r < 16; r++) xbuf[r * 64 + lane] = ps[r];
        }
        if (p == 1) partial(sum, w);      // own share, while the others finish theirs
        team_sync();
        const u64 *src = lds.xbuf_team + from * NTT_XBUF_ELEMS;
#pragma unroll
        for (int r = 0; r < 16; r++) sum[r] = ff_add(sum[r], src[r * 64 + lane]);
        team_sync();
    }
    u32 c[16];
    ntt_inverse_l4_i32(c, sum, xbuf, lds.tw1i, L);
#pragma unroll
    for (int r = 0; r < 16; r++) lds.acc[w * 1024 + lane + 64 * r] += (i32)(r == 0 ? c[r] : 0u - c[r]);
    WAVE_SYNC();
}

// Whole bootstrap body of wave w of a (K+1)-wave ring team for one bit; out_a has K * 1024 entries.
template <int K, class TeamSync>
FF_FN void brr_bootstrap(i32 *out_a, i32 *out_b, const BrSource &s0, const BrSource &s1, i32 c0, long bit,
                         const u64 *bk, int n, i32 mu, const BrRingLds &lds, const NttLane &L, int w, TeamSync &&team_sync)
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
    WAVE_SYNC();
    team_sync();
    for (int i = 0; i < n; i++) {
        br_pace(lds.pace, (u32)i);
        const u32 a = WAVE_UNIFORM((u32)lds.bara[i]);
        if (a == 0) continue;
        brr_step<K>(a, bk + (long)i * BK_ROW_ELEMS_K(K), lds, L, w, team_sync);
    }
    br_pace_done(lds.pace);
    team_sync();
    // sample extraction (tlwe_cpu.py:55-58)
    for (int j = tid; j < K * 1024; j += THREADS) {
        const int m = j >> 10, jj = j & 1023;
        const u32 v = (u32)lds.acc[j];
        out_a[m * 1024 + ((1024 - jj) & 1023)] = (i32)(jj == 0 ? v : 0u - v);
    }
    if (tid == 0) *out_b = lds.acc[K * 1024];
}

