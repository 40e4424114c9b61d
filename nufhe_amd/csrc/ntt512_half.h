// ntt512_half.h -- the negacyclic NTT-1024 of ntt1024.h split into its two HALF RINGS, for the small-batch kernel that
// puts two wavefronts on every transform (k_bootstrap_team8, blind_rotate.h brh_*).
//
// X^1024 + 1 = (X^512 - i)(X^512 + i) over GF(P), i = psi^512 = 2^48, so Z_P[X]/(X^1024 + 1) is the product of two rings
// of length 512 and the external product of the blind rotation runs in each of them independently:
//
//   split      a(X) = a_lo + X^512 a_hi  ->  y^h = a_lo + s_h i a_hi,   s_0 = +1, s_1 = -1            (in lane: coefficient
//                                                                         j' and j' + 512 sit in the same lane)
//   forward    A_(2 kappa + h) = sum_{j' < 512} y^h_j' psi^((4 kappa + c) j'),  c = 2 h + 1           -- the SAME values as
//              the even (h = 0) / odd (h = 1) outputs of the 1024-point transform, so the bootstrapping key is the same
//              set of field elements in another order (nth_freq_index)
//   inverse    Y^h_j' = (1/1024) sum_kappa S_(2 kappa + h) psi^(-(4 kappa + c) j')
//   join       a_j' = Y^0 + Y^1,   a_(j' + 512) = i (Y^1 - Y^0)
//
// A half transform is a 512-point transform on ONE wavefront with 8 values per lane, factored 8 x 8 x 8:
//   j' = j1 + 64 j2' (lane j1, register j2'), kappa = kappa2 + 8 (a + 8 b)
//   pass 1 (in lane, j2' -> kappa2): pre-twist 2^(6 c j2') -- on gadget digits a placement of the digit in one limb --, then an
//           8-point cyclic transform whose root is 2^24: EVERY butterfly twiddle is a limb rotation (ff24.h)
//   layer 1: psi^((4 kappa2 + c) j1) (table, general multiplication) on the way through exchange 1:
//           (lane j1, register kappa2) -> (lane (kappa2, q), register r), j1 = q + 8 r
//   pass 2 (r -> a): 8-point, root 2^24
//   layer 2: 2^(3 a q) on the way through exchange 2: (lane (kappa2, q), register a) -> (lane (kappa2, a), register q).
//           Powers of two, but with BOTH indices three bits wide the per-lane shift-and-rotate of ntt1024_l4.h costs more
//           than the general multiplication that the exchange of packed 64-bit words gives almost for free
//           (pack -> 8-byte exchange -> 64 x 64 product -> split: the same sequence as layer 1)
//   pass 3 (q -> b): 8-point, root 2^24
// The arithmetic between the layers is the redundant 24-bit-limb form of ff24.h; limb bounds are noted at each step
// (|w| <= 2^30 wherever l4_to_u64 packs).
//
// Reference semantics replaced: nufhe/transform/ntt.mako:42-494 (the same transform, other factorisation); results are
// bit-identical because every step is exact modulo P.  Compiles for the host (tests/emu).
#pragma once
#include "ff24.h"
#include "ntt1024.h"
#include "ntt_tables.h"

#define NTH_ROW 68                      /* u64 per exchange row: 64 + 4 padding */
#define NTH_XBUF_ELEMS (8 * NTH_ROW)    /* 544 u64 = 4352 bytes per wave */
#define NTH_TW1_ELEMS 512               /* per half and direction */
#define NTH_TW2_ELEMS 64
// table block of one direction pair: fwd1[2][512] | inv1[2][512] | fwd2[64] | inv2[64]
#define NTH_TABLE_ELEMS (4 * NTH_TW1_ELEMS + 2 * NTH_TW2_ELEMS)

FF_FN constexpr int nth_br3(int i) { return ((i & 1) << 2) | (i & 2) | ((i & 4) >> 2); }

// natural-order frequency index (of the 1024-point transform) held by (lane, reg) of half h after nth_forward
FF_FN constexpr int nth_freq_index(int h, int lane, int reg)
{
    // lane = 8 kappa2 + a, reg = b, kappa = kappa2 + 8 (a + 8 b), k = 2 kappa + h
    return 2 * ((lane >> 3) + 8 * ((lane & 7) + 8 * reg)) + h;
}

struct NthTables {
    const u64 *fwd1;    // [r * 64 + lane']  psi^((4 kappa2 + c)(q + 8 r)),         lane' = 8 kappa2 + q   (this half)
    const u64 *inv1;    // [kappa2 * 64 + j1] psi^(-(4 kappa2 + c) j1) / 1024                              (this half)
    const u64 *fwd2;    // [q * 8 + a]        2^(3 a q)
    const u64 *inv2;    // [a * 8 + q]        2^(-3 a q)
};

FF_FN NthTables nth_tables(const u64 *block, int h)
{
    NthTables t;
    t.fwd1 = block + h * NTH_TW1_ELEMS;
    t.inv1 = block + 2 * NTH_TW1_ELEMS + h * NTH_TW1_ELEMS;
    t.fwd2 = block + 4 * NTH_TW1_ELEMS;
    t.inv2 = block + 4 * NTH_TW1_ELEMS + NTH_TW2_ELEMS;
    return t;
}

// 8-point cyclic transform with root 2^(24 SGN) (SGN = +1 forward, -1 inverse), decimation in frequency:
// natural-order input, bit-reversed output (x[i] = X[nth_br3(i)]); every twiddle is a limb rotation
template <int SGN>
FF_FN void l4_ntt8_dif(L4 (&x)[8])
{
    constexpr int W = 24 * SGN;
#define BFLY(i, j, S) l4_bfly<(S)>(x[i], x[j])
    BFLY(0, 4, 0);  BFLY(1, 5, W);  BFLY(2, 6, 2 * W);  BFLY(3, 7, 3 * W);
    BFLY(0, 2, 0);  BFLY(1, 3, 2 * W);  BFLY(4, 6, 0);  BFLY(5, 7, 2 * W);
    BFLY(0, 1, 0);  BFLY(2, 3, 0);  BFLY(4, 5, 0);  BFLY(6, 7, 0);
#undef BFLY
}

// one "layer": the elements cross an exchange as packed 64-bit words and the RECEIVING lane multiplies them with its
// table entries and splits the 128-bit products straight into limbs.  wr(i) / rd(i): element offsets in the buffer.
template <class Wr, class Rd, class Tw>
FF_FN void nth_layer(L4 (&x)[8], u64 *xbuf, Wr &&wr, Rd &&rd, Tw &&tw)
{
#pragma unroll
    for (int i = 0; i < 8; i++) xbuf[wr(i)] = l4_to_u64(x[i]);
    WAVE_SYNC();
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 lo, hi;
        ff_mul_wide(xbuf[rd(i)], tw(i), lo, hi);
        l4_from_u128(x[i], lo, hi);
    }
    WAVE_SYNC();
}

// Forward half transform of a polynomial of gadget digits (|d| <= 2^9): d[j2] = coefficient lane + 64 j2, j2 < 16.
// out[b]: a 64-bit representative (not canonical) of A_k, k = nth_freq_index(H, lane, b).
template <int H>
FF_FN void nth_forward_small(u64 (&out)[8], const i32 (&d)[16], u64 *xbuf, const NthTables &T, int lane)
{
    constexpr int C = 2 * H + 1;
    L4 x[8];
    // y = (lo +- i hi) 2^(6 C j2'): both digits land in ONE limb each, two limbs apart (i = 2^48); |limb| <= 2^27
#define PLACE(j)                                                              \
    do {                                                                      \
        L4 p, q;                                                              \
        l4_place<6 * C * (j)>(p, d[j]);                                       \
        l4_place<6 * C * (j) + 48>(q, d[(j) + 8]);                            \
        if (H == 0) l4_add(x[j], p, q); else l4_sub(x[j], p, q);              \
    } while (0)
    PLACE(0); PLACE(1); PLACE(2); PLACE(3); PLACE(4); PLACE(5); PLACE(6); PLACE(7);
#undef PLACE
    // pass 1: every output limb collects at most one limb of each of the 8 inputs: |limb| <= 2^30
    l4_ntt8_dif<1>(x);
    const int k2 = lane >> 3, q = lane & 7;
    // layer 1: (lane j1, register kappa2 = br3(i)) -> (lane (kappa2, q), register r)
    nth_layer(x, xbuf, [&](int i) { return nth_br3(i) * NTH_ROW + lane; }, [&](int r) { return k2 * NTH_ROW + q + 8 * r; },
              [&](int r) { return T.fwd1[r * 64 + lane]; });
    // products: limbs in (-2^24, 2^24); pass 2 (r -> a): <= 2^27
    l4_ntt8_dif<1>(x);
    // layer 2: (lane (kappa2, q), register a = br3(i)) -> (lane (kappa2, a), register q'), factor 2^(3 a q')
    nth_layer(x, xbuf, [&](int i) { return k2 * NTH_ROW + nth_br3(i) * 8 + q; }, [&](int qq) { return k2 * NTH_ROW + q * 8 + qq; },
              [&](int qq) { return T.fwd2[qq * 8 + q]; });
    // pass 3 (q' -> b): <= 2^27, then the packed representatives in natural order of b
    l4_ntt8_dif<1>(x);
#pragma unroll
    for (int i = 0; i < 8; i++) out[nth_br3(i)] = l4_to_u64(x[i]);
}

// Inverse half transform: in[b] = field element (any 64-bit representative) of frequency nth_freq_index(H, lane, b);
// out[j2'] = a 64-bit representative of Y^H_j', j' = lane + 64 j2' (1/1024 included).
template <int H>
FF_FN void nth_inverse(u64 (&out)[8], const u64 (&in)[8], u64 *xbuf, const NthTables &T, int lane)
{
    constexpr int C = 2 * H + 1;
    L4 x[8];
#pragma unroll
    for (int b = 0; b < 8; b++) l4_from_u64(x[b], in[b]);          // limbs < 2^24
    // pass 3 backwards (b -> q'): <= 2^27
    l4_ntt8_dif<-1>(x);
    const int k2 = lane >> 3, a = lane & 7;                        // this lane is (kappa2, a)
    // layer 2 backwards: (lane (kappa2, a), register q' = br3(i)) -> (lane (kappa2, q), register a'), factor 2^(-3 a' q)
    nth_layer(x, xbuf, [&](int i) { return k2 * NTH_ROW + a * 8 + nth_br3(i); }, [&](int aa) { return k2 * NTH_ROW + aa * 8 + a; },
              [&](int aa) { return T.inv2[aa * 8 + a]; });
    // (from here on `a` plays the part of q: the lane is (kappa2, q)); pass 2 backwards (a' -> r): <= 2^27
    l4_ntt8_dif<-1>(x);
    // layer 1 backwards: (lane (kappa2, q), register r = br3(i)) -> (lane j1, register kappa2'), j1 = q + 8 r
    nth_layer(x, xbuf, [&](int i) { return k2 * NTH_ROW + a + 8 * nth_br3(i); }, [&](int kk) { return kk * NTH_ROW + lane; },
              [&](int kk) { return T.inv1[kk * 64 + lane]; });
    // pass 1 backwards (kappa2 -> j2'): <= 2^27, post-twist 2^(-6 C j2') and packing
    l4_ntt8_dif<-1>(x);
#define POST(j)                                                               \
    do {                                                                      \
        L4 y;                                                                 \
        l4_mul_pow2<-6 * C * (j)>(y, x[nth_br3(j)]);                          \
        out[j] = l4_to_u64(y);                                                \
    } while (0)
    POST(0); POST(1); POST(2); POST(3); POST(4); POST(5); POST(6); POST(7);
#undef POST
}

// host-side construction of the table block (NTH_TABLE_ELEMS u64), same root of unity as ntt_tables.h
static inline void nth_make_tables(u64 *block)
{
    const u64 psi = ntt_host_pow(NTT_ROOT_2_32, (1ULL << 32) / 2048);
    const u64 psi_inv = ntt_host_pow(psi, FF_P - 2);
    const u64 n_inv = ntt_host_pow(1024, FF_P - 2);
    const u64 two_inv = ntt_host_pow(2, FF_P - 2);
    for (int h = 0; h < 2; h++) {
        const int c = 2 * h + 1;
        for (int r = 0; r < 8; r++)
            for (int lane = 0; lane < 64; lane++) {
                const int k2 = lane >> 3, q = lane & 7;
                block[h * NTH_TW1_ELEMS + r * 64 + lane] = ntt_host_pow(psi, (u64)(4 * k2 + c) * (u64)(q + 8 * r));
            }
        for (int k2 = 0; k2 < 8; k2++)
            for (int j1 = 0; j1 < 64; j1++)
                block[2 * NTH_TW1_ELEMS + h * NTH_TW1_ELEMS + k2 * 64 + j1] =
                    ff_mul(ntt_host_pow(psi_inv, (u64)(4 * k2 + c) * (u64)j1), n_inv);
    }
    for (int q = 0; q < 8; q++)
        for (int a = 0; a < 8; a++) {
            block[4 * NTH_TW1_ELEMS + q * 8 + a] = ntt_host_pow(2, (u64)(3 * a * q));
            block[4 * NTH_TW1_ELEMS + NTH_TW2_ELEMS + a * 8 + q] = ntt_host_pow(two_inv, (u64)(3 * a * q));
        }
}
