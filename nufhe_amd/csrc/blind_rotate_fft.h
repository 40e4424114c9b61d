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

struct BrFftLds {
    cplx *xbuf;          // FFT_XBUF_ELEMS complex
    i32 *acc;            // [2][1024] accumulator mirror
    uint16_t *bara;      // [BR_MAX_LWE]
    const cplx *tw1;     // [512]
    const cplx *tw2;     // [64]
};

// res[mo][r] = coefficient lane + 64 r of  sum_{m,d} digit_d(T_m) (*) BK_row[m][d][mo]
template <class TSource>
FF_FN void brf_external_product(u32 (&res)[2][16], TSource &&tsrc, const cplx *row, const BrFftLds &lds,
                                const FftLane &L)
{
    const int lane = L.lane;
    cplx sum[2][8];
#pragma unroll
    for (int r = 0; r < 8; r++) { sum[0][r] = cplx{0.0, 0.0}; sum[1][r] = cplx{0.0, 0.0}; }
#pragma unroll 1
    for (int m = 0; m < 2; m++) {
        u32 T[16];
        tsrc(m, T);
#pragma unroll 1
        for (int d = 0; d < 2; d++) {
            cplx x[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const i32 dr = d == 0 ? br_digit<0>(T[r]) : br_digit<1>(T[r]);
                const i32 di = d == 0 ? br_digit<0>(T[r + 8]) : br_digit<1>(T[r + 8]);
                x[r] = cplx{(double)dr, -(double)di};     // a_j - i a_{j+512}
            }
            fft_forward(x, lds.xbuf, lds.tw1, lds.tw2, L);
            const cplx *poly = row + (m * 2 + d) * 2 * BKF_POLY_ELEMS;
#pragma unroll
            for (int mo = 0; mo < 2; mo++) {
                const cplx *p = poly + mo * BKF_POLY_ELEMS + lane;
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const cplx k = p[r * 64];
                    sum[mo][r].re += x[r].re * k.re - x[r].im * k.im;
                    sum[mo][r].im += x[r].re * k.im + x[r].im * k.re;
                }
            }
        }
    }
#pragma unroll
    for (int mo = 0; mo < 2; mo++) {
        fft_inverse(sum[mo], lds.xbuf, lds.tw1, lds.tw2, L);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            res[mo][r] = fft_round_to_u32(sum[mo][r].re);
            res[mo][r + 8] = fft_round_to_u32(-sum[mo][r].im);
        }
    }
}

FF_FN void brf_init_acc(u32 (&acc)[2][16], u32 barb, i32 mu, i32 *lds_acc, int lane)
{
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const u32 j = (u32)(lane + 64 * r);
        const u32 t = (j + barb) & 2047u;
        acc[0][r] = 0;
        acc[1][r] = (t < 1024u) ? (u32)mu : 0u - (u32)mu;
        lds_acc[j] = 0;
        lds_acc[1024 + j] = (i32)acc[1][r];
    }
    WAVE_SYNC();
}

FF_FN void brf_step(u32 (&acc)[2][16], u32 a, const cplx *row, const BrFftLds &lds, const FftLane &L)
{
    const int lane = L.lane;
    u32 res[2][16];
    brf_external_product(
        res,
        [&](int m, u32 (&T)[16]) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const u32 j = (u32)(lane + 64 * r);
                const u32 t = (j - a) & 2047u;
                const u32 v = (u32)lds.acc[m * 1024 + (t & 1023u)];
                T[r] = ((t & 1024u) ? 0u - v : v) - acc[m][r];
            }
        },
        row, lds, L);
#pragma unroll
    for (int mo = 0; mo < 2; mo++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            acc[mo][r] += res[mo][r];
            lds.acc[mo * 1024 + lane + 64 * r] = (i32)acc[mo][r];
        }
    WAVE_SYNC();
}

FF_FN void brf_blind_rotate(u32 (&acc)[2][16], const cplx *bk, int n, u32 barb, i32 mu, const BrFftLds &lds,
                            const FftLane &L)
{
    brf_init_acc(acc, barb, mu, lds.acc, L.lane);
    for (int i = 0; i < n; i++) {
        const u32 a = WAVE_UNIFORM((u32)lds.bara[i]);
        if (a == 0) continue;
        brf_step(acc, a, bk + (long)i * BKF_ROW_ELEMS, lds, L);
    }
}
