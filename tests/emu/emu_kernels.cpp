// emu_kernels.cpp -- TEST INFRASTRUCTURE: C entry points that run the wave-level device code of
// nufhe_amd/csrc/*.h on the CPU through the fibre emulator (tests/emu/emu_wave.*), so that the
// index math / arithmetic of the HIP kernels can be checked against the oracle without a GPU.
#include <cstring>
#include <vector>

#define NUFHE_EMU 1
#include "../../nufhe_amd/csrc/ff.h"
#include "../../nufhe_amd/csrc/ntt1024.h"
#include "../../nufhe_amd/csrc/ntt_tables.h"
#include "../../nufhe_amd/csrc/blind_rotate.h"
#include "../../nufhe_amd/csrc/ff24.h"
#include "../../nufhe_amd/csrc/ntt1024_l4.h"
#include "../../nufhe_amd/csrc/ntt512_half.h"
#include "../../nufhe_amd/csrc/l4_hook.h"
#include "../../nufhe_amd/csrc/fft512.h"
#include "../../nufhe_amd/csrc/blind_rotate_fft.h"
#include "../../nufhe_amd/csrc/blind_rotate_xfft.h"
#include "emu_wave.h"

double g_emu_fft_max_frac = 0.0, g_emu_fft_max_abs = 0.0;   // fft512.h (NUFHE_EMU)
double g_emu_xfft_max_frac = 0.0, g_emu_xfft_max_abs = 0.0; // blind_rotate_xfft.h (NUFHE_EMU)
static u64 g_tw1f[1024], g_tw1i[1024], g_tw1x[1024];
static bool g_tables = false;
static void tables()
{
    if (!g_tables) { ntt_make_tables(g_tw1f, g_tw1i); ntt_make_tw1x(g_tw1x, g_tw1f); g_tables = true; }
}

template <int S> static void lsh_all(u64 *r, const u64 *a, long n) { for (long i = 0; i < n; i++) r[i] = ff_mul_pow2<S>(a[i]); }
template <int S> struct LshTable {
    static void fill(void (**t)(u64 *, const u64 *, long)) { t[S] = lsh_all<S>; LshTable<S - 1>::fill(t); }
};
template <> struct LshTable<-1> { static void fill(void (**)(u64 *, const u64 *, long)) {} };

extern "C" {

// ff primitives (host build of the same inline functions the GPU uses)
void emu_ff_add(u64 *r, const u64 *a, const u64 *b, long n) { for (long i = 0; i < n; i++) r[i] = ff_add(a[i], b[i]); }
void emu_ff_sub(u64 *r, const u64 *a, const u64 *b, long n) { for (long i = 0; i < n; i++) r[i] = ff_sub(a[i], b[i]); }
void emu_ff_mul(u64 *r, const u64 *a, const u64 *b, long n) { for (long i = 0; i < n; i++) r[i] = ff_mul(a[i], b[i]); }
void emu_ff_lsh_var(u64 *r, const u64 *a, const u32 *s, long n) { for (long i = 0; i < n; i++) r[i] = ff_mul_pow2_var(a[i], s[i]); }

// r = a * 2^s for every compile-time instantiation s in [0,192)
void emu_ff_lsh_const(u64 *r, const u64 *a, int s, long n)
{
    static void (*table[192])(u64 *, const u64 *, long);
    static bool init = false;
    if (!init) { LshTable<191>::fill(table); init = true; }
    table[s](r, a, n);
}

// forward NTT of one polynomial of canonical field elements; out in natural frequency order
void emu_ntt_forward(u64 *out, const u64 *in)
{
    tables();
    std::vector<u64> xbuf(NTT_XBUF_ELEMS);
    emu_run_wave([&](int lane) {
        NttLane L = ntt_lane_init(lane);
        u64 x[16];
        for (int r = 0; r < 16; r++) x[r] = in[ntt_coef_index(lane, r)];
        ntt_forward(x, xbuf.data(), g_tw1f, L);
        for (int r = 0; r < 16; r++) out[ntt_freq_index(lane, r)] = x[r];
    });
}

void emu_ntt_inverse(u64 *out, const u64 *in)
{
    tables();
    std::vector<u64> xbuf(NTT_XBUF_ELEMS);
    emu_run_wave([&](int lane) {
        NttLane L = ntt_lane_init(lane);
        u64 x[16];
        for (int r = 0; r < 16; r++) x[r] = in[ntt_freq_index(lane, r)];
        ntt_inverse(x, xbuf.data(), g_tw1i, L);
        for (int r = 0; r < 16; r++) out[ntt_coef_index(lane, r)] = x[r];
    });
}


// limb-form (ff24.h) primitives through the shared test-hook dispatcher; arrays are u32[n][4]
void emu_l4_op(u32 *out, u32 *out2, const u32 *a, const u32 *b, const u32 *c, int op, int shift, long n)
{
    for (long i = 0; i < n; i++)
        l4_hook(out + 4 * i, out2 + 4 * i, a + 4 * i, b ? b + 4 * i : nullptr, c ? c + 4 * i : nullptr, op, shift);
}

// limb-form forward transform of a digit polynomial; out: 64-bit representatives, natural frequency order
void emu_ntt_forward_small_l4(u64 *out, const i32 *in)
{
    tables();
    std::vector<u64> xbuf(NTT_XBUF_ELEMS);
    emu_run_wave([&](int lane) {
        NttLane L = ntt_lane_init(lane);
        i32 d[16];
        u64 x[16];
        for (int r = 0; r < 16; r++) d[r] = in[ntt_coef_index(lane, r)];
        ntt_forward_small_l4(x, d, xbuf.data(), g_tw1x, L);
        for (int r = 0; r < 16; r++) out[ntt_freq_index(lane, r)] = x[r];
    });
}

// limb-form inverse transform down to coefficients mod 2^32 (sign convention undone here)
void emu_ntt_inverse_l4_i32(u32 *out, const u64 *in)
{
    tables();
    std::vector<u64> xbuf(NTT_XBUF_ELEMS);
    emu_run_wave([&](int lane) {
        NttLane L = ntt_lane_init(lane);
        u64 x[16];
        u32 c[16];
        for (int r = 0; r < 16; r++) x[r] = in[ntt_freq_index(lane, r)];
        ntt_inverse_l4_i32(c, x, xbuf.data(), g_tw1i, L);
        for (int r = 0; r < 16; r++) out[ntt_coef_index(lane, r)] = r == 0 ? c[r] : 0u - c[r];
    });
}

// half-ring transforms (ntt512_half.h): the forward transform of a digit polynomial by two waves (h = 0: even, h = 1: odd
// output frequencies); out: canonical values in natural frequency order of the 1024-point transform
static u64 g_nth[NTH_TABLE_ELEMS];
static bool g_nth_ready = false;
static void nth_ready() { if (!g_nth_ready) { nth_make_tables(g_nth); g_nth_ready = true; } }

void emu_nth_forward_small(u64 *out, const i32 *in)
{
    nth_ready();
    for (int h = 0; h < 2; h++) {
        std::vector<u64> xbuf(NTH_XBUF_ELEMS);
        const NthTables T = nth_tables(g_nth, h);
        emu_run_wave([&](int lane) {
            i32 d[16];
            u64 x[8];
            for (int r = 0; r < 16; r++) d[r] = in[lane + 64 * r];
            if (h == 0) nth_forward_small<0>(x, d, xbuf.data(), T, lane); else nth_forward_small<1>(x, d, xbuf.data(), T, lane);
            for (int b = 0; b < 8; b++) out[nth_freq_index(h, lane, b)] = ff_canon(x[b]);
        });
    }
}

// inverse of the whole 1024-point transform through the two half rings + the join; in: canonical values in natural
// frequency order; out: coefficients mod 2^32 (the caller guarantees small integer coefficients, as ntt_inverse_l4_i32)
void emu_nth_inverse_i32(u32 *out, const u64 *in)
{
    nth_ready();
    std::vector<u64> Y(2 * 512);
    for (int h = 0; h < 2; h++) {
        std::vector<u64> xbuf(NTH_XBUF_ELEMS);
        const NthTables T = nth_tables(g_nth, h);
        emu_run_wave([&](int lane) {
            u64 x[8], y[8];
            for (int b = 0; b < 8; b++) x[b] = in[nth_freq_index(h, lane, b)];
            if (h == 0) nth_inverse<0>(y, x, xbuf.data(), T, lane); else nth_inverse<1>(y, x, xbuf.data(), T, lane);
            for (int j = 0; j < 8; j++) Y[h * 512 + lane + 64 * j] = ff_canon(y[j]);
        });
    }
    for (int j = 0; j < 512; j++) {
        const u64 lo = ff_add(Y[j], Y[512 + j]);
        const u64 hi = ff_mul_pow2<48>(ff_sub(Y[512 + j], Y[j]));
        out[j] = (u32)ff_to_i32(lo);
        out[j + 512] = (u32)ff_to_i32(hi);
    }
}

// reference-format key polynomials (natural-order NTT, Montgomery) -> wave layout, plain
void emu_bk_from_reference(u64 *out, const u64 *in, long polys)
{
    for (long p = 0; p < polys; p++)
        for (int lane = 0; lane < 64; lane++)
            for (int r = 0; r < 16; r++)
                out[p * 1024 + bk_elem_offset(lane, r)] = ff_mul_pow2<128>(in[p * 1024 + ntt_freq_index(lane, r)]);
}

}  // extern "C" (templates need C++ linkage)

// one bit: tmp = (0, c0) + p0 * src0 + p1 * src1 -> bootstrap without keyswitch (mask size K)
template <int K>
static void bootstrap_bit(i32 *out_a, i32 *out_b, const u64 *bk_internal, int n,
                          const i32 *a0, const i32 *b0, i32 p0, const i32 *a1, const i32 *b1, i32 p1,
                          i32 c0, i32 mu)
{
    tables();
    std::vector<u64> xbuf(NTT_XBUF_ELEMS);
    std::vector<i32> accbuf((K + 1) * 1024);
    std::vector<uint16_t> bara(BR_MAX_LWE);
    BrLds lds{xbuf.data(), accbuf.data(), bara.data(), g_tw1x, g_tw1i, {nullptr, nullptr}};
    BrSource s0{a0, b0, 0, 0, p0}, s1{a1, b1, 0, 0, p1};
    emu_run_wave([&](int lane) {
        NttLane L = ntt_lane_init(lane);
        u32 barb = br_prologue(s0, s1, c0, 0, n, lds, lane);
        WAVE_SYNC();
        u32 acc[K + 1][16];
        br_blind_rotate<K>(acc, bk_internal, n, barb, mu, lds, L);
        br_extract<K>(out_a, out_b, acc, lane);
    });
}

// the 4-wave team variant (blind_rotate.h, brt_*): 256 fibres, work-group barrier = emu_team_sync
extern "C" void emu_bootstrap_bit_team(i32 *out_a, i32 *out_b, const u64 *bk_internal, int n,
                                       const i32 *a0, const i32 *b0, i32 p0, const i32 *a1, const i32 *b1, i32 p1,
                                       i32 c0, i32 mu)
{
    tables();
    std::vector<u64> xbuf(BRT_WAVES * NTT_XBUF_ELEMS), part(BRT_PART_ELEMS);
    std::vector<i32> accbuf(2 * 1024);
    std::vector<uint16_t> bara(BR_MAX_LWE);
    BrSource s0{a0, b0, 0, 0, p0}, s1{a1, b1, 0, 0, p1};
    emu_run_team(BRT_WAVES, [&](int w, int lane) {
        BrTeamLds lds{xbuf.data() + w * NTT_XBUF_ELEMS, accbuf.data(), bara.data(), part.data(), g_tw1x, g_tw1i};
        NttLane L = ntt_lane_init(lane);
        brt_bootstrap(out_a, out_b, s0, s1, c0, 0, bk_internal, n, mu, lds, L, w, [] { emu_team_sync(); });
    });
}

// wave-layout key (bk_internal) -> half-ring layout [poly][h][reg 8][lane 64] (the permutation of k_bk_to_half)
extern "C" void emu_bk_to_half(u64 *out, const u64 *in, long polys)
{
    for (long p = 0; p < polys; p++)
        for (int h = 0; h < 2; h++)
            for (int lane = 0; lane < 64; lane++)
                for (int r = 0; r < 8; r++) {
                    const int k = nth_freq_index(h, lane, r);
                    const int k2 = k & 15, k1a = (k >> 4) & 15, k1b = k >> 8;
                    const int sl = 4 * k2 + (k1a & 3), sr = 4 * (k1a >> 2) + k1b;      // inverse of ntt_freq_index
                    out[p * 1024 + h * 512 + r * 64 + lane] = in[p * 1024 + bk_elem_offset(sl, sr)];
                }
}

// the 8-wave half-ring team (blind_rotate.h, brh_*): 512 fibres; the key in the half-ring layout
extern "C" void emu_bootstrap_bit_team8(i32 *out_a, i32 *out_b, const u64 *bk_half, int n,
                                        const i32 *a0, const i32 *b0, i32 p0, const i32 *a1, const i32 *b1, i32 p1,
                                        i32 c0, i32 mu)
{
    nth_ready();
    std::vector<u64> xbuf(BRH_WAVES * NTH_XBUF_ELEMS), part(BRH_PART_ELEMS), join(BRH_JOIN_ELEMS);
    std::vector<i32> accbuf(2 * 1024);
    std::vector<uint16_t> bara(BR_MAX_LWE);
    BrSource s0{a0, b0, 0, 0, p0}, s1{a1, b1, 0, 0, p1};
    emu_run_team(BRH_WAVES, [&](int w, int lane) {
        BrHalfLds lds{xbuf.data() + w * NTH_XBUF_ELEMS, accbuf.data(), bara.data(), part.data(), join.data(), g_nth};
        brh_bootstrap(out_a, out_b, s0, s1, c0, 0, bk_half, n, mu, lds, lane, w, [] { emu_team_sync(); });
    });
}

// the 2-wave pair variant (blind_rotate.h, brp_*): 128 fibres, pair barrier = emu_team_sync
extern "C" void emu_bootstrap_bit_pair(i32 *out_a, i32 *out_b, const u64 *bk_internal, int n,
                                       const i32 *a0, const i32 *b0, i32 p0, const i32 *a1, const i32 *b1, i32 p1,
                                       i32 c0, i32 mu)
{
    tables();
    std::vector<u64> xbuf(2 * NTT_XBUF_ELEMS);
    std::vector<i32> accbuf(2 * 1024);
    std::vector<uint16_t> bara(BR_MAX_LWE);
    BrSource s0{a0, b0, 0, 0, p0}, s1{a1, b1, 0, 0, p1};
    emu_run_team(2, [&](int w, int lane) {
        BrPairLds lds{xbuf.data() + w * NTT_XBUF_ELEMS, xbuf.data() + (w ^ 1) * NTT_XBUF_ELEMS, accbuf.data(), bara.data(),
                      g_tw1x, g_tw1i, {nullptr, nullptr}};
        NttLane L = ntt_lane_init(lane);
        if (w == 0) brp_bootstrap<0>(out_a, out_b, s0, s1, c0, 0, bk_internal, n, mu, lds, L, [] { emu_team_sync(); });
        else brp_bootstrap<1>(out_a, out_b, s0, s1, c0, 0, bk_internal, n, mu, lds, L, [] { emu_team_sync(); });
    });
}

// the 3-wave team variant for tlwe_mask_size = 2 (blind_rotate.h, brtk_*): 192 fibres
extern "C" void emu_bootstrap_bit_team_k2(i32 *out_a, i32 *out_b, const u64 *bk_internal, int n,
                                          const i32 *a0, const i32 *b0, i32 p0, const i32 *a1, const i32 *b1, i32 p1,
                                          i32 c0, i32 mu)
{
    tables();
    std::vector<u64> xbuf(3 * NTT_XBUF_ELEMS), part(BRTK_PART_ELEMS(2));
    std::vector<i32> accbuf(3 * 1024);
    std::vector<uint16_t> bara(BR_MAX_LWE);
    BrSource s0{a0, b0, 0, 0, p0}, s1{a1, b1, 0, 0, p1};
    emu_run_team(3, [&](int w, int lane) {
        BrTeamLds lds{xbuf.data() + w * NTT_XBUF_ELEMS, accbuf.data(), bara.data(), part.data(), g_tw1x, g_tw1i};
        NttLane L = ntt_lane_init(lane);
        brtk_bootstrap<2>(out_a, out_b, s0, s1, c0, 0, bk_internal, n, mu, lds, L, w, [] { emu_team_sync(); });
    });
}

// the 3-wave ring variant for tlwe_mask_size = 2 (blind_rotate.h, brr_*): partial sums handed round through the
// exchange buffers
extern "C" void emu_bootstrap_bit_ring_k2(i32 *out_a, i32 *out_b, const u64 *bk_internal, int n,
                                          const i32 *a0, const i32 *b0, i32 p0, const i32 *a1, const i32 *b1, i32 p1,
                                          i32 c0, i32 mu)
{
    tables();
    std::vector<u64> xbuf(3 * NTT_XBUF_ELEMS);
    std::vector<i32> accbuf(3 * 1024);
    std::vector<uint16_t> bara(BR_MAX_LWE);
    BrSource s0{a0, b0, 0, 0, p0}, s1{a1, b1, 0, 0, p1};
    emu_run_team(3, [&](int w, int lane) {
        BrRingLds lds{xbuf.data(), accbuf.data(), bara.data(), g_tw1x, g_tw1i, {nullptr, nullptr}};
        NttLane L = ntt_lane_init(lane);
        brr_bootstrap<2>(out_a, out_b, s0, s1, c0, 0, bk_internal, n, mu, lds, L, w, [] { emu_team_sync(); });
    });
}

extern "C" void emu_bootstrap_bit(i32 *out_a, i32 *out_b, const u64 *bk_internal, int n,
                                  const i32 *a0, const i32 *b0, i32 p0, const i32 *a1, const i32 *b1, i32 p1,
                                  i32 c0, i32 mu)
{
    bootstrap_bit<1>(out_a, out_b, bk_internal, n, a0, b0, p0, a1, b1, p1, c0, mu);
}

extern "C" void emu_bootstrap_bit_k2(i32 *out_a, i32 *out_b, const u64 *bk_internal, int n,
                                     const i32 *a0, const i32 *b0, i32 p0, const i32 *a1, const i32 *b1, i32 p1,
                                     i32 c0, i32 mu)
{
    bootstrap_bit<2>(out_a, out_b, bk_internal, n, a0, b0, p0, a1, b1, p1, c0, mu);
}

extern "C" {

// reads and resets the rounding-margin statistics gathered by fft_round_to_u32 (host build only)
void emu_fft_margin(double *max_frac, double *max_abs)
{
    *max_frac = g_emu_fft_max_frac; *max_abs = g_emu_fft_max_abs;
    g_emu_fft_max_frac = 0.0; g_emu_fft_max_abs = 0.0;
}

// the final rounding of the FFT path as compiled from fft512.h
void emu_fft_round(uint32_t *out, const double *in, long n)
{
    for (long i = 0; i < n; i++) out[i] = fft_round_to_u32(in[i]);
}

static cplx g_ftw1[FFT_TW1_ELEMS], g_ftw2[FFT_TW2_ELEMS];
static bool g_ftables = false;
static void ftables() { if (!g_ftables) { fft_make_tables(g_ftw1, g_ftw2); g_ftables = true; } }

// forward folded FFT of one int32 polynomial; out: 512 complex (re, im interleaved), natural order
void emu_fft_forward(double *out, const i32 *in)
{
    ftables();
    std::vector<cplx> xbuf(FFT_XBUF_ELEMS);
    emu_run_wave([&](int lane) {
        FftLane L = fft_lane_init(lane);
        cplx x[8];
        for (int r = 0; r < 8; r++) x[r] = cplx{(double)in[lane + 64 * r], -(double)in[lane + 64 * r + 512]};
        fft_forward(x, xbuf.data(), g_ftw1, g_ftw2, L);
        for (int r = 0; r < 8; r++) { int k = fft_freq_index(lane, r); out[2 * k] = x[r].re; out[2 * k + 1] = x[r].im; }
    });
}

void emu_fft_inverse(i32 *out, const double *in)
{
    ftables();
    std::vector<cplx> xbuf(FFT_XBUF_ELEMS);
    emu_run_wave([&](int lane) {
        FftLane L = fft_lane_init(lane);
        cplx x[8];
        for (int r = 0; r < 8; r++) { int k = fft_freq_index(lane, r); x[r] = cplx{in[2 * k], in[2 * k + 1]}; }
        fft_inverse(x, xbuf.data(), g_ftw1, g_ftw2, L);
        for (int r = 0; r < 8; r++) {
            out[lane + 64 * r] = (i32)fft_round_to_u32(x[r].re);
            out[lane + 64 * r + 512] = (i32)fft_round_to_u32(-x[r].im);
        }
    });
}


// natural-order FFT-domain key polynomials (complex128 [polys][512]) -> wave layout
void emu_bkf_from_reference(double *out, const double *in, long polys)
{
    for (long p = 0; p < polys; p++)
        for (int lane = 0; lane < 64; lane++)
            for (int r = 0; r < 8; r++) {
                const long o = p * 512 + bkf_elem_offset(lane, r), i = p * 512 + fft_freq_index(lane, r);
                out[2 * o] = in[2 * i]; out[2 * o + 1] = in[2 * i + 1];
            }
}

// 4-wave team variant of the FFT body
void emu_bootstrap_bit_fft_team(i32 *out_a, i32 *out_b, const double *bk_internal, int n,
                                const i32 *a0, const i32 *b0, i32 p0, const i32 *a1, const i32 *b1, i32 p1,
                                i32 c0, i32 mu)
{
    ftables();
    std::vector<cplx> xbuf(BRT_WAVES * FFT_XBUF_ELEMS), part(BRFT_PART_ELEMS);
    std::vector<i32> accbuf(2 * 1024);
    std::vector<uint16_t> bara(BR_MAX_LWE);
    BrSource s0{a0, b0, 0, 0, p0}, s1{a1, b1, 0, 0, p1};
    emu_run_team(BRT_WAVES, [&](int w, int lane) {
        BrFftTeamLds lds{xbuf.data() + w * FFT_XBUF_ELEMS, accbuf.data(), bara.data(), part.data(), g_ftw1, g_ftw2};
        FftLane L = fft_lane_init(lane);
        brft_bootstrap(out_a, out_b, s0, s1, c0, 0, (const cplx *)bk_internal, n, mu, lds, L, w, [] { emu_team_sync(); });
    });
}

// 2-wave pair variant of the FFT body (brfp_*): 128 fibres
void emu_bootstrap_bit_fft_pair(i32 *out_a, i32 *out_b, const double *bk_internal, int n,
                                const i32 *a0, const i32 *b0, i32 p0, const i32 *a1, const i32 *b1, i32 p1,
                                i32 c0, i32 mu)
{
    ftables();
    std::vector<cplx> xbuf(4 * FFT_XBUF_ELEMS);
    std::vector<i32> accbuf(2 * 1024);
    std::vector<uint16_t> bara(BR_MAX_LWE);
    BrSource s0{a0, b0, 0, 0, p0}, s1{a1, b1, 0, 0, p1};
    emu_run_team(2, [&](int w, int lane) {
        BrFftPairLds lds{xbuf.data() + w * 2 * FFT_XBUF_ELEMS, xbuf.data() + (w * 2 + 1) * FFT_XBUF_ELEMS,
                         xbuf.data() + (w ^ 1) * 2 * FFT_XBUF_ELEMS, accbuf.data(), bara.data(), g_ftw1, g_ftw2,
                         {nullptr, nullptr}};
        FftLane L = fft_lane_init(lane);
        if (w == 0) brfp_bootstrap<0>(out_a, out_b, s0, s1, c0, 0, (const cplx *)bk_internal, n, mu, lds, L, [] { emu_team_sync(); });
        else brfp_bootstrap<1>(out_a, out_b, s0, s1, c0, 0, (const cplx *)bk_internal, n, mu, lds, L, [] { emu_team_sync(); });
    });
}

// tlwe_mask_size = 2 with the FFT transform (brfk_*): out_a has 2048 entries
void emu_bootstrap_bit_fft_k2(i32 *out_a, i32 *out_b, const double *bk_internal, int n,
                              const i32 *a0, const i32 *b0, i32 p0, const i32 *a1, const i32 *b1, i32 p1,
                              i32 c0, i32 mu)
{
    ftables();
    std::vector<cplx> xbuf(2 * FFT_XBUF_ELEMS);
    std::vector<i32> accbuf(3 * 1024);
    BrFftLdsK lds{xbuf.data(), xbuf.data() + FFT_XBUF_ELEMS, accbuf.data(), g_ftw1, g_ftw2};
    BrSource s0{a0, b0, 0, 0, p0}, s1{a1, b1, 0, 0, p1};
    emu_run_wave([&](int lane) {
        FftLane L = fft_lane_init(lane);
        u32 acc[3][16];
        brfk_bootstrap_body<2>(acc, s0, s1, c0, 0, (const cplx *)bk_internal, n, mu, lds, L);
        br_extract<2>(out_a, out_b, acc, lane);
    });
}

// the 3-wave ring variant of the k = 2 FFT body (brfr_*): no partial-sum buffer
void emu_bootstrap_bit_fft_ring_k2(i32 *out_a, i32 *out_b, const double *bk_internal, int n,
                                   const i32 *a0, const i32 *b0, i32 p0, const i32 *a1, const i32 *b1, i32 p1,
                                   i32 c0, i32 mu)
{
    ftables();
    std::vector<cplx> xbuf(3 * 2 * FFT_XBUF_ELEMS);
    std::vector<i32> accbuf(3 * 1024);
    std::vector<uint16_t> bara(BR_MAX_LWE);
    BrSource s0{a0, b0, 0, 0, p0}, s1{a1, b1, 0, 0, p1};
    emu_run_team(3, [&](int w, int lane) {
        BrFftRingLds lds{xbuf.data(), accbuf.data(), bara.data(), g_ftw1, g_ftw2, {nullptr, nullptr}};
        FftLane L = fft_lane_init(lane);
        brfr_bootstrap<2>(out_a, out_b, s0, s1, c0, 0, (const cplx *)bk_internal, n, mu, lds, L, w, [] { emu_team_sync(); });
    });
}

// the 3-wave team variant of the k = 2 FFT body (brftk_*): 192 fibres
void emu_bootstrap_bit_fft_team_k2(i32 *out_a, i32 *out_b, const double *bk_internal, int n,
                                   const i32 *a0, const i32 *b0, i32 p0, const i32 *a1, const i32 *b1, i32 p1,
                                   i32 c0, i32 mu)
{
    ftables();
    std::vector<cplx> xbuf(3 * 2 * FFT_XBUF_ELEMS), part(BRFTK_PART_ELEMS(2));
    std::vector<i32> accbuf(3 * 1024);
    std::vector<uint16_t> bara(BR_MAX_LWE);
    BrSource s0{a0, b0, 0, 0, p0}, s1{a1, b1, 0, 0, p1};
    emu_run_team(3, [&](int w, int lane) {
        BrFftTeamLdsK lds{xbuf.data() + w * 2 * FFT_XBUF_ELEMS, xbuf.data() + (w * 2 + 1) * FFT_XBUF_ELEMS, accbuf.data(),
                          bara.data(), part.data(), g_ftw1, g_ftw2};
        FftLane L = fft_lane_init(lane);
        brftk_bootstrap<2>(out_a, out_b, s0, s1, c0, 0, (const cplx *)bk_internal, n, mu, lds, L, w, [] { emu_team_sync(); });
    });
}

void emu_bootstrap_bit_fft(i32 *out_a, i32 *out_b, const double *bk_internal, int n,
                           const i32 *a0, const i32 *b0, i32 p0, const i32 *a1, const i32 *b1, i32 p1,
                           i32 c0, i32 mu)
{
    ftables();
    std::vector<cplx> xbuf(2 * FFT_XBUF_ELEMS);
    std::vector<u32> park((BRF_PARK - 4) * 64);
    BrFftLds lds{xbuf.data(), xbuf.data() + FFT_XBUF_ELEMS, park.data(), g_ftw1, g_ftw2, {nullptr, nullptr}};
    BrSource s0{a0, b0, 0, 0, p0}, s1{a1, b1, 0, 0, p1};
    emu_run_wave([&](int lane) {
        FftLane L = fft_lane_init(lane);
        u32 barb = brf_prologue(s0, s1, c0, 0, n, lds, lane);
        WAVE_SYNC();
        u32 acc[2][16];
        brf_blind_rotate(acc, (const cplx *)bk_internal, n, barb, mu, lds, L);
        br_extract<1>(out_a, out_b, acc, lane);
    });
}

// ---- exact-FFT engine (blind_rotate_xfft.h) ----------------------------------------------------------------------

// distance of the values handed to xfft_round_lo32 from the nearest integer, and their largest magnitude; reset on read
void emu_xfft_margin(double *max_frac, double *max_abs)
{
    *max_frac = g_emu_xfft_max_frac; *max_abs = g_emu_xfft_max_abs;
    g_emu_xfft_max_frac = 0.0; g_emu_xfft_max_abs = 0.0;
}

// int32 TGSW polynomials [polys][1024] -> split key image complex128 [polys][half 2][reg 8][lane 64]
void emu_bkx_from_coeffs(double *out, const i32 *in, long polys)
{
    ftables();
    std::vector<cplx> xbuf(FFT_XBUF_ELEMS);
    cplx *o = (cplx *)out;
    for (long p = 0; p < polys; p++)
        for (int h = 0; h < 2; h++)
            emu_run_wave([&](int lane) {
                FftLane L = fft_lane_init(lane);
                cplx x[8];
                for (int r = 0; r < 8; r++) {
                    i32 lo0, hi0, lo1, hi1;
                    xfft_split(in[p * 1024 + lane + 64 * r], lo0, hi0);
                    xfft_split(in[p * 1024 + lane + 64 * r + 512], lo1, hi1);
                    x[r] = h == 0 ? cplx{(double)lo0, -(double)lo1} : cplx{(double)hi0, -(double)hi1};
                }
                fft_forward(x, xbuf.data(), g_ftw1, g_ftw2, L);
                for (int r = 0; r < 8; r++) o[(p * 2 + h) * BKF_POLY_ELEMS + bkf_elem_offset(lane, r)] = x[r];
            });
}

// one external product: res[2][1024] = sum_{m,d} digit_d(T_m) (*) row[m][d][:]  (mod 2^32), row in the split layout
void emu_xfft_external_product(i32 *res, const i32 *T, const double *row)
{
    ftables();
    std::vector<cplx> xbuf(2 * FFT_XBUF_ELEMS);
    BrXfftLds lds{xbuf.data(), xbuf.data() + FFT_XBUF_ELEMS, g_ftw1, g_ftw2, {nullptr, nullptr}, nullptr};
    emu_run_wave([&](int lane) {
        FftLane L = fft_lane_init(lane);
        u32 t[2][16], r_[2][16];
        for (int m = 0; m < 2; m++)
            for (int r = 0; r < 16; r++) t[m][r] = (u32)T[m * 1024 + lane + 64 * r];
        brx_external_product(r_, t, (const cplx *)row, lds, lds.tw2, L);
        for (int m = 0; m < 2; m++)
            for (int r = 0; r < 16; r++) res[m * 1024 + lane + 64 * r] = (i32)r_[m][r];
    });
}

void emu_bootstrap_bit_xfft(i32 *out_a, i32 *out_b, const double *bkx, int n,
                            const i32 *a0, const i32 *b0, i32 p0, const i32 *a1, const i32 *b1, i32 p1,
                            i32 c0, i32 mu)
{
    ftables();
    std::vector<cplx> xbuf(2 * FFT_XBUF_ELEMS);
    std::vector<u32> park(32 * 64);
    BrXfftLds lds{xbuf.data(), xbuf.data() + FFT_XBUF_ELEMS, g_ftw1, g_ftw2, {nullptr, nullptr}, park.data()};
    BrSource s0{a0, b0, 0, 0, p0}, s1{a1, b1, 0, 0, p1};
    emu_run_wave([&](int lane) {
        FftLane L = fft_lane_init(lane);
        u32 barb = brf_prologue(s0, s1, c0, 0, n, brx_as_fft_lds(lds), lane);
        WAVE_SYNC();
        u32 acc[2][16];
        brx_blind_rotate(acc, (const cplx *)bkx, n, barb, mu, lds, L);
        br_extract<1>(out_a, out_b, acc, lane);
    });
}

// tlwe_mask_size = 2 on the exact engine: one external product, T int32 [3][1024], row = 18 split polynomials
void emu_xfft_external_product_k2(i32 *res, const i32 *T, const double *row)
{
    ftables();
    std::vector<cplx> xbuf(2 * FFT_XBUF_ELEMS);
    std::vector<i32> accv(3 * 1024);
    BrFftLdsK lds{xbuf.data(), xbuf.data() + FFT_XBUF_ELEMS, accv.data(), g_ftw1, g_ftw2};
    emu_run_wave([&](int lane) {
        FftLane L = fft_lane_init(lane);
        brxk_external_product<2>(
            [&](int m, u32(&t)[16]) { for (int r = 0; r < 16; r++) t[r] = (u32)T[m * 1024 + lane + 64 * r]; },
            [&](int mo, int r, u32 v) { res[mo * 1024 + lane + 64 * r] = (i32)v; }, (const cplx *)row, lds, L);
    });
}

void emu_bootstrap_bit_xfft_k2(i32 *out_a, i32 *out_b, const double *bkx, int n,
                               const i32 *a0, const i32 *b0, i32 p0, const i32 *a1, const i32 *b1, i32 p1,
                               i32 c0, i32 mu)
{
    ftables();
    std::vector<cplx> xbuf(2 * FFT_XBUF_ELEMS);
    std::vector<i32> accv(3 * 1024);
    BrFftLdsK lds{xbuf.data(), xbuf.data() + FFT_XBUF_ELEMS, accv.data(), g_ftw1, g_ftw2};
    BrSource s0{a0, b0, 0, 0, p0}, s1{a1, b1, 0, 0, p1};
    emu_run_wave([&](int lane) {
        FftLane L = fft_lane_init(lane);
        u32 acc[3][16];
        brxk_bootstrap_body<2>(acc, s0, s1, c0, 0, (const cplx *)bkx, n, mu, lds, L);
        br_extract<2>(out_a, out_b, acc, lane);
    });
}

// the 4-wave quad variant of the exact engine (brxq_*): 256 fibres
void emu_bootstrap_bit_xfft_quad(i32 *out_a, i32 *out_b, const double *bkx, int n,
                                 const i32 *a0, const i32 *b0, i32 p0, const i32 *a1, const i32 *b1, i32 p1,
                                 i32 c0, i32 mu, int split)
{
    ftables();
    std::vector<cplx> xbuf(8 * FFT_XBUF_ELEMS);
    std::vector<i32> accbuf(2 * 1024);
    std::vector<uint16_t> bara(BR_MAX_LWE);
    BrSource s0{a0, b0, 0, 0, p0}, s1{a1, b1, 0, 0, p1};
    emu_run_team(4, [&](int w, int lane) {
        BrXfftQuadLds lds{xbuf.data() + w * FFT_XBUF_ELEMS, xbuf.data() + (split ? 4 + w : w) * FFT_XBUF_ELEMS, xbuf.data(), accbuf.data(), bara.data(), g_ftw1, g_ftw2};
        FftLane L = fft_lane_init(lane);
        auto sync = [] { emu_team_sync(); };
        switch (w) {
        case 0: if (split) brxq_bootstrap<0, 1, true>(out_a, out_b, s0, s1, c0, 0, (const cplx *)bkx, n, mu, lds, L, sync);
                else brxq_bootstrap<0, 1, false>(out_a, out_b, s0, s1, c0, 0, (const cplx *)bkx, n, mu, lds, L, sync);
                break;
        case 1: if (split) brxq_bootstrap<1, 1, true>(out_a, out_b, s0, s1, c0, 0, (const cplx *)bkx, n, mu, lds, L, sync);
                else brxq_bootstrap<1, 1, false>(out_a, out_b, s0, s1, c0, 0, (const cplx *)bkx, n, mu, lds, L, sync);
                break;
        case 2: if (split) brxq_bootstrap<2, 1, true>(out_a, out_b, s0, s1, c0, 0, (const cplx *)bkx, n, mu, lds, L, sync);
                else brxq_bootstrap<2, 1, false>(out_a, out_b, s0, s1, c0, 0, (const cplx *)bkx, n, mu, lds, L, sync);
                break;
        default: if (split) brxq_bootstrap<3, 1, true>(out_a, out_b, s0, s1, c0, 0, (const cplx *)bkx, n, mu, lds, L, sync);
                else brxq_bootstrap<3, 1, false>(out_a, out_b, s0, s1, c0, 0, (const cplx *)bkx, n, mu, lds, L, sync);
                break;
        }
    });
}

// six waves per bit, tlwe_mask_size = 2 (brxq_* with K = 2): 384 fibres; out_a has 2048 entries
void emu_bootstrap_bit_xfft_hex_k2(i32 *out_a, i32 *out_b, const double *bkx, int n,
                                   const i32 *a0, const i32 *b0, i32 p0, const i32 *a1, const i32 *b1, i32 p1,
                                   i32 c0, i32 mu)
{
    ftables();
    std::vector<cplx> xbuf(12 * FFT_XBUF_ELEMS);
    std::vector<i32> accbuf(3 * 1024);
    std::vector<uint16_t> bara(BR_MAX_LWE);
    BrSource s0{a0, b0, 0, 0, p0}, s1{a1, b1, 0, 0, p1};
    emu_run_team(6, [&](int w, int lane) {
        BrXfftQuadLds lds{xbuf.data() + w * FFT_XBUF_ELEMS, xbuf.data() + (6 + w) * FFT_XBUF_ELEMS, xbuf.data(), accbuf.data(),
                          bara.data(), g_ftw1, g_ftw2};
        FftLane L = fft_lane_init(lane);
        auto sync = [] { emu_team_sync(); };
        switch (w) {
        case 0: brxq_bootstrap<0, 2, true>(out_a, out_b, s0, s1, c0, 0, (const cplx *)bkx, n, mu, lds, L, sync); break;
        case 1: brxq_bootstrap<1, 2, true>(out_a, out_b, s0, s1, c0, 0, (const cplx *)bkx, n, mu, lds, L, sync); break;
        case 2: brxq_bootstrap<2, 2, true>(out_a, out_b, s0, s1, c0, 0, (const cplx *)bkx, n, mu, lds, L, sync); break;
        case 3: brxq_bootstrap<3, 2, true>(out_a, out_b, s0, s1, c0, 0, (const cplx *)bkx, n, mu, lds, L, sync); break;
        case 4: brxq_bootstrap<4, 2, true>(out_a, out_b, s0, s1, c0, 0, (const cplx *)bkx, n, mu, lds, L, sync); break;
        default: brxq_bootstrap<5, 2, true>(out_a, out_b, s0, s1, c0, 0, (const cplx *)bkx, n, mu, lds, L, sync); break;
        }
    });
}

// the 4-wave quad variant of the FFT body (brfq_*): 256 fibres, two exchange buffers per wave
void emu_bootstrap_bit_fft_quad(i32 *out_a, i32 *out_b, const double *bk_internal, int n,
                                const i32 *a0, const i32 *b0, i32 p0, const i32 *a1, const i32 *b1, i32 p1,
                                i32 c0, i32 mu)
{
    ftables();
    std::vector<cplx> xbuf(8 * FFT_XBUF_ELEMS);
    std::vector<i32> accbuf(2 * 1024);
    std::vector<uint16_t> bara(BR_MAX_LWE);
    BrSource s0{a0, b0, 0, 0, p0}, s1{a1, b1, 0, 0, p1};
    emu_run_team(4, [&](int w, int lane) {
        BrFftQuadLds lds{xbuf.data() + w * FFT_XBUF_ELEMS, xbuf.data() + (4 + w) * FFT_XBUF_ELEMS, xbuf.data(),
                         xbuf.data() + (4 + (w ^ 1)) * FFT_XBUF_ELEMS, accbuf.data(), bara.data(), g_ftw1, g_ftw2};
        FftLane L = fft_lane_init(lane);
        auto sync = [] { emu_team_sync(); };
        switch (w) {
        case 0: brfq_bootstrap<0, 1>(out_a, out_b, s0, s1, c0, 0, (const cplx *)bk_internal, n, mu, lds, L, sync); break;
        case 1: brfq_bootstrap<1, 1>(out_a, out_b, s0, s1, c0, 0, (const cplx *)bk_internal, n, mu, lds, L, sync); break;
        case 2: brfq_bootstrap<2, 1>(out_a, out_b, s0, s1, c0, 0, (const cplx *)bk_internal, n, mu, lds, L, sync); break;
        default: brfq_bootstrap<3, 1>(out_a, out_b, s0, s1, c0, 0, (const cplx *)bk_internal, n, mu, lds, L, sync); break;
        }
    });
}

// six waves per bit, tlwe_mask_size = 2, FFT body (brfq_* with K = 2): 384 fibres; out_a has 2048 entries
void emu_bootstrap_bit_fft_hex_k2(i32 *out_a, i32 *out_b, const double *bk_internal, int n,
                                  const i32 *a0, const i32 *b0, i32 p0, const i32 *a1, const i32 *b1, i32 p1,
                                  i32 c0, i32 mu)
{
    ftables();
    std::vector<cplx> xbuf(12 * FFT_XBUF_ELEMS);
    std::vector<i32> accbuf(3 * 1024);
    std::vector<uint16_t> bara(BR_MAX_LWE);
    BrSource s0{a0, b0, 0, 0, p0}, s1{a1, b1, 0, 0, p1};
    emu_run_team(6, [&](int w, int lane) {
        BrFftQuadLds lds{xbuf.data() + w * FFT_XBUF_ELEMS, xbuf.data() + (6 + w) * FFT_XBUF_ELEMS, xbuf.data(),
                         xbuf.data() + (6 + (w ^ 1)) * FFT_XBUF_ELEMS, accbuf.data(), bara.data(), g_ftw1, g_ftw2};
        FftLane L = fft_lane_init(lane);
        auto sync = [] { emu_team_sync(); };
        switch (w) {
        case 0: brfq_bootstrap<0, 2>(out_a, out_b, s0, s1, c0, 0, (const cplx *)bk_internal, n, mu, lds, L, sync); break;
        case 1: brfq_bootstrap<1, 2>(out_a, out_b, s0, s1, c0, 0, (const cplx *)bk_internal, n, mu, lds, L, sync); break;
        case 2: brfq_bootstrap<2, 2>(out_a, out_b, s0, s1, c0, 0, (const cplx *)bk_internal, n, mu, lds, L, sync); break;
        case 3: brfq_bootstrap<3, 2>(out_a, out_b, s0, s1, c0, 0, (const cplx *)bk_internal, n, mu, lds, L, sync); break;
        case 4: brfq_bootstrap<4, 2>(out_a, out_b, s0, s1, c0, 0, (const cplx *)bk_internal, n, mu, lds, L, sync); break;
        default: brfq_bootstrap<5, 2>(out_a, out_b, s0, s1, c0, 0, (const cplx *)bk_internal, n, mu, lds, L, sync); break;
        }
    });
}

}  // extern "C"
