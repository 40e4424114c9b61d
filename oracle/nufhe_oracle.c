/*
 * nufhe_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the reference's per-kernel CPU reference functions
 * (nucypher/nufhe `*_cpu.py`), composed in the order of the reference's
 * multi-kernel bootstrap driver.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library.  The product path
 * (nufhe_amd/ + libnufhe_hip.so) never links, imports or calls it.
 *
 * Parity pinning: every function here is checked in tests/test_oracle_*.py
 * against (a) golden vectors produced by the reference's own *_cpu.py files
 * executed in the build container (tests/golden/, generator:
 * tests/golden/make_golden.py) and (b) the known-answer constants the
 * reference's tests hold (root of unity, R^-1, gnum_to_i32 KATs, mod/mul/lsh
 * regressions).
 *
 * Each function cites the reference file:line it follows (paths relative to
 * the reference repo root).
 *
 * All Torus32 arithmetic is two's-complement int32 with silent wraparound
 * (done in uint32_t to avoid C signed-overflow UB).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef uint64_t u64;
typedef uint32_t u32;
typedef int32_t i32;

#define FF_P 0xFFFFFFFF00000001ULL           /* 2^64 - 2^32 + 1, nufhe/transform/ntt_cpu.py:23 */
#define FF_ROOT_2_32 0xa70dc47e4cbdf43fULL   /* nufhe/transform/ntt_cpu.py:109 */
#define FF_RINV 0xfffffffe00000001ULL        /* 2^-64 mod P, nufhe/polynomial_transform_ntt.py:66 */

/* ------------------------------------------------------------------ */
/* Finite field arithmetic, P = 2^64 - 2^32 + 1                        */
/* Behavioural spec: nufhe/transform/ntt_cpu.py:21-66 (GaloisNumber),  */
/* nufhe/transform/arithmetic.mako:78-462.                             */
/* ------------------------------------------------------------------ */

static inline u64 ff_mod(u64 x) { return x >= FF_P ? x - FF_P : x; } /* arithmetic.mako:164-194 */

static inline u64 ff_add(u64 a, u64 b) /* arithmetic.mako:78-119; a,b < P */
{
    u128 s = (u128)a + b;
    if (s >= FF_P) s -= FF_P;
    return (u64)s;
}

static inline u64 ff_sub(u64 a, u64 b) /* arithmetic.mako:122-161; a,b < P */
{
    return a >= b ? a - b : a + (FF_P - b);
}

static inline u64 ff_reduce128(u128 x)
{
    /* x = lo + 2^64 (h0 + 2^32 h1); 2^64 = 2^32 - 1, 2^96 = -1 (mod P) */
    u64 lo = (u64)x, hi = (u64)(x >> 64);
    u64 h0 = hi & 0xffffffffULL, h1 = hi >> 32;
    u64 t = ff_sub(ff_mod(lo), h1);
    u64 m = (h0 << 32) - h0; /* h0 * (2^32 - 1) < P */
    return ff_add(t, m);
}

static inline u64 ff_mul(u64 a, u64 b) /* arithmetic.mako:197-333 */
{
    return ff_reduce128((u128)a * b);
}

static u64 ff_pow(u64 x, u64 e) /* ntt_cpu.py:40-53, arithmetic.mako:422-442 */
{
    u64 r = 1;
    while (e) {
        if (e & 1) r = ff_mul(r, x);
        x = ff_mul(x, x);
        e >>= 1;
    }
    return r;
}

static inline u64 ff_inv(u64 x) { return ff_pow(x, FF_P - 2); } /* ntt_cpu.py:55-56 */

/* i32 -> field element: negative values map to P - |x| (ntt.mako:395-399) */
static inline u64 ff_from_i32(i32 x) { return x < 0 ? FF_P - (u64)(-(int64_t)x) : (u64)x; }

/* field element -> i32: values above P/2 are negative, truncated to 32 bits
 * (ntt_cpu.py:74-80 with explicit two's-complement wrap, ntt.mako:402-408) */
static inline i32 ff_to_i32(u64 v) { return (i32)((u32)v - (u32)(v > FF_P / 2)); }

/* exported element-wise wrappers (KAT tests) */
void orc_ff_add(u64 *r, const u64 *a, const u64 *b, long n) { for (long i = 0; i < n; i++) r[i] = ff_add(a[i], b[i]); }
void orc_ff_sub(u64 *r, const u64 *a, const u64 *b, long n) { for (long i = 0; i < n; i++) r[i] = ff_sub(a[i], b[i]); }
void orc_ff_mod(u64 *r, const u64 *a, long n) { for (long i = 0; i < n; i++) r[i] = ff_mod(a[i]); }
void orc_ff_mul(u64 *r, const u64 *a, const u64 *b, long n) { for (long i = 0; i < n; i++) r[i] = ff_mul(a[i], b[i]); }
/* Montgomery product a*b*2^-64, polynomial_transform_ntt.py:65-69, arithmetic.mako:355-419 */
void orc_ff_mul_prepared(u64 *r, const u64 *a, const u64 *b, long n) { for (long i = 0; i < n; i++) r[i] = ff_mul(ff_mul(a[i], b[i]), FF_RINV); }
/* x * 2^64 mod P, arithmetic.py:172-195, arithmetic.mako:336-352 */
void orc_ff_prepare_for_mul(u64 *r, const u64 *a, long n) { for (long i = 0; i < n; i++) r[i] = ff_mul(ff_mod(a[i]), 0xffffffffULL); }
void orc_ff_pow(u64 *r, const u64 *a, const u32 *e, long n) { for (long i = 0; i < n; i++) r[i] = ff_pow(a[i], e[i]); }
/* 2^-e mod P, arithmetic.mako:445-462 */
void orc_ff_inv_pow2(u64 *r, const u32 *e, long n) { for (long i = 0; i < n; i++) r[i] = ff_inv(ff_pow(2, e[i])); }
/* a * 2^s mod P for s in [0,192), arithmetic.mako:465-1045 */
void orc_ff_lsh(u64 *r, const u64 *a, const u32 *s, long n) { for (long i = 0; i < n; i++) r[i] = ff_mul(a[i], ff_pow(2, s[i])); }
void orc_ff_to_i32(i32 *r, const u64 *a, long n) { for (long i = 0; i < n; i++) r[i] = ff_to_i32(a[i]); }
void orc_ff_from_i32(u64 *r, const i32 *a, long n) { for (long i = 0; i < n; i++) r[i] = ff_from_i32(a[i]); }
u64 orc_ff_root_of_unity(u64 order) { return ff_pow(FF_ROOT_2_32, (1ULL << 32) / order); } /* ntt_cpu.py:96-109 */

/* ------------------------------------------------------------------ */
/* Negacyclic NTT, natural order in and out.                           */
/* forward: A_k = sum_j a_j psi^j w^(jk), psi = root(2N), w = psi^2    */
/*   nufhe/transform/ntt.py:30-44 (ntt_transform_ref),                 */
/*   nufhe/transform/ntt_cpu.py:145-185 (fft_generic / ntt)            */
/* ------------------------------------------------------------------ */

typedef struct {
    int n, logn;
    u64 *psi;      /* psi^j */
    u64 *psi_inv;  /* psi^-j / n */
    u64 *w;        /* w^j, j < n/2 */
    u64 *w_inv;
} ntt_plan;

#define MAX_PLANS 17
static ntt_plan g_plans[MAX_PLANS];

static const ntt_plan *get_plan(int n)
{
    int logn = 0;
    while ((1 << logn) < n) logn++;
    ntt_plan *p = &g_plans[logn];
    if (p->n == n) return p;
#pragma omp critical(orc_plan)
    {
        if (p->n != n) {
            u64 psi = orc_ff_root_of_unity(2 * (u64)n);
            u64 psi_i = ff_inv(psi), ninv = ff_inv((u64)n);
            u64 w = ff_mul(psi, psi), w_i = ff_inv(w);
            u64 *a = malloc(sizeof(u64) * n), *b = malloc(sizeof(u64) * n);
            u64 *c = malloc(sizeof(u64) * n), *d = malloc(sizeof(u64) * n);
            u64 x = 1, y = ninv, z = 1, t = 1;
            for (int j = 0; j < n; j++) {
                a[j] = x; b[j] = y;
                x = ff_mul(x, psi); y = ff_mul(y, psi_i);
                if (j < n / 2) { c[j] = z; d[j] = t; z = ff_mul(z, w); t = ff_mul(t, w_i); }
            }
            p->psi = a; p->psi_inv = b; p->w = c; p->w_inv = d; p->logn = logn;
            __sync_synchronize();
            p->n = n;
        }
    }
    return p;
}

static void bitrev_permute(u64 *x, int n, int logn)
{
    for (int i = 0; i < n; i++) {
        int j = 0;
        for (int b = 0; b < logn; b++) j |= ((i >> b) & 1) << (logn - 1 - b);
        if (j > i) { u64 t = x[i]; x[i] = x[j]; x[j] = t; }
    }
}

/* in-place cyclic NTT (DIT after bit reversal), ntt_cpu.py:145-185 */
static void cyclic_ntt(u64 *x, const ntt_plan *p, const u64 *w)
{
    int n = p->n, logn = p->logn;
    bitrev_permute(x, n, logn);
    for (int stage = 0; stage < logn; stage++) {
        int mmax = 1 << stage, istep = mmax * 2, tstep = n / istep;
        for (int m = 0; m < mmax; m++) {
            u64 tw = w[m * tstep];
            for (int i = m; i < n; i += istep) {
                int j = i + mmax;
                u64 t = ff_mul(x[j], tw);
                x[j] = ff_sub(x[i], t);
                x[i] = ff_add(x[i], t);
            }
        }
    }
}

static void ntt_forward_ff(u64 *x, const ntt_plan *p) /* ntt.py:43-44 */
{
    for (int j = 0; j < p->n; j++) x[j] = ff_mul(x[j], p->psi[j]);
    cyclic_ntt(x, p, p->w);
}

static void ntt_inverse_ff(u64 *x, const ntt_plan *p) /* ntt.py:36-41 */
{
    cyclic_ntt(x, p, p->w_inv);
    for (int j = 0; j < p->n; j++) x[j] = ff_mul(x[j], p->psi_inv[j]);
}

/* forward_transform_ref (i32_conversion=True), polynomial_transform_ntt.py:45-46 */
void orc_ntt_forward_i32(u64 *out, const i32 *in, long batch, int n)
{
    const ntt_plan *p = get_plan(n);
#pragma omp parallel for schedule(static)
    for (long b = 0; b < batch; b++) {
        u64 *o = out + b * n;
        for (int j = 0; j < n; j++) o[j] = ff_from_i32(in[b * n + j]);
        ntt_forward_ff(o, p);
    }
}

/* ntt_transform_ref(i32_conversion=False), ntt.py:30-44 */
void orc_ntt_forward_u64(u64 *out, const u64 *in, long batch, int n)
{
    const ntt_plan *p = get_plan(n);
#pragma omp parallel for schedule(static)
    for (long b = 0; b < batch; b++) {
        u64 *o = out + b * n;
        for (int j = 0; j < n; j++) o[j] = ff_mod(in[b * n + j]);
        ntt_forward_ff(o, p);
    }
}

/* inverse_transform_ref (i32_conversion=True), polynomial_transform_ntt.py:49-50 */
void orc_ntt_inverse_i32(i32 *out, const u64 *in, long batch, int n)
{
    const ntt_plan *p = get_plan(n);
#pragma omp parallel for schedule(static)
    for (long b = 0; b < batch; b++) {
        u64 *t = malloc(sizeof(u64) * n);
        for (int j = 0; j < n; j++) t[j] = ff_mod(in[b * n + j]);
        ntt_inverse_ff(t, p);
        for (int j = 0; j < n; j++) out[b * n + j] = ff_to_i32(t[j]);
        free(t);
    }
}

void orc_ntt_inverse_u64(u64 *out, const u64 *in, long batch, int n)
{
    const ntt_plan *p = get_plan(n);
#pragma omp parallel for schedule(static)
    for (long b = 0; b < batch; b++) {
        u64 *o = out + b * n;
        for (int j = 0; j < n; j++) o[j] = ff_mod(in[b * n + j]);
        ntt_inverse_ff(o, p);
    }
}

/* O(N^2) schoolbook negacyclic product mod 2^32 (test/test_transform/test_computation.py:71-124
 * poly_mul_ref): independent check of the transform-based product */
void orc_poly_mul_schoolbook(i32 *out, const i32 *a, const i32 *b, long batch, int n)
{
#pragma omp parallel for schedule(static)
    for (long t = 0; t < batch; t++) {
        const i32 *pa = a + t * n, *pb = b + t * n;
        u32 *o = (u32 *)(out + t * n);
        for (int k = 0; k < n; k++) o[k] = 0;
        for (int i = 0; i < n; i++) {
            u32 ai = (u32)pa[i];
            for (int j = 0; j < n; j++) {
                u32 pr = ai * (u32)pb[j];
                int k = i + j;
                if (k < n) o[k] += pr; else o[k - n] -= pr;
            }
        }
    }
}

/* ------------------------------------------------------------------ */
/* Mod-switch: Torus32ToPhaseReference, numeric_functions_cpu.py:23-37 */
/* ------------------------------------------------------------------ */
void orc_t32_to_phase(i32 *result, const i32 *phase, long n, u32 mspace_size)
{
    u32 interv = (u32)((1ULL << 32) / mspace_size);
    u32 half = interv / 2;
    for (long i = 0; i < n; i++) result[i] = (i32)(((u32)phase[i] + half) / interv);
}

/* ------------------------------------------------------------------ */
/* ShiftTorusPolynomialReference, polynomials_cpu.py:25-59            */
/* result/source: [batch][polys][N]; powers: [batch] (already         */
/* selected when powers_view)                                         */
/* ------------------------------------------------------------------ */
static void shift_one(i32 *res, const i32 *src, int N, int power, int minus_one)
{
    if (power < N) { /* polynomials_cpu.py:49-51 */
        for (int j = 0; j < power; j++) res[j] = (i32)(0u - (u32)src[j + N - power]);
        for (int j = power; j < N; j++) res[j] = src[j - power];
    } else {         /* polynomials_cpu.py:52-55 */
        power -= N;
        for (int j = 0; j < power; j++) res[j] = src[j + N - power];
        for (int j = power; j < N; j++) res[j] = (i32)(0u - (u32)src[j - power]);
    }
    if (minus_one) /* polynomials_cpu.py:57-58 */
        for (int j = 0; j < N; j++) res[j] = (i32)((u32)res[j] - (u32)src[j]);
}

void orc_shift_torus_polynomial(i32 *result, const i32 *source, const i32 *powers,
                                long batch, int polys, int N, int minus_one, int invert_powers)
{
    for (long b = 0; b < batch; b++) {
        int power = powers[b];
        if (invert_powers) power = 2 * N - power; /* polynomials_cpu.py:43-44 */
        for (int p = 0; p < polys; p++)
            shift_one(result + (b * polys + p) * N, source + (b * polys + p) * N, N, power, minus_one);
    }
}

/* ------------------------------------------------------------------ */
/* TLWE: trivial, extract.  tlwe_cpu.py:26-60                          */
/* ------------------------------------------------------------------ */
void orc_tlwe_noiseless_trivial(i32 *a, float *cv, const i32 *mu, long batch, int mask_size, int N)
{
    for (long b = 0; b < batch; b++) {
        memset(a + b * (mask_size + 1) * N, 0, sizeof(i32) * mask_size * N);   /* tlwe_cpu.py:34 */
        memcpy(a + (b * (mask_size + 1) + mask_size) * N, mu + b * N, sizeof(i32) * N); /* :35 */
        cv[b] = 0;                                                              /* :36 */
    }
}

void orc_tlwe_extract_lwe_samples(i32 *result_a, i32 *result_b, const i32 *tlwe_a,
                                  long batch, int mask_size, int N)
{
    for (long b = 0; b < batch; b++) {
        for (int m = 0; m < mask_size; m++) {
            const i32 *src = tlwe_a + (b * (mask_size + 1) + m) * N;
            i32 *dst = result_a + (b * mask_size + m) * N;
            dst[0] = src[0];                                                   /* tlwe_cpu.py:55 */
            for (int j = 1; j < N; j++) dst[j] = (i32)(0u - (u32)src[N - j]);  /* :56 */
        }
        result_b[b] = tlwe_a[(b * (mask_size + 1) + mask_size) * N];           /* :58 */
    }
}

/* ------------------------------------------------------------------ */
/* TGSW decomposition, tgsw_cpu.py:26-49; params from tgsw.py:43-57    */
/* sample [batch][k+1][N] -> result [batch][k+1][l][N]                 */
/* ------------------------------------------------------------------ */
static i32 tgsw_offset(int l, int log2_base) /* tgsw.py:49-52 */
{
    int64_t s = 0;
    for (int p = 1; p <= l; p++) s += (int64_t)1 << (32 - p * log2_base);
    s *= (1 << log2_base) / 2;
    return (i32)(u32)(u64)s;
}

void orc_tgsw_decomp(i32 *result, const i32 *sample, long batch, int mask_size, int l,
                     int log2_base, int N)
{
    i32 off = tgsw_offset(l, log2_base);
    i32 mask = (1 << log2_base) - 1, half = 1 << (log2_base - 1);
    for (long b = 0; b < batch; b++)
        for (int m = 0; m <= mask_size; m++)
            for (int p = 1; p <= l; p++) {
                const i32 *s = sample + (b * (mask_size + 1) + m) * N;
                i32 *r = result + ((b * (mask_size + 1) + m) * l + (p - 1)) * N;
                int sh = 32 - p * log2_base;
                for (int j = 0; j < N; j++) {
                    i32 x = (i32)((u32)s[j] + (u32)off);
                    r[j] = ((x >> sh) & mask) - half; /* arithmetic shift, tgsw_cpu.py:47 */
                }
            }
}

/* ------------------------------------------------------------------ */
/* MAC in transformed space, tgsw_cpu.py:52-79                         */
/* sample u64 [batch][k+1][l][N], bk u64 [n][k+1][l][k+1][N] (NTT,     */
/* Montgomery-prepared), result u64 [batch][k+1][N]                    */
/* ------------------------------------------------------------------ */
void orc_tlwe_transformed_add_mul(u64 *result, const u64 *sample, const u64 *bk, int bk_row,
                                  long batch, int mask_size, int l, int N)
{
    int k1 = mask_size + 1;
#pragma omp parallel for schedule(static)
    for (long b = 0; b < batch; b++) {
        u64 *r = result + b * k1 * N;
        for (int i = 0; i < k1 * N; i++) r[i] = 0;                        /* tgsw_cpu.py:68 */
        for (int mi = 0; mi < k1; mi++)
            for (int d = 0; d < l; d++) {                                 /* :69-70 */
                const u64 *s = sample + ((b * k1 + mi) * l + d) * N;
                const u64 *row = bk + (((long)bk_row * k1 + mi) * l + d) * k1 * N;
                for (int mo = 0; mo < k1; mo++)
                    for (int f = 0; f < N; f++)
                        r[mo * N + f] = ff_add(r[mo * N + f],
                                               ff_mul(ff_mul(s[f], row[mo * N + f]), FF_RINV)); /* :71-77 */
            }
    }
}

/* ------------------------------------------------------------------ */
/* External product, TGswTransformedExternalMulReference               */
/* tgsw_cpu.py:82-106: decomp -> forward -> MAC -> inverse             */
/* accum i32 [batch][k+1][N] in/out                                    */
/* ------------------------------------------------------------------ */
static void external_mul_one(i32 *accum, const u64 *bk, int bk_row, int mask_size, int l,
                             int log2_base, int N, i32 *dec, u64 *tr, u64 *acc_tr)
{
    int k1 = mask_size + 1;
    const ntt_plan *p = get_plan(N);
    orc_tgsw_decomp(dec, accum, 1, mask_size, l, log2_base, N);            /* tgsw_cpu.py:101 */
    for (int t = 0; t < k1 * l; t++) {                                     /* :102 */
        for (int j = 0; j < N; j++) tr[t * N + j] = ff_from_i32(dec[t * N + j]);
        ntt_forward_ff(tr + t * N, p);
    }
    for (int i = 0; i < k1 * N; i++) acc_tr[i] = 0;                        /* :103 */
    for (int mi = 0; mi < k1; mi++)
        for (int d = 0; d < l; d++) {
            const u64 *s = tr + (mi * l + d) * N;
            const u64 *row = bk + (((long)bk_row * k1 + mi) * l + d) * k1 * N;
            for (int mo = 0; mo < k1; mo++)
                for (int f = 0; f < N; f++)
                    acc_tr[mo * N + f] = ff_add(acc_tr[mo * N + f],
                                                ff_mul(ff_mul(s[f], row[mo * N + f]), FF_RINV));
        }
    for (int mo = 0; mo < k1; mo++) {                                      /* :104 */
        ntt_inverse_ff(acc_tr + mo * N, p);
        for (int j = 0; j < N; j++) accum[mo * N + j] = ff_to_i32(acc_tr[mo * N + j]);
    }
}

void orc_tgsw_external_mul(i32 *accum, const u64 *bk, int bk_row, long batch, int mask_size,
                           int l, int log2_base, int N)
{
    int k1 = mask_size + 1;
#pragma omp parallel
    {
        i32 *dec = malloc(sizeof(i32) * k1 * l * N);
        u64 *tr = malloc(sizeof(u64) * k1 * l * N);
        u64 *acc_tr = malloc(sizeof(u64) * k1 * N);
#pragma omp for schedule(static)
        for (long b = 0; b < batch; b++)
            external_mul_one(accum + b * k1 * N, bk, bk_row, mask_size, l, log2_base, N, dec, tr, acc_tr);
        free(dec); free(tr); free(acc_tr);
    }
}

/* ------------------------------------------------------------------ */
/* blind_rotate (multi-kernel driver), bootstrap.py:96-142:            */
/* for i<n: ACC = BK_i (*) ((X^bara_i - 1) ACC) + ACC                  */
/* accum i32 [batch][k+1][N] in/out; bara i32 [batch][bara_stride]     */
/* ------------------------------------------------------------------ */
void orc_blind_rotate(i32 *accum, const u64 *bk, const i32 *bara, long bara_stride, int n_iter,
                      long batch, int mask_size, int l, int log2_base, int N)
{
    int k1 = mask_size + 1;
#pragma omp parallel
    {
        i32 *dec = malloc(sizeof(i32) * k1 * l * N);
        u64 *tr = malloc(sizeof(u64) * k1 * l * N);
        u64 *acc_tr = malloc(sizeof(u64) * k1 * N);
        i32 *tmp = malloc(sizeof(i32) * k1 * N);
#pragma omp for schedule(dynamic, 1)
        for (long b = 0; b < batch; b++) {
            i32 *acc = accum + b * k1 * N;
            for (int i = 0; i < n_iter; i++) {
                int power = bara[b * bara_stride + i];
                for (int m = 0; m < k1; m++)                  /* bootstrap.py:103, tlwe.py:168-169 */
                    shift_one(tmp + m * N, acc + m * N, N, power, 1);
                external_mul_one(tmp, bk, i, mask_size, l, log2_base, N, dec, tr, acc_tr); /* :106 */
                for (int j = 0; j < k1 * N; j++)              /* :109, tlwe.py:173-175 */
                    acc[j] = (i32)((u32)acc[j] + (u32)tmp[j]);
            }
        }
        free(dec); free(tr); free(acc_tr); free(tmp);
    }
}

/* ------------------------------------------------------------------ */
/* LWE keyswitch, LweKeyswitchReference lwe_cpu.py:62-93               */
/* ks_a i32 [input_size][t][base][output_size], ks_b/ks_cv [..][base]  */
/* source_a [batch][input_size], source_b [batch]                      */
/* ------------------------------------------------------------------ */
void orc_lwe_keyswitch(i32 *result_a, i32 *result_b, float *result_cv,
                       const i32 *ks_a, const i32 *ks_b, const float *ks_cv,
                       const i32 *source_a, const i32 *source_b,
                       long batch, int input_size, int output_size, int t, int log2_base)
{
    int base = 1 << log2_base;
    u32 prec_offset = 1u << (32 - (1 + log2_base * t)); /* lwe_cpu.py:70 */
    u32 mask = base - 1;
#pragma omp parallel for schedule(static)
    for (long b = 0; b < batch; b++) {
        u32 *ra = (u32 *)(result_a + b * output_size);
        u32 rb = (u32)source_b[b];                        /* lwe_cpu.py:81 */
        float rcv = 0;                                    /* :82 */
        for (int i = 0; i < output_size; i++) ra[i] = 0;  /* :80 */
        for (int l = 0; l < input_size; l++) {
            i32 ai = (i32)((u32)source_a[b * input_size + l] + prec_offset);
            for (int j = 0; j < t; j++) {
                u32 x = ((u32)(ai >> (32 - (j + 1) * log2_base))) & mask; /* :76 */
                long row = ((long)l * t + j) * base + x;
                const i32 *ka = ks_a + row * output_size;
                for (int i = 0; i < output_size; i++) ra[i] -= (u32)ka[i]; /* :86-92 */
                rb -= (u32)ks_b[row];
                rcv += ks_cv[row];
            }
        }
        result_b[b] = (i32)rb;
        result_cv[b] = rcv;
    }
}

/* LweLinearReference, lwe_cpu.py:115-123 (no broadcasting here; the Python side expands) */
void orc_lwe_linear(i32 *ra, i32 *rb, float *rcv, const i32 *sa, const i32 *sb, const float *scv,
                    i32 p, int add_result, long batch, int n)
{
    for (long b = 0; b < batch; b++) {
        for (int i = 0; i < n; i++) {
            u32 v = (u32)p * (u32)sa[b * n + i];
            ra[b * n + i] = (i32)((add_result ? (u32)ra[b * n + i] : 0u) + v);
        }
        rb[b] = (i32)((add_result ? (u32)rb[b] : 0u) + (u32)p * (u32)sb[b]);
        rcv[b] = (add_result ? rcv[b] : 0.0f) + (float)((int64_t)p * p) * scv[b];
    }
}

/* LweNoiselessTrivialConstantReference, lwe_cpu.py:136-143 */
void orc_lwe_trivial_const(i32 *ra, i32 *rb, float *rcv, i32 mu, long batch, int n)
{
    for (long b = 0; b < batch; b++) {
        for (int i = 0; i < n; i++) ra[b * n + i] = 0;
        rb[b] = mu;
        rcv[b] = 0;
    }
}

/* ------------------------------------------------------------------ */
/* bootstrap, bootstrap.py:206-229 + blind_rotate_and_extract :154-196 */
/* x (a [batch][n], b [batch]) -> extracted LWE (a [batch][k*N], b)    */
/* ------------------------------------------------------------------ */
void orc_bootstrap_extract(i32 *ext_a, i32 *ext_b, const u64 *bk, const i32 *xa, const i32 *xb,
                           i32 mu, long batch, int n, int mask_size, int l, int log2_base, int N)
{
    int k1 = mask_size + 1;
    i32 *bara = malloc(sizeof(i32) * batch * n);
    i32 *barb = malloc(sizeof(i32) * batch);
    i32 *acc = malloc(sizeof(i32) * batch * k1 * N);
    i32 *tv = malloc(sizeof(i32) * N), *tvb = malloc(sizeof(i32) * N);
    float cv;
    orc_t32_to_phase(barb, xb, batch, 2 * N);                      /* bootstrap.py:220 */
    orc_t32_to_phase(bara, xa, batch * n, 2 * N);                  /* :221 */
    for (int j = 0; j < N; j++) tv[j] = mu;                        /* :224 */
    for (long b = 0; b < batch; b++) {
        orc_shift_torus_polynomial(tvb, tv, barb + b, 1, 1, N, 0, 1);  /* :177-178 */
        orc_tlwe_noiseless_trivial(acc + b * k1 * N, &cv, tvb, 1, mask_size, N); /* :181-182 */
    }
    orc_blind_rotate(acc, bk, bara, n, n, batch, mask_size, l, log2_base, N); /* :189-190 */
    orc_tlwe_extract_lwe_samples(ext_a, ext_b, acc, batch, mask_size, N);     /* :193 */
    free(bara); free(barb); free(acc); free(tv); free(tvb);
}

/* Full binary gate: (0,c) + pa*a + pb*b -> bootstrap(mu) -> keyswitch.
 * gates.py:81-121 (NAND: c=+2^29, pa=pb=-1) and siblings :124-597.    */
void orc_gate_binary(i32 *ra, i32 *rb, float *rcv,
                     const u64 *bk, const i32 *ks_a, const i32 *ks_b, const float *ks_cv,
                     const i32 *a_a, const i32 *a_b, const i32 *b_a, const i32 *b_b,
                     i32 c, i32 pa, i32 pb, i32 mu,
                     long batch, int n, int mask_size, int l, int log2_base, int N,
                     int ks_t, int ks_log2_base)
{
    long ext = (long)mask_size * N;
    i32 *ta = malloc(sizeof(i32) * batch * n), *tb = malloc(sizeof(i32) * batch);
    i32 *ea = malloc(sizeof(i32) * batch * ext), *eb = malloc(sizeof(i32) * batch);
    for (long b = 0; b < batch; b++) {
        for (int i = 0; i < n; i++)
            ta[b * n + i] = (i32)((u32)pa * (u32)a_a[b * n + i] + (u32)pb * (u32)b_a[b * n + i]);
        tb[b] = (i32)((u32)c + (u32)pa * (u32)a_b[b] + (u32)pb * (u32)b_b[b]);
    }
    orc_bootstrap_extract(ea, eb, bk, ta, tb, mu, batch, n, mask_size, l, log2_base, N);
    orc_lwe_keyswitch(ra, rb, rcv, ks_a, ks_b, ks_cv, ea, eb, batch, (int)ext, n, ks_t, ks_log2_base);
    free(ta); free(tb); free(ea); free(eb);
}

/* MUX gate, gates.py:600-664 */
void orc_gate_mux(i32 *ra, i32 *rb, float *rcv,
                  const u64 *bk, const i32 *ks_a, const i32 *ks_b, const float *ks_cv,
                  const i32 *a_a, const i32 *a_b, const i32 *b_a, const i32 *b_b,
                  const i32 *c_a, const i32 *c_b,
                  long batch, int n, int mask_size, int l, int log2_base, int N,
                  int ks_t, int ks_log2_base)
{
    const u32 MU = 1u << 29;             /* phase_to_t32(1, 8) */
    const u32 AND_CONST = 0u - (1u << 29); /* phase_to_t32(-1, 8), gates.py:638 */
    long ext = (long)mask_size * N;
    i32 *ta = malloc(sizeof(i32) * batch * n), *tb = malloc(sizeof(i32) * batch);
    i32 *u1a = malloc(sizeof(i32) * batch * ext), *u1b = malloc(sizeof(i32) * batch);
    i32 *u2a = malloc(sizeof(i32) * batch * ext), *u2b = malloc(sizeof(i32) * batch);
    for (long b = 0; b < batch; b++) {   /* (0,-1/8) + a + b, gates.py:639-641 */
        for (int i = 0; i < n; i++) ta[b * n + i] = (i32)((u32)a_a[b * n + i] + (u32)b_a[b * n + i]);
        tb[b] = (i32)(AND_CONST + (u32)a_b[b] + (u32)b_b[b]);
    }
    orc_bootstrap_extract(u1a, u1b, bk, ta, tb, (i32)MU, batch, n, mask_size, l, log2_base, N);
    for (long b = 0; b < batch; b++) {   /* (0,-1/8) - a + c, gates.py:648-650 */
        for (int i = 0; i < n; i++) ta[b * n + i] = (i32)((u32)c_a[b * n + i] - (u32)a_a[b * n + i]);
        tb[b] = (i32)(AND_CONST - (u32)a_b[b] + (u32)c_b[b]);
    }
    orc_bootstrap_extract(u2a, u2b, bk, ta, tb, (i32)MU, batch, n, mask_size, l, log2_base, N);
    for (long b = 0; b < batch; b++) {   /* (0,1/8) + u1 + u2, gates.py:657-661 */
        for (long i = 0; i < ext; i++) u1a[b * ext + i] = (i32)((u32)u1a[b * ext + i] + (u32)u2a[b * ext + i]);
        u1b[b] = (i32)(MU + (u32)u1b[b] + (u32)u2b[b]);
    }
    orc_lwe_keyswitch(ra, rb, rcv, ks_a, ks_b, ks_cv, u1a, u1b, batch, (int)ext, n, ks_t, ks_log2_base);
    free(ta); free(tb); free(u1a); free(u1b); free(u2a); free(u2b);
}

/* ------------------------------------------------------------------ */
/* Key generation / client side (needed so the oracle and the GPU see  */
/* the same inputs; not on the hot path)                               */
/* ------------------------------------------------------------------ */

/* TLweEncryptZeroReference, tlwe_cpu.py:64-89
 * result_a [batch][k+1][N]; key i32 [k][N]; noises1 [batch][k][N]; noises2 [batch][N] */
void orc_tlwe_encrypt_zero(i32 *result_a, float *result_cv, const i32 *key, const i32 *noises1,
                           const i32 *noises2, double noise, long batch, int mask_size, int N)
{
    const ntt_plan *p = get_plan(N);
    int k1 = mask_size + 1;
    u64 *tkey = malloc(sizeof(u64) * mask_size * N);
    for (int m = 0; m < mask_size; m++) {
        for (int j = 0; j < N; j++) tkey[m * N + j] = ff_from_i32(key[m * N + j]);
        ntt_forward_ff(tkey + m * N, p);                                   /* tlwe_cpu.py:76 */
    }
#pragma omp parallel
    {
        u64 *t = malloc(sizeof(u64) * N);
#pragma omp for schedule(static)
        for (long b = 0; b < batch; b++) {
            i32 *r = result_a + b * k1 * N;
            for (int j = 0; j < N; j++) r[mask_size * N + j] = noises2[b * N + j]; /* :82 */
            for (int m = 0; m < mask_size; m++) {
                const i32 *a = noises1 + (b * mask_size + m) * N;
                for (int j = 0; j < N; j++) { r[m * N + j] = a[j]; t[j] = ff_from_i32(a[j]); } /* :81 */
                ntt_forward_ff(t, p);                                      /* :77 */
                for (int j = 0; j < N; j++) t[j] = ff_mul(t[j], tkey[m * N + j]); /* :78 */
                ntt_inverse_ff(t, p);                                      /* :79 */
                for (int j = 0; j < N; j++)                                /* :83-84 */
                    r[mask_size * N + j] = (i32)((u32)r[mask_size * N + j] + (u32)ff_to_i32(t[j]));
            }
            result_cv[b] = (float)(noise * noise);                         /* :86 */
        }
        free(t);
    }
    free(tkey);
}

/* TGswAddMessageReference, tgsw_cpu.py:109-126: result_a [n][k+1][l][k+1][N] */
void orc_tgsw_add_message(i32 *result_a, const i32 *messages, long count, int mask_size, int l,
                          int log2_base, int N)
{
    int k1 = mask_size + 1;
    for (long i = 0; i < count; i++)
        for (int m = 0; m < k1; m++)
            for (int d = 0; d < l; d++) {
                u32 bp = 1u << (32 - (d + 1) * log2_base);                 /* tgsw.py:47 */
                i32 *x = result_a + ((((i * k1 + m) * l + d) * k1) + m) * N;
                x[0] = (i32)((u32)x[0] + (u32)messages[i] * bp);           /* tgsw_cpu.py:121-124 */
            }
}

/* TLweTransformSamples: forward NTT + Montgomery prepare, tlwe_gpu.py:199-236 */
void orc_tlwe_transform_samples(u64 *out, const i32 *in, long polys, int N)
{
    orc_ntt_forward_i32(out, in, polys, N);
    orc_ff_prepare_for_mul(out, out, polys * N);
}

/* MakeLweKeyswitchKeyReference, lwe_cpu.py:27-59
 * ks_a [in][t][base][out]; noises_a [in][t][base-1][out]; noises_b [in][t][base-1] */
void orc_make_lwe_keyswitch_key(i32 *ks_a, i32 *ks_b, float *ks_cv, const i32 *in_key,
                                const i32 *out_key, const i32 *noises_a, const i32 *noises_b,
                                double noise, int input_size, int output_size, int t, int log2_base)
{
    int base = 1 << log2_base;
    for (int l = 0; l < input_size; l++)
        for (int j = 0; j < t; j++) {
            long r0 = ((long)l * t + j) * base;
            memset(ks_a + r0 * output_size, 0, sizeof(i32) * output_size); /* lwe_cpu.py:31-33 */
            ks_b[r0] = 0; ks_cv[r0] = 0;
            for (int h = 1; h < base; h++) {
                long nr = ((long)l * t + j) * (base - 1) + (h - 1);
                const i32 *na = noises_a + nr * output_size;
                u32 msg = (u32)in_key[l] * (u32)h * (1u << (32 - (j + 1) * log2_base)); /* :54 */
                u32 dot = 0;
                for (int i = 0; i < output_size; i++) {
                    ks_a[(r0 + h) * output_size + i] = na[i];              /* :35 */
                    dot += (u32)na[i] * (u32)out_key[i];                   /* :24, :36 */
                }
                ks_b[r0 + h] = (i32)(msg + (u32)noises_b[nr] + dot);       /* :36 */
                ks_cv[r0 + h] = (float)(noise * noise);                             /* :37 */
            }
        }
}

/* LweEncryptReference, lwe_cpu.py:96-104 */
void orc_lwe_encrypt(i32 *ra, i32 *rb, float *rcv, const i32 *messages, const i32 *key,
                     const i32 *noises_a, const i32 *noises_b, double noise, long batch, int n)
{
    for (long b = 0; b < batch; b++) {
        u32 dot = 0;
        for (int i = 0; i < n; i++) {
            ra[b * n + i] = noises_a[b * n + i];
            dot += (u32)noises_a[b * n + i] * (u32)key[i];
        }
        rb[b] = (i32)((u32)noises_b[b] + (u32)messages[b] + dot);
        rcv[b] = (float)(noise * noise);
    }
}

/* LweDecryptReference, lwe_cpu.py:107-112 */
void orc_lwe_decrypt(i32 *result, const i32 *la, const i32 *lb, const i32 *key, long batch, int n)
{
    for (long b = 0; b < batch; b++) {
        u32 dot = 0;
        for (int i = 0; i < n; i++) dot += (u32)la[b * n + i] * (u32)key[i];
        result[b] = (i32)((u32)lb[b] - dot);
    }
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
