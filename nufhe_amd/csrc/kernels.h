// kernels.h -- launch descriptors and host launchers of kernels.hip (internal to libnufhe_hip.so)
#pragma once
#include <hip/hip_runtime.h>

#include "blind_rotate.h"
#include "blind_rotate_fft.h"

#define BR_BLOCK_THREADS 512
#define BR_WAVES_PER_BLOCK 8
#define KS_BLOCK_THREADS 256
#define KS_TILE_BITS 64

struct BrJob {
    BrSource s0, s1;
    i32 c0;
};

// Fused bootstrap launch: global bit g in [0, nbits_total); job = g / bits_per_job (at most 2 jobs:
// MUX runs its two blind rotations in one launch), bit = g % bits_per_job.
#define CLOCK_PROBE_WORDS (2 + 3 * 8)

struct BrLaunch {
    BrJob job[2];
    long bits_per_job;
    long nbits_total;
    const void *bk;       // wave layout: NTT u64 [n][8][1024] / FFT complex128 [n][8][512]
    int n;
    i32 mu;
    i32 *out_a;           // extracted LWE(1024): out_a[g * out_a_stride + j]
    i32 *out_b;           // out_b[g * out_b_stride]
    long out_a_stride;
    long out_b_stride;
    const void *tw_a;     // NTT: tw1f u64[1024]; FFT: tw1 complex[512]
    const void *tw_b;     // NTT: tw1i u64[1024]; FFT: tw2 complex[64]
    // half-ring team kernel (k_bootstrap_team8): the key in the half-ring layout and the table block of ntt512_half.h;
    // bk_half == nullptr keeps the smallest batches on the 4-wave team kernel
    const void *bk_half;
    const void *tw_half;
    // profiling only (nufhe_profile_enable), else nullptr: wave 0 of work-group 0 of the wave-per-bit kernels stores
    // how long it lived, {shader-clock ticks (s_memtime), constant 100 MHz ticks (s_memrealtime)}
    unsigned long long *clock_probe;     // nullptr or CLOCK_PROBE_WORDS words (ClockProbe, kernels.hip)
    // exact-FFT engine (transform 2, kernels_xfft.hip): global parking space of the accumulators, 2048 words per bit
    u32 *park;
};

struct KsLaunch {
    u32 *acc;             // [nbits][n] zero-initialised accumulator
    const i32 *ks_a3;     // [input_size][8][3][n] (digits 1..3)
    const i32 *src1_a;    // LWE(1024) source, plus optional second source added on the fly (MUX)
    const i32 *src2_a;
    long src1_stride, src2_stride;
    long nbits;
    int n;
    int j_per_block;
    int input_size;       // mask_size * 1024
    // optional: the key as signed byte planes in MFMA operand order (k_ks_planes); non-null selects k_keyswitch_mfma
    const signed char *ks_planes;
    // scratch of k_keyswitch_mfma: the digits of the source, transposed: u16 [input_size][nbits rounded up to 64]
    unsigned short *digits_t;
};

// ---- heterogeneous gate batch (nufhe_gate_batch): device tables, ascending `start` --------------------------------
// One blind rotation per row of a combined LWE(n) scratch: rows [start, start + nbits) of group g hold
// (0, c0) + p0 * src0[bit] + p1 * src1[bit] (int32 wraparound), bit = row - start.
struct BatchRot {
    const i32 *a0, *b0, *a1, *b1;
    long a0_stride, b0_stride, a1_stride, b1_stride;
    long start, nbits;
    i32 p0, p1, c0, pad_;
};
// One result slice per gate of the batch: output bits [start, start + nbits) go to its own view; `second` >= 0 (MUX):
// the rows of the second rotation's extracted samples, added to the first rotation's (with +mu on b) before the keyswitch.
struct BatchOut {
    i32 *a, *b;
    float *cv;
    long a_stride, b_stride;
    long start, nbits;
    long second;
};

static_assert(sizeof(BatchRot) % 16 == 0 && sizeof(BatchOut) % 16 == 0, "the job tables are written in 16-byte words");

struct KsFinal {
    // heterogeneous batch: results go to the views of `batch_outs` (n_batch_outs entries) instead of out_a / out_b / out_cv
    const BatchOut *batch_outs;
    int n_batch_outs;
    const u32 *acc;
    const i32 *ks_b;      // [1024][8][4]
    const float *ks_cv;   // [1024][8][4]
    const i32 *src1_a, *src2_a, *src1_b, *src2_b;
    long src1_stride, src2_stride, src1_bstride, src2_bstride;
    i32 c0;               // constant added to b (MUX: +1/8)
    i32 *out_a, *out_b;
    float *out_cv;
    long out_a_stride, out_b_stride;
    int n;
    int input_size;       // mask_size * 1024
    // keys whose non-zero variance entries all hold ONE value v (every generated key: lwe_cpu.py:37):
    // cv_table[c] = v added c times in float32, sequentially; the variance of a result is then
    // cv_table[number of non-zero digits] -- the value the reference's sequential sum reaches, without
    // the 8192-term serial addition.  NULL: general keys, sequential sum.
    const float *cv_table;
};

struct LweView {
    i32 *a;
    i32 *b;
    float *cv;
    long a_stride, b_stride;
};

// Batch-size switches of launch_bootstrap / the keyswitch, in ciphertext bits (rotations of one launch).  They come from
// the DEVICE, not from the box the kernels were tuned on: br_tuning_for() looks the part up in a table of measured
// switch points keyed by (gcnArchName, CU count) and, for a part that was never measured, scales the ratios of the
// measured one by its CU count (`measured` = 0 then).  The context holds one (nufhe_ctx_get_tuning / _set_tuning).
struct BrTuning {
    long team_max_bits;        // NTT: batches up to here run 8 (k = 1, half rings) / 4 / 3 (k = 2) waves per bit
    long team_max_bits_fft;    // FFT: the 4 / 3-wave kernels up to here (k = 1: only when the pair kernel is off)
    long pair_max_bits_ntt;    // NTT k = 1: above the team limit and up to here 2 waves per bit; 0 = off
    long pair_max_bits_fft;    // FFT k = 1: up to here 2 waves per bit; 0 = off
    int ring_k2;               // k = 2: 3 waves per bit without a partial-sum buffer above the team limit; 0 = wave kernels
    int k2_roomy_ratio_pct;    // NTT k = 2 wave kernels: time of a 6 x CUs round over a 4 x CUs round, in percent
    long ks_mfma_min_bits;     // keyswitch on the matrix cores for batches ABOVE this many bits
    int measured;              // 1: this (architecture, CU count) has an entry in the table
    int num_cus;
};
BrTuning br_tuning_for(const char *arch_name, int num_cus);

hipError_t kernels_init_device(int *num_cus, char *arch_name, size_t arch_len);
// transform: 0 = NTT, 1 = FFT, 2 = exact FFT on a split key (NTT parameters, P.bk = the split image, P.park set)
#define BR_TRANSFORM_XFFT 2
hipError_t launch_bootstrap(const BrLaunch &P, int transform, int mask_size, const BrTuning &T, hipStream_t stream);
hipError_t launch_ff_op(u64 *out, const u64 *a, const u64 *b, const u64 *c, const u64 *d, const u64 *e, int op,
                        int shift, long count, hipStream_t stream);
hipError_t launch_ks_make(i32 *ks_b, float *ks_cv, const i32 *noises_a, const i32 *noises_b, const i32 *in_key,
                          const i32 *out_key, float variance, long rows, int n, hipStream_t stream);
hipError_t launch_ks_to_reference(i32 *out_a, const i32 *ks_a3, long groups, int n, hipStream_t stream);
hipError_t launch_tgsw_add_message(i32 *tgsw, const i32 *messages, long count, int mask_size, hipStream_t stream);
hipError_t launch_l4_op(u32 *out, u32 *out2, const u32 *a, const u32 *b, const u32 *c, int op, int shift, long count,
                        hipStream_t stream);
hipError_t launch_blind_rotate_accum(i32 *accum, const void *bk, const i32 *bara, long bara_stride, int row0,
                                     int n_rows, int external_mul_only, long batch, const void *tw_a,
                                     const void *tw_b, int transform, int mask_size, hipStream_t stream);
hipError_t launch_fft_forward(cplx *out, const i32 *in, long batch, const cplx *tw1, const cplx *tw2,
                              hipStream_t stream);
hipError_t launch_fft_inverse(i32 *out, const cplx *in, long batch, const cplx *tw1, const cplx *tw2,
                              hipStream_t stream);
hipError_t launch_bkf_from_coeffs(cplx *out, const i32 *in, long polys, const cplx *tw1, const cplx *tw2,
                                  hipStream_t stream);
// exact-FFT engine (kernels_xfft.hip)
hipError_t xfft_init();
// quad_max_bits: batches up to here run four waves per bit (k = 1)
hipError_t launch_bootstrap_xfft(const BrLaunch &P, int mask_size, int num_cus, long quad_max_bits, hipStream_t stream);
hipError_t launch_blind_rotate_accum_xfft(i32 *accum, const cplx *bkx, const i32 *bara, long bara_stride, int row0, int n_rows,
                                          int external_mul_only, long batch, const cplx *tw1, const cplx *tw2, u32 *park,
                                          int mask_size, hipStream_t stream);
// FFT keys, up to 1 x CUs bits: four waves per bit (kernels_xfft.hip, brfq_*)
hipError_t launch_bootstrap_fft_quad(const BrLaunch &P, hipStream_t stream);
hipError_t launch_bootstrap_fft_hex_k2(const BrLaunch &P, hipStream_t stream);      // tlwe_mask_size = 2: six waves per bit
// int32 TGSW polynomials [polys][1024] -> split key image complex128 [polys][2][8][64] (blind_rotate_xfft.h)
hipError_t launch_bkx_from_coeffs(cplx *out, const i32 *in, long polys, const cplx *tw1, const cplx *tw2, hipStream_t stream);
// NTT key in the wave layout -> its int32 coefficients [polys][1024]
// (*not_int32, a zeroed device word, is set when some coefficient is not a centred 32-bit integer)
hipError_t launch_bk_to_coeffs(i32 *out, const u64 *bk_wave, long polys, const u64 *tw1f, const u64 *tw1i, int *not_int32,
                               hipStream_t stream);
hipError_t launch_bkf_permute(cplx *out, const cplx *in, long polys, int to_reference, hipStream_t stream);
hipError_t launch_ntt_forward(u64 *out, const void *in, int mode, long batch, const u64 *tw1f, const u64 *tw1i,
                              hipStream_t stream);
hipError_t launch_ntt_inverse(void *out, const u64 *in, int mode, long batch, const u64 *tw1f, const u64 *tw1i,
                              hipStream_t stream);
hipError_t launch_poly_mul(i32 *out, const i32 *x, const i32 *y, const i32 *base, long out_stride, long batch,
                           long y_batch, const u64 *tw1f, const u64 *tw1i, hipStream_t stream);
hipError_t launch_poly_mul_strided(i32 *out, long out_stride, const i32 *x, long x_stride, const i32 *y,
                                   const i32 *base, long base_stride, long batch, const u64 *tw1f, const u64 *tw1i,
                                   hipStream_t stream);
hipError_t launch_bk_from_reference(u64 *out, const u64 *in, long polys, hipStream_t stream);
hipError_t launch_bk_to_reference(u64 *out, const u64 *in, long polys, hipStream_t stream);
hipError_t launch_bk_to_half(u64 *out, const u64 *in, long polys, hipStream_t stream);
hipError_t launch_bk_from_coeffs(u64 *out, const i32 *in, long polys, const u64 *tw1f, const u64 *tw1i,
                                 hipStream_t stream);
// byte planes of the keyswitch key for k_keyswitch_mfma: planes[4][input_size][2][KSM_COLS][16]
#define KSM_COLS 512
static inline size_t ks_planes_bytes(int input_size) { return (size_t)4 * input_size * 2 * KSM_COLS * 16; }
hipError_t launch_ks_planes(signed char *planes, const i32 *ks_a3, int input_size, int n, hipStream_t stream);
hipError_t launch_keyswitch(const KsLaunch &P, const KsFinal &F, hipStream_t stream);
// `bytes` bytes of host memory into device memory at d_dst THROUGH KERNEL ARGUMENTS (chunks of TABLE_CHUNK_BYTES per launch):
// the host buffer may be freed as soon as the call returns, and under stream capture the bytes live in the graph's nodes
#define TABLE_CHUNK_BYTES 3584
hipError_t launch_write_table(void *d_dst, const void *h_src, size_t bytes, hipStream_t stream);
hipError_t launch_batch_combine(i32 *out_a, i32 *out_b, const BatchRot *rots, int n_rots, long rows, int n, hipStream_t stream);
hipError_t launch_batch_mux_fold(i32 *ext_a, i32 *ext_b, const BatchOut *outs, int n_outs, long out_bits, int ext, i32 mu,
                                 hipStream_t stream);
hipError_t launch_lwe_linear(const LweView &res, const LweView &src, i32 p, int add_result, long nbits, int size,
                             hipStream_t stream);
hipError_t launch_lwe_trivial_const(const LweView &res, i32 mu, long nbits, int size, hipStream_t stream);
hipError_t launch_lwe_phase(i32 *out, long out_stride, const i32 *a, long a_stride, const i32 *base, long base_stride,
                            const i32 *key, i32 sign, long count, int n, hipStream_t stream);
hipError_t launch_t32_to_phase(i32 *result, const i32 *phase, long count, u32 mspace, hipStream_t stream);
hipError_t launch_shift_tp(i32 *result, const i32 *source, const i32 *powers, long powers_stride, long powers_idx,
                           long batch, int polys, int minus_one, int invert_powers, hipStream_t stream);
hipError_t launch_tgsw_decompose(i32 *result, const i32 *sample, long polys, hipStream_t stream);
hipError_t launch_tgsw_mac(u64 *result, const u64 *sample, const u64 *bk, int bk_row, long batch, int mask_size,
                           hipStream_t stream);
hipError_t launch_tlwe_extract(i32 *ra, i32 *rb, const i32 *tlwe, long batch, int mask_size, hipStream_t stream);
