/*
 * nufhe_hip.h -- C ABI of libnufhe_hip.so, the MI355X (gfx950) implementation of nufhe's
 * bootstrapped-gate hot path.  Called from Python through ctypes (nufhe_amd/_lib.py); every
 * entry point replaces one Reikna `Computation.__call__` (or a fused group of them) of the
 * reference (nucypher/nufhe).  Reference citations are relative to the reference repo root.
 *
 * Conventions
 *  - extern "C", POD arguments only, no exceptions cross the boundary: every entry point catches what its host side may
 *    throw (std::bad_alloc -> NUFHE_ENOMEM, anything else -> NUFHE_EHIP).
 *  - every function returns 0 on success, a negative NUFHE_E* code on failure; the message is
 *    available (per thread) from nufhe_last_error().
 *  - pointers named d_* are DEVICE pointers owned by the caller (e.g. torch tensors' data_ptr);
 *    pointers named h_* are HOST pointers.  Strides are in ELEMENTS.
 *  - all work is enqueued on the context's HIP stream; nothing synchronises unless stated.
 *  - scheme parameters are the reference's defaults (nufhe/api_low_level.py:49-61):
 *    N = 1024, k = 1 (tlwe_mask_size), l = 2, Bg = 2^10, keyswitch t = 8, base 4; the LWE size n
 *    (500) is a run-time argument, n <= 512.
 */
#ifndef NUFHE_HIP_H
#define NUFHE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NUFHE_OK 0
#define NUFHE_EINVAL (-1)   /* bad argument */
#define NUFHE_EHIP (-2)     /* HIP runtime error */
#define NUFHE_ENODEV (-3)   /* no usable GPU */
#define NUFHE_ENOKEY (-4)   /* key part not uploaded */
#define NUFHE_ENOMEM (-5)   /* host allocation failed (std::bad_alloc caught at the boundary) */

#define NUFHE_TRANSFORM_NTT 0   /* negacyclic NTT-1024 over 2^64 - 2^32 + 1 (bit-exact path) */
#define NUFHE_TRANSFORM_FFT 1   /* fp64 folded FFT-512 (polynomial_transform_fft.py), tolerance path */

#define NUFHE_N 1024        /* TLWE polynomial degree */
#define NUFHE_KS_T 8        /* keyswitch decomposition length */
#define NUFHE_KS_BASE 4     /* keyswitch base */

typedef struct nufhe_ctx nufhe_ctx;            /* one GPU + one stream; replaces the Reikna Thread */
typedef struct nufhe_cloudkey nufhe_cloudkey;  /* device copies of BootstrapKey + LweKeyswitchKey */

/* Bumped whenever a struct passed by value or an entry point's signature changes (3: nufhe_lwe gained `size`;
 * 4: output-stride checks, key images; 5: key images carry a header, nufhe_gate_batch, nufhe_ctx_pin_scratch,
 * nufhe_ctx_get_tuning / _set_tuning; 6: NUFHE_ENOMEM, every entry point catches C++ exceptions, nufhe_gate_batch
 * refuses overlapping result views and MUX jobs with mu != 2^29, exact-FFT engine entry points).  A
 * binding compares nufhe_abi_version() with the macro of the header it was written against before the first real
 * call (nufhe_amd/_lib.py does). */
#define NUFHE_ABI_VERSION 6

const char *nufhe_last_error(void);
const char *nufhe_version(void);
int nufhe_abi_version(void);

/* ---- device / context (replaces reikna.cluda Thread creation, api_high_level.py:130-181) ---- */
int nufhe_device_count(int *count);
int nufhe_device_name(int device, char *buf, size_t buflen);
/* own_stream == 0: enqueue on `stream`, an existing hipStream_t (e.g. torch's current stream; NULL is
 * the device's default stream, which is what torch uses unless told otherwise) so that the caller's
 * copies and the library's kernels are ordered; own_stream != 0: `stream` is ignored and a private
 * non-blocking stream is created (the caller must synchronise through nufhe_ctx_synchronize). */
int nufhe_ctx_create(int device, void *stream, int own_stream, nufhe_ctx **ctx);
int nufhe_ctx_destroy(nufhe_ctx *ctx);
int nufhe_ctx_synchronize(nufhe_ctx *ctx);
int nufhe_ctx_device(nufhe_ctx *ctx, int *device);
void *nufhe_ctx_stream(nufhe_ctx *ctx);

/* ---- raw device memory (for callers without an allocator of their own) ---- */
int nufhe_alloc(nufhe_ctx *ctx, size_t bytes, void **d_ptr);
int nufhe_free(nufhe_ctx *ctx, void *d_ptr);
int nufhe_h2d(nufhe_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);   /* synchronous */
int nufhe_d2h(nufhe_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);   /* synchronous */

/* ---- single-process multi-GPU result collection (the reference drives several GPUs from the threads of one process and
 * its main thread collects the slices, examples/multi_gpu.py:46-114).  Slice i -- bytes[i] bytes at d_srcs[i] on the
 * device of srcs[i] -- lands at d_dst + dst_offsets[i] on the device of `dst`: a peer copy enqueued on the SOURCE
 * context's stream (ordered behind the gate that wrote the slice, no host synchronisation), and dst's stream waits for
 * all of them.  Contexts of one device are allowed (own streams).  One process per GPU gathers over RCCL instead
 * (nufhe_amd/multi_gpu.py); the library itself carries no RCCL dependency.
 * Ordering contract: the copies are ordered BEHIND everything already queued on dst's stream when the call is made (d_dst
 * may be a recycled block still in use by a kernel on that stream) and dst's stream is ordered behind the copies; work
 * queued on OTHER streams that touches d_dst is the caller's to order.  The calling thread's current HIP device is the
 * same after the call as before it. */
int nufhe_gather(nufhe_ctx *dst, void *d_dst, const size_t *dst_offsets, nufhe_ctx *const *srcs,
                 const void *const *d_srcs, const size_t *bytes, int count);

/* ---- cloud key (BootstrapKey bootstrap.py:44-92, LweKeyswitchKey lwe.py:254-308) ---- */
/* transform: NUFHE_TRANSFORM_NTT or NUFHE_TRANSFORM_FFT; mask_size: tlwe_mask_size k, 1 or 2 (both transforms)
 * (NuFHEParameters(transform_type=..., tlwe_mask_size=...), api_low_level.py:44-47).  They fix the
 * domain and the shape [n][k+1][2][k+1][...] of the bootstrapping key and the keyswitch input size k*1024. */
int nufhe_cloudkey_create(nufhe_ctx *ctx, int lwe_size, int transform, int mask_size, nufhe_cloudkey **key);
int nufhe_cloudkey_destroy(nufhe_cloudkey *key);

/* ---- engine of the NTT path (no reference counterpart: the reference has one NTT implementation,
 * polynomial_transform_ntt.py:29-131) ----
 * The gates of a NUFHE_TRANSFORM_NTT key compute the EXACT integer negacyclic convolutions of tgsw_cpu.py:82-106; which
 * arithmetic produces them is an implementation choice:
 *   NUFHE_ENGINE_NATIVE     the 64-bit prime-field NTT kernels (u64 mod 2^64 - 2^32 + 1), the default;
 *   NUFHE_ENGINE_EXACT_FFT  the fp64 folded FFT on a key whose coefficients are split into two balanced 16-bit halves:
 *                           all sums stay below 2^36, where the worst-case fp64 error (0.037) cannot
 *                           change a rounding -- the same words as the native engine for every key and every input,
 *                           at ~0.5 x its time on large batches and ~0.4 x on small ones (four waves per bit up to
 *                           2 x CUs bits; tlwe_mask_size 1 and 2).  The split image (65.5 MB for n = 500, k = 1) is
 *                           derived from the uploaded key on first use; key images / downloads are unaffected. */
#define NUFHE_ENGINE_NATIVE 0
#define NUFHE_ENGINE_EXACT_FFT 1
int nufhe_cloudkey_set_engine(nufhe_cloudkey *key, int engine);
int nufhe_cloudkey_get_engine(nufhe_cloudkey *key, int *engine);
/* Bootstrapping key in the REFERENCE's storage format, as produced by TLweTransformSamples
 * (tlwe_gpu.py:199-236) and pickled by BootstrapKey.dump (bootstrap.py:78-80).  Host pointer.
 *   NTT key: uint64 [n][k+1][2][k+1][1024], natural-order NTT, Montgomery-prepared (x * 2^64)
 *   FFT key: complex128 [n][2][2][2][512], natural-order folded FFT (fft_transform_ref) */
int nufhe_bk_upload_reference(nufhe_cloudkey *key, const void *h_bk);
/* Inverse of the above (for BootstrapKey.dump): writes the same format to the host. */
int nufhe_bk_download_reference(nufhe_cloudkey *key, void *h_bk);
/* Bootstrapping key from coefficient-domain TGSW samples, int32 [n][k+1][2][k+1][1024] on the DEVICE
 * (replaces tgsw_transform_samples, tgsw.py:135-138). */
int nufhe_bk_from_coeffs(nufhe_cloudkey *key, const int32_t *d_tgsw);
/* Keyswitch key, reference layout: a int32 [k*1024][8][4][n], b int32 [k*1024][8][4],
 * cv float [k*1024][8][4] (lwe_gpu.py:139-141).  Host pointers.  The base-0 slice must be zero
 * (lwe_cpu.py:30-33); it is checked and NUFHE_EINVAL is returned otherwise. */
int nufhe_ks_upload(nufhe_cloudkey *key, const int32_t *h_ks_a, const int32_t *h_ks_b,
                    const float *h_ks_cv);
/* Device image of a complete cloud key (bootstrapping key in the library's layout + keyswitch key) in ONE caller-owned
 * device buffer of nufhe_cloudkey_image_bytes() bytes: export on the rank that made / loaded the key, move the buffer
 * with a device collective (RCCL broadcast over xGMI; SURVEY 8e "or ncclBroadcast from rank 0" -- the reference ships
 * pickles through the host, examples/multi_gpu.py:86-107), import into a key created with the same (lwe_size, transform,
 * mask_size) on the receiving rank.  export enqueues copies on the stream; import synchronises (it rebuilds the derived
 * layouts).  The image starts with a 256-byte header (magic, NUFHE_ABI_VERSION, lwe_size, transform, mask_size, byte
 * count): import returns NUFHE_EINVAL for a buffer that is not an image, comes from another library build or holds a key
 * of other parameters (NTT and FFT keys have the same size -- the header is what tells them apart).  export synchronises
 * too (the header is staged from the host). */
int nufhe_cloudkey_image_bytes(nufhe_cloudkey *key, size_t *bytes);
int nufhe_cloudkey_export_image(nufhe_cloudkey *key, void *d_image);
int nufhe_cloudkey_import_image(nufhe_cloudkey *key, const void *d_image);
/* MakeLweKeyswitchKey on the device (lwe.py:265-295, lwe_gpu.py:63-124, lwe_gpu.mako:18-56;
 * lwe_cpu.py:27-59): builds the key in the library's own layout from DEVICE arrays
 *   d_noises_a int32 [k*1024][8][3][n]  uniform masks of the digits 1..3 (become the key's `a` as they are)
 *   d_noises_b int32 [k*1024][8][3]     centred Gaussian noises
 *   d_in_key   int32 [k*1024]           extracted TLWE key bits;  d_out_key int32 [n]  LWE key bits
 * b = in_key * digit * 2^(32 - 2 (position + 1)) + noise + <mask, out_key>, variances = `variance`. */
int nufhe_ks_make(nufhe_cloudkey *key, const int32_t *d_noises_a, const int32_t *d_noises_b,
                  const int32_t *d_in_key, const int32_t *d_out_key, float variance);
/* The keyswitch key back in the reference layout of nufhe_ks_upload (LweKeyswitchKey.dump). */
int nufhe_ks_download_reference(nufhe_cloudkey *key, int32_t *h_ks_a, int32_t *h_ks_b, float *h_ks_cv);
/* TGswAddMessage (tgsw.py:142-161, tgsw_gpu.mako:18-39; tgsw_cpu.py:109-126): d_tgsw int32
 * [count][k+1][2][k+1][1024] on the device; adds messages[s] * 2^(32 - 10 (d + 1)) to coefficient 0 of
 * polynomial m of row (m, d) of sample s. */
int nufhe_tgsw_add_message(nufhe_ctx *ctx, int32_t *d_tgsw, const int32_t *d_messages, long count,
                           int mask_size);

/* ---- LWE sample descriptor: a[bit * a_stride + i], b[bit * b_stride], cv[bit * cv_stride] ----
 * Carries what the reference's typed Reikna signatures carry (lwe_gpu.py:151-159, blind_rotate.py:226-234: the
 * array shapes are part of the computation's signature and a mismatch is refused before anything is launched):
 * `size` is the LWE dimension of THIS operand and every entry point checks it against what its key / its other
 * operands require (n for gate operands and results, k * 1024 for extracted samples) and returns NUFHE_EINVAL
 * on a mismatch, a NULL a / b, a negative stride or 0 < a_stride < size -- never an out-of-bounds access.  A RESULT of more than one bit must
 * not have a zero (broadcast) stride: every bit's work-group would write the same row. */
typedef struct {
    int32_t *a;
    int32_t *b;
    float *cv;          /* may be NULL where the variance is not needed */
    long a_stride;      /* elements between consecutive bits of a (>= size, or 0 to broadcast) */
    long b_stride;      /* elements between consecutive bits of b and cv (>= 1, or 0 to broadcast) */
    int32_t size;       /* number of mask coefficients per bit (the length of the last axis of a) */
} nufhe_lwe;

/* ---- hot path ---- */

/* LweLinear (lwe_gpu.py:287-316, lwe_gpu.mako:123-169): res = [res +] p * src on (a, b),
 * cv = [cv +] p^2 * cv; size = LWE size of both (must equal res.size and src.size). */
int nufhe_lwe_linear(nufhe_ctx *ctx, nufhe_lwe res, nufhe_lwe src, int32_t p, int add_result,
                     long nbits, int size);
/* LweNoiselessTrivialConstant (lwe_gpu.py:340-344): res = (0, mu), cv = 0 */
int nufhe_lwe_trivial_const(nufhe_ctx *ctx, nufhe_lwe res, int32_t mu, long nbits, int size);

/* bootstrap (bootstrap.py:206-229) = mod-switch + blind rotate + extract [+ keyswitch].
 * x: LWE(n) input.  no_keyswitch != 0: result is LWE(k*1024) (extracted); else LWE(n). */
int nufhe_bootstrap(nufhe_ctx *ctx, nufhe_cloudkey *key, nufhe_lwe result, nufhe_lwe x,
                    int32_t mu, long nbits, int no_keyswitch);
/* lwe_keyswitch (lwe.py:311-322, lwe_gpu.mako:59-120): LWE(k*1024) -> LWE(n) */
int nufhe_keyswitch(nufhe_ctx *ctx, nufhe_cloudkey *key, nufhe_lwe result, nufhe_lwe src,
                    long nbits);

/* Fused binary gate: result = KS(BS_mu((0, c) + pa * a + pb * b)).  Covers gate_nand and its nine
 * siblings (gates.py:81-597), e.g. NAND: c = 2^29, pa = pb = -1, mu = 2^29. */
int nufhe_gate_binary(nufhe_ctx *ctx, nufhe_cloudkey *key, nufhe_lwe result, nufhe_lwe a,
                      nufhe_lwe b, int32_t c, int32_t pa, int32_t pb, int32_t mu, long nbits);
/* Fused MUX (gates.py:600-664): a ? b : c */
int nufhe_gate_mux(nufhe_ctx *ctx, nufhe_cloudkey *key, nufhe_lwe result, nufhe_lwe a, nufhe_lwe b,
                   nufhe_lwe c, long nbits);

/* Heterogeneous gate batch: a LIST of independent gates -- different kinds, different sizes, different buffers -- as ONE
 * bootstrap launch (circuit-level fusion; the reference's circuit example is operators_integer.py:64-95,
 * where each gate is its own chain of launches).  Every gate costs the same 500 dependent blind-rotation steps however
 * few bits it has, and a launch of up to one bit per CU takes as long as one bit: four independent 64-bit gates finish in
 * the time of one.  The library (1) writes the linear pre-combination (0, c0) + pa a + pb b of every gate (MUX: its two,
 * gates.py:639-650) into the rows of one scratch array, (2) bootstraps all rows in one launch of the kernel family the
 * TOTAL row count selects, (3) adds the two extracted samples of MUX rows (+ (0, mu), gates.py:657-661) and (4) runs one
 * keyswitch whose finalize step writes every gate's slice to its own result view.  Results are word-for-word those of
 * the per-gate entry points (int32 wraparound sums are order independent).
 *   kind NUFHE_JOB_BINARY: result = KS(BS_mu((0, c0) + pa * a + pb * b)); `c` is ignored.
 *   kind NUFHE_JOB_MUX:    result = a ? b : c; c0 / pa / pb are ignored (the reference's constants are used), mu must be 2^29.
 * nbits may differ per job (0 allowed).  Every operand of every job is read (step 1) before any result is written
 * (step 4): a job may name its own or another job's result buffer as an operand and sees the values from BEFORE the
 * call (carry = MUX(same, carry, a) in place is fine); two jobs must not write the same buffer.  The job list is read
 * during the call (its tables reach the device inside kernel arguments): `jobs` may be freed on return, and the call can
 * be captured into a hipGraph like the other gate entry points (run it once eagerly first: scratch is sized on first use). */
#define NUFHE_JOB_BINARY 0
#define NUFHE_JOB_MUX 1
typedef struct {
    int32_t kind;
    int32_t c0, pa, pb;
    long nbits;
    nufhe_lwe result, a, b, c;
} nufhe_gate_job;
int nufhe_gate_batch(nufhe_ctx *ctx, nufhe_cloudkey *key, const nufhe_gate_job *jobs, int n_jobs, int32_t mu);

/* Client side.  out[i * out_stride] = base[i * base_stride] + sign * <a[i], key>
 * with int32 wraparound, sign = +1 or -1:
 *   LweEncrypt (lwe_gpu.py:186-239):  b = (mu + e) + a.s       (base = mu + e, sign = +1)
 *   LweDecrypt (lwe_gpu.py:242-284):  phi = b - a.s            (base = b,      sign = -1)
 *   MakeLweKeyswitchKey body (lwe_gpu.mako:18-56): b = (message + e) + a.s */
int nufhe_lwe_phase(nufhe_ctx *ctx, int32_t *d_out, long out_stride, const int32_t *d_a, long a_stride,
                    const int32_t *d_base, long base_stride, const int32_t *d_key, int32_t sign,
                    long count, int size);

/* ---- per-kernel entry points (reference unit tests' granularity; also used by key generation) */

/* Torus32ToPhase (numeric_functions_gpu.py:39-77) */
int nufhe_t32_to_phase(nufhe_ctx *ctx, int32_t *d_result, const int32_t *d_phase, long count,
                       uint32_t mspace_size);
/* ShiftTorusPolynomial (polynomials_gpu.py:31-86): result/source [batch][polys][1024],
 * powers [batch * powers_stride + powers_idx] */
int nufhe_shift_torus_polynomial(nufhe_ctx *ctx, int32_t *d_result, const int32_t *d_source,
                                 const int32_t *d_powers, long powers_stride, long powers_idx,
                                 long batch, int polys, int minus_one, int invert_powers);
/* tlwe_extract_lwe_samples (tlwe_gpu.mako:54-84): tlwe [batch][k+1][1024] -> a [batch][k*1024], b */
int nufhe_tlwe_extract(nufhe_ctx *ctx, int32_t *d_result_a, int32_t *d_result_b,
                       const int32_t *d_tlwe, long batch, int mask_size);
/* TGswTorus32PolynomialDecompH (tgsw_gpu.py:60-107, tgsw_cpu.py:26-49; unit test test/test_tgsw.py:44-69):
 * sample int32 [polys][1024] -> result int32 [polys][2][1024], the gadget digits (l = 2, Bg = 2^10) of every
 * coefficient -- the function the fused blind rotation applies to (X^a - 1) ACC. */
int nufhe_tgsw_decompose(nufhe_ctx *ctx, int32_t *d_result, const int32_t *d_sample, long polys);
/* TLweTransformedAddMulTo (tgsw_cpu.py:52-79; unit test test/test_tgsw.py:72-115): the multiply-accumulate of the
 * external product on REFERENCE-format transformed arrays (natural order, key Montgomery-prepared):
 *   result uint64 [batch][k+1][1024] = sum_{m,d} sample [batch][k+1][2][1024] * bk [bk_len][k+1][2][k+1][1024] (row
 *   bk_row), with the paired two-products-one-reduction arithmetic of the fused kernel. */
int nufhe_tgsw_mac(nufhe_ctx *ctx, uint64_t *d_result, const uint64_t *d_sample, const uint64_t *d_bk, int bk_len,
                   int bk_row, long batch, int mask_size);
/* Transform (transform/computation.py:28-99): batched negacyclic NTT-1024, natural order,
 * identical values to ntt_transform_ref (transform/ntt.py:30-44). */
int nufhe_ntt_forward_i32(nufhe_ctx *ctx, uint64_t *d_out, const int32_t *d_in, long batch);
int nufhe_ntt_forward_u64(nufhe_ctx *ctx, uint64_t *d_out, const uint64_t *d_in, long batch);
int nufhe_ntt_inverse_i32(nufhe_ctx *ctx, int32_t *d_out, const uint64_t *d_in, long batch);
int nufhe_ntt_inverse_u64(nufhe_ctx *ctx, uint64_t *d_out, const uint64_t *d_in, long batch);
/* Folded FFT-512 (transform/fft.py:27-51, i32_conversion=True): forward int32 [batch][1024] ->
 * complex128 [batch][512] (re, im interleaved), natural order; inverse rounds to nearest. */
int nufhe_fft_forward_i32(nufhe_ctx *ctx, double *d_out, const int32_t *d_in, long batch);
int nufhe_fft_inverse_i32(nufhe_ctx *ctx, int32_t *d_out, const double *d_in, long batch);
/* Element-wise GF(2^64 - 2^32 + 1) primitives exactly as the device kernels use them (reference test
 * surface: test/test_transform/test_arithmetic.py over transform/arithmetic.mako).  Inputs must be
 * canonical (< P) unless stated.  op: 0 a+b, 1 a-b, 2 a*b, 3 a*b+c*d, 4 a*b+c*d+e,
 * 5 a*2^(b&31), 6 a*2^shift (0 <= shift < 192), 7 reduce(a + 2^64 (uint32)b) for ANY 64-bit a,
 * 8 int32 conversions: low word = to_i32(from_i32((int32)a)), high word = to_i32(a).
 * Unused operand pointers may be NULL. */
int nufhe_ff_op(nufhe_ctx *ctx, uint64_t *d_out, const uint64_t *d_a, const uint64_t *d_b,
                const uint64_t *d_c, const uint64_t *d_d, const uint64_t *d_e, int op, int shift,
                long count);
/* The same field in the redundant 24-bit-limb form used inside the blind-rotation transforms
 * (csrc/ff24.h; replaces the add/sub/lsh family of transform/arithmetic.mako:78-161,465-1045 on the
 * hot path).  Arrays are uint32 [count][4] (limbs, or 64-bit words split into 32-bit halves); the op
 * codes are listed in csrc/l4_hook.h.  d_b / d_c may be NULL for the ops that do not read them. */
int nufhe_l4_op(nufhe_ctx *ctx, uint32_t *d_out, uint32_t *d_out2, const uint32_t *d_a,
                const uint32_t *d_b, const uint32_t *d_c, int op, int shift, long count);
/* Negacyclic product mod 2^32 of int32 polynomials through the NTT: out[b] = x[b] * y[b % y_batch] */
int nufhe_poly_mul_i32(nufhe_ctx *ctx, int32_t *d_out, const int32_t *d_x, const int32_t *d_y,
                       long batch, long y_batch);
/* TGswTransformedExternalMul (tgsw_gpu.py:110-169): accum int32 [batch][k+1][1024] in place,
 * against row bk_row of the uploaded bootstrapping key */
int nufhe_external_mul(nufhe_ctx *ctx, nufhe_cloudkey *key, int32_t *d_accum, int bk_row,
                       long batch);
/* blind_rotate (bootstrap.py:119-142): accum int32 [batch][k+1][1024] in place,
 * bara int32 [batch][bara_stride] in [0, 2N), rows [0, n_rows) of the key */
int nufhe_blind_rotate(nufhe_ctx *ctx, nufhe_cloudkey *key, int32_t *d_accum,
                       const int32_t *d_bara, long bara_stride, int n_rows, long batch);

/* ---- key generation helpers ("next" row f1; device side of TLweEncryptZero) ---- */
/* TLweEncryptZero (tlwe_gpu.py:111-196): result_a [batch][k+1][1024] =
 * (noises1, noises2 + sum_i noises1[:, i] * key[i]);  key int32 [k][1024], noises1 [batch][k][1024],
 * noises2 [batch][1024] */
int nufhe_tlwe_encrypt_zero(nufhe_ctx *ctx, int32_t *d_result_a, const int32_t *d_key,
                            const int32_t *d_noises1, const int32_t *d_noises2, long batch, int mask_size);

/* Batch-size switches of the bootstrap.  A bootstrap of up to `bits` ciphertext bits (FFT: bits/2) runs the
 * small-batch kernel (4 wavefronts share a bit, 3 for tlwe_mask_size = 2; NTT ~2.7x, FFT ~1.4x shorter latency, one
 * bit per CU at a time), larger batches the throughput kernel (one wavefront per bit).  bits < 0 restores the
 * defaults (in ciphertext bits: the CU count); 0 disables the
 * small-batch kernels.  Results are bit-identical either way (FFT: identical on every tested input).
 * The same switch (zero / non-zero) governs the four- / six-wavefronts-per-bit kernels of the fp64 paths: exact-FFT engine
 * up to 2 x CUs bits (k = 2: 1 x CUs), FFT keys up to 1 x CUs bits (the latter only while the pair switch below is on).
 * No reference counterpart (the reference always splits a bit over 512+ threads, blind_rotate.py:89-187). */
int nufhe_ctx_set_team_max_bits(nufhe_ctx *ctx, long bits);
/* k = 1 only.  NTT: batches above the small-batch limit and up to `bits` run the medium-batch kernel (2 wavefronts
 * per bit, up to 4 bits per CU at a time; ~1.5x shorter latency than one wavefront per bit while the batch cannot
 * give every SIMD two bits).  bits < 0 restores the default (4 x the CU count); 0 disables it.
 * FFT: any non-zero value enables the 2-wavefront kernel for batches up to 3 x the CU count (it is ahead of the
 * 4-wavefront kernel at every size, which then only runs when this switch is 0).
 * k = 2, either transform: any non-zero value enables the 3-wavefront kernel without a partial-sum buffer for every
 * batch above the small-batch limit (rounds of 2 x CUs bits; with 0 the one-wavefront-per-bit kernels run instead). */
int nufhe_ctx_set_pair_max_bits(nufhe_ctx *ctx, long bits);
/* NTT, k = 1: batches up to the team limit run EIGHT wavefronts per bit (two per digit transform, the two half rings of
 * X^1024 + 1 = (X^512 - i)(X^512 + i); csrc/ntt512_half.h) instead of four -- 4.8 ms instead of 5.8 ms per gate (measured); the key
 * is kept a second time in the half-ring layout (made on first use).  0 switches back to the 4-wave team kernel (default
 * 1).  Identical results either way. */
int nufhe_ctx_set_team8(nufhe_ctx *ctx, int enable);
/* Where the switch points come from.  Every batch-size switch above is a figure in ciphertext bits that the library
 * derives from the DEVICE when the context is created: a table of measured switch points keyed by (gcnArchName, CU
 * count) -- MI355X / gfx950 / 256 CUs today -- and, for a part without an entry, the measured part's bits-per-CU ratios
 * scaled by its CU count (`measured` = 0).  nufhe_ctx_get_tuning reads the context's set; nufhe_ctx_set_tuning replaces
 * it (NULL: back to the table).  The per-switch calls above stay what they were: overrides on top of this set
 * (negative = no override).  No switch changes a result: the kernel families are bit-identical (NTT) / identical on
 * every tested input (FFT); tests/test_gpu_gates.py::test_every_switch_point_plus_minus_one_bit. */
typedef struct {
    long team_max_bits;        /* NTT: up to here the 8- / 4- / 3-wavefronts-per-bit kernels */
    long team_max_bits_fft;    /* FFT: the 4- / 3-wavefront kernels up to here */
    long pair_max_bits_ntt;    /* NTT, k = 1: up to here 2 wavefronts per bit; 0 = off */
    long pair_max_bits_fft;    /* FFT, k = 1: up to here 2 wavefronts per bit; 0 = off */
    long ks_mfma_min_bits;     /* keyswitch on the matrix cores for batches ABOVE this size (mode 1) */
    int32_t ring_k2;           /* k = 2: the 3-wavefront kernel without a partial-sum buffer above the team limit */
    int32_t k2_roomy_ratio_pct;/* NTT k = 2 wave kernels: duration of a 6 x CUs round over a 4 x CUs round, percent */
    int32_t measured;          /* out: 1 = (arch_name, num_cus) has a table entry */
    int32_t num_cus;           /* out */
    char arch_name[64];        /* out: gcnArchName */
} nufhe_tuning;
int nufhe_ctx_get_tuning(nufhe_ctx *ctx, nufhe_tuning *tuning);
int nufhe_ctx_set_tuning(nufhe_ctx *ctx, const nufhe_tuning *tuning);
/* Keyswitch kernel.  The matrix-core kernel writes the digit selection as a one-hot int8 matrix product against the
 * key split into four signed byte planes (v_mfma_i32_16x16x64_i8); the LDS-window kernel gathers key rows by digit.
 * mode 1 (default): matrix cores for batches of more than tuning.ks_mfma_min_bits (2 x CUs) bits (0.30 vs 1.15 ms at 4096 bits), the LDS-window
 * kernel below (it is ahead up to there); 0: never; 2: always.  Both are exact integer sums mod 2^32:
 * identical results. */
int nufhe_ctx_set_keyswitch_mfma(nufhe_ctx *ctx, int mode);

/* Scratch pinning for captured graphs.  The library's scratch buffers (extracted samples, keyswitch accumulators) grow on
 * demand; a hipGraph captured from gate calls holds raw pointers into them.  While the pin count is positive a buffer
 * that has to grow is kept alive (retired) instead of freed, so earlier captures stay valid; the retired buffers are
 * released when the count returns to zero.  delta = +1 before capturing, -1 when the graph is destroyed
 * (nufhe_amd/graph.py does both).  A buffer that would have to grow DURING a capture is an error (NUFHE_EINVAL): run the
 * circuit once eagerly first. */
int nufhe_ctx_pin_scratch(nufhe_ctx *ctx, int delta);

/* ---- measurement: time of the last fused gate / bootstrap kernels, from HIP events on the
 * context's stream (milliseconds; blind-rotate kernel and keyswitch kernels separately) ---- */
int nufhe_profile_enable(nufhe_ctx *ctx, int enable);
int nufhe_profile_last(nufhe_ctx *ctx, float *blind_rotate_ms, float *keyswitch_ms);
/* The same two durations for EVERY gate profiled since nufhe_profile_enable / the previous call (the most recent
 * `capacity` of at most 256, oldest first), read in one go: a timed loop records events and never waits on one.
 * Synchronises on the last gate's end event; resets the history. */
int nufhe_profile_history(nufhe_ctx *ctx, float *blind_rotate_ms, float *keyswitch_ms, int capacity, int *count);
/* Sustained shader clock of the last profiled bootstrap launch, measured inside the kernel: one wavefront of the
 * wave-per-bit kernel reads the shader-clock counter and the constant 100 MHz counter around its blind rotation;
 * *shader_ghz = their ratio, *wave_ms = how long that wavefront ran (the kernel runs ceil(bits / (8 x CUs)) such
 * rounds).  Fails for batches that ran the small- or medium-batch kernels. */
int nufhe_profile_clock(nufhe_ctx *ctx, double *shader_ghz, double *wave_ms);
/* Start / end (ms after the first wave's start, from the constant 100 MHz counter) and SIMD number of every wave of
 * work-group 0 of the last profiled wave-per-bit bootstrap launch (up to 8): two waves that share a SIMD are paced
 * against each other (csrc/blind_rotate.h, BrPace) and must END together -- the self-check of bench.py and of
 * tests/test_gpu_kernels.py::test_simd_partners_finish_together. */
int nufhe_profile_waves(nufhe_ctx *ctx, double *start_ms, double *end_ms, int *simd, int capacity, int *count);

#ifdef __cplusplus
}
#endif
#endif /* NUFHE_HIP_H */
