// api.hip -- the extern "C" boundary declared in include/nufhe_hip.h.
// Owns: the HIP stream of a context, the twiddle tables, lazily grown scratch (extracted LWE(1024)
// samples between the bootstrap and keyswitch kernels, keyswitch accumulators), and the device
// copies of the cloud key in the layouts the kernels want.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <exception>
#include <new>
#include <vector>

#include "../../include/nufhe_hip.h"
#include "kernels.h"
#include "blind_rotate_xfft.h"
#include "ntt_tables.h"

namespace {

// a fixed buffer, not a std::string: fail() is called from the catch handlers of the boundary and must not allocate
thread_local char g_last_error[512] = {0};

int fail(int code, const char *fmt, ...) noexcept
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
    return code;
}

// Every extern "C" body sits between these two: no C++ exception crosses the boundary (include/nufhe_hip.h).  The host
// side allocates (contexts, key holders, staging vectors, job tables): std::bad_alloc becomes NUFHE_ENOMEM, anything
// else NUFHE_EHIP, both with a message for nufhe_last_error().
#define NUFHE_API_BEGIN try {
#define NUFHE_API_END                                                                                      \
    } catch (const std::bad_alloc &) {                                                                     \
        return fail(NUFHE_ENOMEM, "%s: out of host memory (std::bad_alloc)", __func__);                    \
    } catch (const std::exception &ex) {                                                                   \
        return fail(NUFHE_EHIP, "%s: unexpected C++ exception: %s", __func__, ex.what());                  \
    } catch (...) {                                                                                        \
        return fail(NUFHE_EHIP, "%s: unexpected C++ exception", __func__);                                 \
    }

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return fail(NUFHE_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

struct Scratch {
    void *ptr = nullptr;
    size_t bytes = 0;
};

}  // namespace

#define PROFILE_RING 256

struct nufhe_ctx {
    int device = 0;
    int num_cus = 256;         // of this context's device
    char arch_name[64] = {0};  // gcnArchName of this context's device
    BrTuning tuning;           // batch-size switches derived from (arch_name, num_cus) at creation (kernels.h)
    hipStream_t stream = nullptr;
    bool own_stream = false;
    u64 *d_tw1f = nullptr;
    u64 *d_tw1x = nullptr;    // forward table permuted for the limb-form transform (ntt_make_tw1x)
    u64 *d_nth = nullptr;     // table block of the half-ring transforms (ntt512_half.h, NTH_TABLE_ELEMS)
    int team8 = 1;            // smallest NTT batches on the 8-waves-per-bit half-ring kernel (nufhe_ctx_set_team8)
    unsigned long long *d_clock = nullptr;   // BrLaunch::clock_probe target (CLOCK_PROBE_WORDS words)
    u64 *d_tw1i = nullptr;
    cplx *d_ftw1 = nullptr;
    cplx *d_ftw2 = nullptr;
    Scratch ext_a, ext_b, ks_acc, ks_digits;
    Scratch batch_a, batch_b;  // nufhe_gate_batch: the combined LWE(n) inputs of all rotations of a batch
    Scratch xfft_park;         // exact-FFT engine: the accumulators' parking space, 8 KiB per rotation (blind_rotate_xfft.h)
    // Scratch that a captured hipGraph points into must not be freed while the graph lives: while scratch_pins > 0 a buffer
    // that has to grow is RETIRED (kept until the last pin is released) instead of freed (nufhe_ctx_pin_scratch)
    int scratch_pins = 0;
    std::vector<void *> retired;
    // job tables of nufhe_gate_batch: a ring of device buffers (a slot is rewritten four batches later, in stream order
    // behind the kernels that read it).  They are FILLED BY KERNELS whose arguments carry the table bytes
    // (launch_write_table), not by copies from host memory: nothing on the host has to outlive the call, and a captured
    // hipGraph holds the tables inside its kernel nodes.
    Scratch tables[4];
    unsigned table_next = 0;
    long team_max_bits = -1;   // nufhe_ctx_set_team_max_bits override of tuning.team_max_bits; -1 = none
    long pair_max_bits = -1;   // nufhe_ctx_set_pair_max_bits override of the pair / ring switches; -1 = none
    int ks_mfma = 1;           // keyswitch on the matrix cores (k_keyswitch_mfma): 0 never, 1 batches > tuning.ks_mfma_min_bits, 2 always
    bool profile = false;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // [0..2]: begin / after bootstrap / end of the LAST profiled gate
    hipEvent_t ev_dst = nullptr;                               // nufhe_gather: "everything queued on this stream so far"
    bool ev_valid = false;
    // history of the profiled gates since the last nufhe_profile_history: PROFILE_RING event triples, so that a timed loop
    // never has to synchronise on an event in order to keep its kernel timings (created on first use)
    std::vector<hipEvent_t> ring;
    long ring_count = 0;
};

struct nufhe_cloudkey {
    nufhe_ctx *ctx = nullptr;
    int n = 0;
    int transform = 0;        // 0 = NTT (u64 residues), 1 = FFT (complex128)
    int mask_size = 1;        // tlwe_mask_size k: TGSW rows have (k+1) * 2 * (k+1) polynomials
    long bk_polys() const { return (long)n * (mask_size + 1) * 2 * (mask_size + 1); }
    int ext_size() const { return mask_size * 1024; }
    void *d_bk = nullptr;     // wave layout: u64 [n][8][1024] or complex128 [n][8][512] (same bytes)
    u64 *d_bk_half = nullptr; // NTT, k = 1: the same key in the half-ring layout of k_bootstrap_team8, made on first use
    bool bk_half_valid = false;
    // NTT, k = 1, engine NUFHE_ENGINE_EXACT_FFT: the same key as the split fp64 image of blind_rotate_xfft.h
    // (complex128 [n][8][2][512], 65.5 MB), derived from d_bk on first use after every change of the key
    int engine = 0;
    cplx *d_bkx = nullptr;
    bool bkx_valid = false;
    i32 *d_ks_a3 = nullptr;   // [k*1024][8][3][n]
    signed char *d_ks_planes = nullptr;   // the same key as signed byte planes in MFMA operand order (k_ks_planes)
    i32 *d_ks_b = nullptr;    // [k*1024][8][4]
    float *d_ks_cv = nullptr; // [k*1024][8][4]
    float *d_cv_table = nullptr;   // [k*1024*8 + 1] for uniform-variance keys (KsFinal::cv_table), else NULL
};

namespace {

// is the context's stream being captured into a graph?  (a failed query counts as "no" and leaves no sticky error)
bool stream_capturing(nufhe_ctx *ctx)
{
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(ctx->stream, &cap) != hipSuccess) { (void)hipGetLastError(); return false; }
    return cap != hipStreamCaptureStatusNone;
}

// buffers retired under a pin whose release fell into a capture (nufhe_ctx_pin_scratch) are freed by the next
// non-capturing call that comes through here
void free_retired_if_unpinned(nufhe_ctx *ctx)
{
    if (ctx->scratch_pins != 0 || ctx->retired.empty()) return;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { (void)hipGetLastError(); return; }
    for (void *p : ctx->retired) hipFree(p);
    ctx->retired.clear();
}

int ensure(nufhe_ctx *ctx, Scratch &s, size_t bytes)
{
    if (s.bytes >= bytes) return NUFHE_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    // no allocation of any kind during a capture: neither growth nor a first-time hipMalloc
    if (stream_capturing(ctx))
        return fail(NUFHE_EINVAL, "a scratch buffer would have to grow (%zu -> %zu bytes) during stream capture: run the "
                    "circuit once eagerly at this size before capturing it", s.bytes, bytes);
    free_retired_if_unpinned(ctx);
    if (s.ptr) {
        if (ctx->scratch_pins > 0) {
            ctx->retired.push_back(s.ptr);          // a captured graph may still point into it
        } else {
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            HIP_TRY(hipFree(s.ptr));
        }
        s.ptr = nullptr;
        s.bytes = 0;
    }
    HIP_TRY(hipMalloc(&s.ptr, bytes));
    s.bytes = bytes;
    return NUFHE_OK;
}

inline LweView view(const nufhe_lwe &x) { return LweView{x.a, x.b, x.cv, x.a_stride, x.b_stride}; }

inline BrSource source(const nufhe_lwe &x, i32 p) { return BrSource{x.a, x.b, x.a_stride, x.b_stride, p}; }

// The checks the reference gets from its typed computation signatures (lwe_gpu.py:151-159, blind_rotate.py:226-234):
// an operand whose LWE dimension is not the one the key / the operation needs is refused, not read out of bounds.
static int check_lwe(const nufhe_lwe &x, int expect_size, const char *what, bool need_cv, long nbits,
                     bool is_output = false)
{
    // a RESULT of more than one bit must give every bit its own row: a broadcast (zero) stride, or rows of `b` that
    // alias, would have all work-groups write the same words
    if (is_output && nbits > 1 && (x.a_stride == 0 || x.b_stride == 0))
        return fail(NUFHE_EINVAL, "%s: zero (broadcast) stride on an output of %ld bits", what, nbits);
    // (an empty batch has no storage: its pointers may be NULL)
    if (nbits > 0 && (x.a == nullptr || x.b == nullptr)) return fail(NUFHE_EINVAL, "%s: NULL a / b pointer", what);
    if (nbits > 0 && need_cv && x.cv == nullptr) return fail(NUFHE_EINVAL, "%s: NULL variance pointer", what);
    if (x.size != expect_size)
        return fail(NUFHE_EINVAL, "%s: LWE size %d, expected %d", what, (int)x.size, expect_size);
    if (x.a_stride < 0 || x.b_stride < 0) return fail(NUFHE_EINVAL, "%s: negative stride", what);
    if (x.a_stride != 0 && x.a_stride < x.size)
        return fail(NUFHE_EINVAL, "%s: a_stride %ld shorter than the sample (%d)", what, x.a_stride, (int)x.size);
    return NUFHE_OK;
}

int check_ctx(nufhe_ctx *ctx)
{
    if (!ctx) return fail(NUFHE_EINVAL, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    return NUFHE_OK;
}

int check_key(nufhe_ctx *ctx, nufhe_cloudkey *key, bool need_bk, bool need_ks)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!key) return fail(NUFHE_EINVAL, "null cloud key");
    if (key->ctx != ctx) return fail(NUFHE_EINVAL, "cloud key belongs to a different context");
    if (need_bk && !key->d_bk) return fail(NUFHE_ENOKEY, "bootstrapping key not uploaded");
    if (need_ks && !key->d_ks_a3) return fail(NUFHE_ENOKEY, "keyswitch key not uploaded");
    return NUFHE_OK;
}

// the three events of the gate about to be profiled: a fresh slot of the history ring (ev[0..2] alias the last one)
// (the slot is only COUNTED by profile_commit, after its end event has been recorded: a gate that fails half way leaves
// the history as it was)
int profile_slot(nufhe_ctx *ctx, hipEvent_t pe[3])
{
    if (ctx->ring.empty()) {
        std::vector<hipEvent_t> fresh(3 * PROFILE_RING, nullptr);
        for (auto &e : fresh) {
            hipError_t err = hipEventCreate(&e);
            if (err != hipSuccess) {
                for (hipEvent_t made : fresh)
                    if (made) hipEventDestroy(made);
                return fail(NUFHE_EHIP, "hipEventCreate (profile ring): %s", hipGetErrorString(err));
            }
        }
        ctx->ring.swap(fresh);
    }
    const long slot = ctx->ring_count % PROFILE_RING;
    for (int k = 0; k < 3; k++) pe[k] = ctx->ring[3 * slot + k];
    return NUFHE_OK;
}

void profile_commit(nufhe_ctx *ctx, hipEvent_t pe[3])
{
    for (int k = 0; k < 3; k++) ctx->ev[k] = pe[k];      // ev[0..2]: the last gate whose three events were all recorded
    ctx->ring_count++;
    ctx->ev_valid = true;
}

// profiling is skipped while the stream is being captured into a graph: event pairs recorded inside a capture have no
// elapsed time
bool profiling_now(nufhe_ctx *ctx)
{
    return ctx->profile && !stream_capturing(ctx);
}

int pick_j_per_block(long nbits, int input_size, int num_cus)
{
    const long tiles = (nbits + KS_TILE_BITS - 1) / KS_TILE_BITS;
    int jsplit = 1;
    // two 64 KiB-LDS blocks fit a CU: aim at one full wave of 2 x CUs blocks (measured: 512 blocks 1.35 ms,
    // 1024 blocks 1.38 ms, 256 blocks 2.01 ms for 4096 bits)
    const long target = 2L * num_cus;
    while (jsplit < 128 && tiles * jsplit < target) jsplit *= 2;
    return input_size / jsplit;
}

// keyswitch of LWE(1024) src1 (+ src2) (+ constant on b) into `result`
int run_keyswitch(nufhe_ctx *ctx, nufhe_cloudkey *key, const nufhe_lwe &result, const i32 *s1a, long s1as,
                  const i32 *s1b, long s1bs, const i32 *s2a, long s2as, const i32 *s2b, long s2bs, i32 c0,
                  long nbits, const BatchOut *batch_outs = nullptr, int n_batch_outs = 0)
{
    int rc = ensure(ctx, ctx->ks_acc, (size_t)nbits * key->n * sizeof(u32));
    if (rc) return rc;
    KsLaunch P;
    P.acc = (u32 *)ctx->ks_acc.ptr;
    P.ks_a3 = key->d_ks_a3;
    P.src1_a = s1a; P.src2_a = s2a;
    P.src1_stride = s1as; P.src2_stride = s2as;
    P.nbits = nbits;
    P.n = key->n;
    P.input_size = key->ext_size();
    P.j_per_block = pick_j_per_block(nbits, P.input_size, ctx->num_cus);
    // the matrix-core kernel walks the whole key per 64-bit tile (0.23 ms at any size up to 1024 bits, 0.30 ms at 4096);
    // the LDS-window kernel is ahead up to 2 x CUs bits (0.08 ms for one bit, 0.23 ms at 512, 0.34 ms at 768)
    P.ks_planes = (ctx->ks_mfma == 2 || (ctx->ks_mfma == 1 && nbits > ctx->tuning.ks_mfma_min_bits)) ? key->d_ks_planes : nullptr;
    P.digits_t = nullptr;
    if (P.ks_planes) {
        rc = ensure(ctx, ctx->ks_digits, (size_t)P.input_size * ((nbits + 63) & ~63L) * sizeof(unsigned short));
        if (rc) return rc;
        P.digits_t = (unsigned short *)ctx->ks_digits.ptr;
    }
    KsFinal F;
    F.batch_outs = batch_outs;
    F.n_batch_outs = n_batch_outs;
    F.acc = P.acc;
    F.ks_b = key->d_ks_b; F.ks_cv = key->d_ks_cv;
    F.src1_a = s1a; F.src2_a = s2a; F.src1_b = s1b; F.src2_b = s2b;
    F.src1_stride = s1as; F.src2_stride = s2as; F.src1_bstride = s1bs; F.src2_bstride = s2bs;
    F.c0 = c0;
    F.out_a = result.a; F.out_b = result.b; F.out_cv = result.cv;
    F.out_a_stride = result.a_stride; F.out_b_stride = result.b_stride;
    F.n = key->n;
    F.input_size = key->ext_size();
    F.cv_table = key->d_cv_table;
    HIP_TRY(launch_keyswitch(P, F, ctx->stream));
    return NUFHE_OK;
}

}  // namespace

extern "C" {

const char *nufhe_last_error(void) { return g_last_error; }
const char *nufhe_version(void) { return "nufhe_hip 0.6 (gfx950)"; }
int nufhe_abi_version(void) { return NUFHE_ABI_VERSION; }

int nufhe_device_count(int *count)
{
    NUFHE_API_BEGIN
    if (!count) return fail(NUFHE_EINVAL, "null argument");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *count = 0;
        return fail(NUFHE_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = c;
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_device_name(int device, char *buf, size_t buflen)
{
    NUFHE_API_BEGIN
    if (!buf || !buflen) return fail(NUFHE_EINVAL, "null argument");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_ctx_create(int device, void *stream, int own_stream, nufhe_ctx **out)
{
    NUFHE_API_BEGIN
    if (!out) return fail(NUFHE_EINVAL, "null argument");
    int count = 0;
    int rc = nufhe_device_count(&count);
    if (rc) return rc;
    if (count == 0) return fail(NUFHE_ENODEV, "no HIP device available");
    if (device < 0 || device >= count) return fail(NUFHE_EINVAL, "device %d out of range [0, %d)", device, count);
    HIP_TRY(hipSetDevice(device));
    nufhe_ctx *ctx = new nufhe_ctx();
    ctx->device = device;
    if (!own_stream) {
        ctx->stream = (hipStream_t)stream;   // NULL = the device's default stream
    } else {
        hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete ctx; return fail(NUFHE_EHIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
        ctx->own_stream = true;
    }
    std::vector<u64> f(1024), i(1024);
    ntt_make_tables(f.data(), i.data());
    std::vector<u64> fx(1024);
    ntt_make_tw1x(fx.data(), f.data());
    hipError_t e = hipMalloc((void **)&ctx->d_tw1f, 1024 * sizeof(u64));
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_clock, CLOCK_PROBE_WORDS * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(ctx->d_clock, 0, CLOCK_PROBE_WORDS * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_nth, NTH_TABLE_ELEMS * sizeof(u64));
    if (e == hipSuccess) {
        std::vector<u64> nth(NTH_TABLE_ELEMS);
        nth_make_tables(nth.data());
        e = hipMemcpy(ctx->d_nth, nth.data(), NTH_TABLE_ELEMS * sizeof(u64), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_tw1x, 1024 * sizeof(u64));
    if (e == hipSuccess) e = hipMemcpy(ctx->d_tw1x, fx.data(), 1024 * sizeof(u64), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_tw1i, 1024 * sizeof(u64));
    if (e == hipSuccess) e = hipMemcpy(ctx->d_tw1f, f.data(), 1024 * sizeof(u64), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ctx->d_tw1i, i.data(), 1024 * sizeof(u64), hipMemcpyHostToDevice);
    std::vector<cplx> g1(FFT_TW1_ELEMS), g2(FFT_TW2_ELEMS);
    fft_make_tables(g1.data(), g2.data());
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_ftw1, FFT_TW1_ELEMS * sizeof(cplx));
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_ftw2, FFT_TW2_ELEMS * sizeof(cplx));
    if (e == hipSuccess) e = hipMemcpy(ctx->d_ftw1, g1.data(), FFT_TW1_ELEMS * sizeof(cplx), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ctx->d_ftw2, g2.data(), FFT_TW2_ELEMS * sizeof(cplx), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = kernels_init_device(&ctx->num_cus, ctx->arch_name, sizeof(ctx->arch_name));
    if (e == hipSuccess) ctx->tuning = br_tuning_for(ctx->arch_name, ctx->num_cus);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev[3]);      // nufhe_gather; ev[0..2] alias slots of the profile ring
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_dst, hipEventDisableTiming);
    if (e != hipSuccess) {
        nufhe_ctx_destroy(ctx);
        return fail(NUFHE_EHIP, "context setup: %s", hipGetErrorString(e));
    }
    *out = ctx;
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_ctx_destroy(nufhe_ctx *ctx)
{
    NUFHE_API_BEGIN
    if (!ctx) return NUFHE_OK;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    if (ctx->d_tw1f) hipFree(ctx->d_tw1f);
    if (ctx->d_tw1x) hipFree(ctx->d_tw1x);
    if (ctx->d_nth) hipFree(ctx->d_nth);
    if (ctx->d_clock) hipFree(ctx->d_clock);
    if (ctx->d_tw1i) hipFree(ctx->d_tw1i);
    if (ctx->d_ftw1) hipFree(ctx->d_ftw1);
    if (ctx->d_ftw2) hipFree(ctx->d_ftw2);
    if (ctx->ext_a.ptr) hipFree(ctx->ext_a.ptr);
    if (ctx->ext_b.ptr) hipFree(ctx->ext_b.ptr);
    if (ctx->ks_acc.ptr) hipFree(ctx->ks_acc.ptr);
    if (ctx->ks_digits.ptr) hipFree(ctx->ks_digits.ptr);
    if (ctx->batch_a.ptr) hipFree(ctx->batch_a.ptr);
    if (ctx->batch_b.ptr) hipFree(ctx->batch_b.ptr);
    if (ctx->xfft_park.ptr) hipFree(ctx->xfft_park.ptr);
    for (void *p : ctx->retired) hipFree(p);
    for (auto &t : ctx->tables)
        if (t.ptr) hipFree(t.ptr);
    if (ctx->ev[3]) hipEventDestroy(ctx->ev[3]);
    if (ctx->ev_dst) hipEventDestroy(ctx->ev_dst);
    for (hipEvent_t e : ctx->ring)
        if (e) hipEventDestroy(e);
    if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_ctx_synchronize(nufhe_ctx *ctx)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_ctx_device(nufhe_ctx *ctx, int *device)
{
    NUFHE_API_BEGIN
    if (!ctx || !device) return fail(NUFHE_EINVAL, "null argument");
    *device = ctx->device;
    return NUFHE_OK;
    NUFHE_API_END
}

void *nufhe_ctx_stream(nufhe_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int nufhe_alloc(nufhe_ctx *ctx, size_t bytes, void **d_ptr)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!d_ptr) return fail(NUFHE_EINVAL, "null argument");
    HIP_TRY(hipMalloc(d_ptr, bytes ? bytes : 1));
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_free(nufhe_ctx *ctx, void *d_ptr)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipFree(d_ptr));
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_h2d(nufhe_ctx *ctx, void *d_dst, const void *h_src, size_t bytes)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_d2h(nufhe_ctx *ctx, void *h_dst, const void *d_src, size_t bytes)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return NUFHE_OK;
    NUFHE_API_END
}

// ---- single-process multi-GPU: result collection (SURVEY 8b "nufhe_gather(ctxs...)") ------------------
// The reference's multi-GPU scheme is ONE process with a thread and a Thread object per GPU whose main thread collects
// the result slices (examples/multi_gpu.py:46-114).  For a host built that way: slice i (bytes[i] bytes at d_srcs[i], on
// srcs[i]'s device) is copied to d_dst + dst_offsets[i] on dst's device by hipMemcpyPeerAsync ON THE SOURCE CONTEXT'S
// STREAM -- behind the gate that produced it, without a host synchronisation -- and dst's stream is made to wait for
// every copy, so whatever the caller enqueues on dst afterwards sees the gathered array.  (One process per GPU uses the
// RCCL gather of nufhe_amd/multi_gpu.py instead.)
int nufhe_gather(nufhe_ctx *dst, void *d_dst, const size_t *dst_offsets, nufhe_ctx *const *srcs, const void *const *d_srcs,
                 const size_t *bytes, int count)
{
    NUFHE_API_BEGIN
    if (!dst || !srcs || !d_srcs || !bytes || !dst_offsets || count < 0) return fail(NUFHE_EINVAL, "null argument");
    for (int i = 0; i < count; i++) {
        if (!srcs[i]) return fail(NUFHE_EINVAL, "null source context %d", i);
        if (bytes[i] && (!d_srcs[i] || !d_dst)) return fail(NUFHE_EINVAL, "null buffer of slice %d", i);
    }
    int caller_device = -1;
    HIP_TRY(hipGetDevice(&caller_device));
    // Ordering contract, both ways.  (1) The copies wait for everything ALREADY QUEUED on dst's stream: d_dst may be a
    // recycled block whose last user -- a kernel still in flight on dst's stream -- must finish before a source stream
    // writes into it.  (2) dst's stream waits for every copy, so whatever is queued on dst afterwards sees the slices.
    int rc = NUFHE_OK;
    {
        hipError_t e = hipSetDevice(dst->device);
        if (e == hipSuccess) e = hipEventRecord(dst->ev_dst, dst->stream);
        if (e != hipSuccess) rc = fail(NUFHE_EHIP, "nufhe_gather, destination stream: %s", hipGetErrorString(e));
    }
    for (int i = 0; i < count && rc == NUFHE_OK; i++) {
        if (!bytes[i]) continue;
        nufhe_ctx *src = srcs[i];
        const bool same_queue = src == dst || src->stream == dst->stream;
        hipError_t e = hipSetDevice(src->device);
        if (e == hipSuccess && !same_queue) e = hipStreamWaitEvent(src->stream, dst->ev_dst, 0);
        if (e == hipSuccess) {
            if (src->device == dst->device)
                e = hipMemcpyAsync((char *)d_dst + dst_offsets[i], d_srcs[i], bytes[i], hipMemcpyDeviceToDevice, src->stream);
            else
                e = hipMemcpyPeerAsync((char *)d_dst + dst_offsets[i], dst->device, d_srcs[i], src->device, bytes[i],
                                       src->stream);
        }
        if (e == hipSuccess && !same_queue) {
            e = hipEventRecord(src->ev[3], src->stream);
            if (e == hipSuccess) e = hipSetDevice(dst->device);
            if (e == hipSuccess) e = hipStreamWaitEvent(dst->stream, src->ev[3], 0);
        }
        if (e != hipSuccess) rc = fail(NUFHE_EHIP, "nufhe_gather, slice %d: %s", i, hipGetErrorString(e));
    }
    // the calling thread's current device is left as it was found
    if (caller_device >= 0) (void)hipSetDevice(caller_device);
    return rc;
    NUFHE_API_END
}

// ---- cloud key ---------------------------------------------------------------------------

int nufhe_cloudkey_create(nufhe_ctx *ctx, int lwe_size, int transform, int mask_size, nufhe_cloudkey **key)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!key) return fail(NUFHE_EINVAL, "null argument");
    if (transform != NUFHE_TRANSFORM_NTT && transform != NUFHE_TRANSFORM_FFT)
        return fail(NUFHE_EINVAL, "unknown transform %d", transform);
    if (mask_size < 1 || mask_size > 2)
        return fail(NUFHE_EINVAL, "unsupported tlwe_mask_size %d (1 or 2)", mask_size);
    if (lwe_size < 1 || lwe_size > BR_MAX_LWE)
        return fail(NUFHE_EINVAL, "lwe_size %d out of range [1, %d]", lwe_size, BR_MAX_LWE);
    nufhe_cloudkey *k = new nufhe_cloudkey();
    k->ctx = ctx;
    k->n = lwe_size;
    k->transform = transform;
    k->mask_size = mask_size;
    *key = k;
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_cloudkey_destroy(nufhe_cloudkey *key)
{
    NUFHE_API_BEGIN
    if (!key) return NUFHE_OK;
    hipSetDevice(key->ctx->device);
    hipStreamSynchronize(key->ctx->stream);
    if (key->d_bk) hipFree(key->d_bk);
    if (key->d_bk_half) hipFree(key->d_bk_half);
    if (key->d_bkx) hipFree(key->d_bkx);
    if (key->d_ks_a3) hipFree(key->d_ks_a3);
    if (key->d_ks_planes) hipFree(key->d_ks_planes);
    if (key->d_ks_b) hipFree(key->d_ks_b);
    if (key->d_ks_cv) hipFree(key->d_ks_cv);
    if (key->d_cv_table) hipFree(key->d_cv_table);
    delete key;
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_cloudkey_set_engine(nufhe_cloudkey *key, int engine)
{
    NUFHE_API_BEGIN
    if (!key) return fail(NUFHE_EINVAL, "null cloud key");
    if (engine != NUFHE_ENGINE_NATIVE && engine != NUFHE_ENGINE_EXACT_FFT) return fail(NUFHE_EINVAL, "unknown engine %d", engine);
    if (engine == NUFHE_ENGINE_EXACT_FFT && key->transform != NUFHE_TRANSFORM_NTT)
        return fail(NUFHE_EINVAL, "the exact-FFT engine serves NTT keys (this key: transform %d)", key->transform);
    key->engine = engine;
    if (engine == NUFHE_ENGINE_NATIVE && key->d_bkx) {
        // the 65.5 MB image goes with the engine
        int rc = check_ctx(key->ctx);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(key->ctx->stream));
        HIP_TRY(hipFree(key->d_bkx));
        key->d_bkx = nullptr;
        key->bkx_valid = false;
    }
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_cloudkey_get_engine(nufhe_cloudkey *key, int *engine)
{
    NUFHE_API_BEGIN
    if (!key || !engine) return fail(NUFHE_EINVAL, "null argument");
    *engine = key->engine;
    return NUFHE_OK;
    NUFHE_API_END
}

static int alloc_bk(nufhe_cloudkey *key)
{
    if (!key->d_bk) HIP_TRY(hipMalloc((void **)&key->d_bk, (size_t)key->bk_polys() * BK_POLY_ELEMS * sizeof(u64)));
    return NUFHE_OK;
}

int nufhe_bk_upload_reference(nufhe_cloudkey *key, const void *h_bk)
{
    NUFHE_API_BEGIN
    if (!key || !h_bk) return fail(NUFHE_EINVAL, "null argument");
    nufhe_ctx *ctx = key->ctx;
    int rc = check_ctx(ctx);
    if (rc) return rc;
    rc = alloc_bk(key);
    if (rc) return rc;
    const size_t bytes = (size_t)key->bk_polys() * BK_POLY_ELEMS * sizeof(u64);
    u64 *tmp = nullptr;
    HIP_TRY(hipMalloc((void **)&tmp, bytes));
    hipError_t e = hipMemcpyAsync(tmp, h_bk, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = key->transform == NUFHE_TRANSFORM_NTT
                ? (key->bk_half_valid = key->bkx_valid = false, launch_bk_from_reference((u64 *)key->d_bk, tmp, key->bk_polys(), ctx->stream))
                : launch_bkf_permute((cplx *)key->d_bk, (const cplx *)tmp, key->bk_polys(), 0, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(tmp);
    if (e != hipSuccess) return fail(NUFHE_EHIP, "bk upload: %s", hipGetErrorString(e));
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_bk_download_reference(nufhe_cloudkey *key, void *h_bk)
{
    NUFHE_API_BEGIN
    if (!key || !h_bk) return fail(NUFHE_EINVAL, "null argument");
    nufhe_ctx *ctx = key->ctx;
    int rc = check_key(ctx, key, true, false);
    if (rc) return rc;
    const size_t bytes = (size_t)key->bk_polys() * BK_POLY_ELEMS * sizeof(u64);
    u64 *tmp = nullptr;
    HIP_TRY(hipMalloc((void **)&tmp, bytes));
    hipError_t e = key->transform == NUFHE_TRANSFORM_NTT
                       ? launch_bk_to_reference(tmp, (const u64 *)key->d_bk, key->bk_polys(), ctx->stream)
                       : launch_bkf_permute((cplx *)tmp, (const cplx *)key->d_bk, key->bk_polys(), 1, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h_bk, tmp, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(tmp);
    if (e != hipSuccess) return fail(NUFHE_EHIP, "bk download: %s", hipGetErrorString(e));
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_bk_from_coeffs(nufhe_cloudkey *key, const int32_t *d_tgsw)
{
    NUFHE_API_BEGIN
    if (!key || !d_tgsw) return fail(NUFHE_EINVAL, "null argument");
    nufhe_ctx *ctx = key->ctx;
    int rc = check_ctx(ctx);
    if (rc) return rc;
    rc = alloc_bk(key);
    if (rc) return rc;
    key->bk_half_valid = key->bkx_valid = false;
    if (key->transform == NUFHE_TRANSFORM_NTT)
        HIP_TRY(launch_bk_from_coeffs((u64 *)key->d_bk, d_tgsw, key->bk_polys(), ctx->d_tw1f, ctx->d_tw1i, ctx->stream));
    else
        HIP_TRY(launch_bkf_from_coeffs((cplx *)key->d_bk, d_tgsw, key->bk_polys(), ctx->d_ftw1, ctx->d_ftw2, ctx->stream));
    return NUFHE_OK;
    NUFHE_API_END
}

// Uniform-variance keys (KsFinal::cv_table): table of the float32 partial sums v, v + v, (v + v) + v, ...
// exactly as a sequential float32 accumulation produces them (volatile: one rounding per addition)
static int set_cv_table(nufhe_cloudkey *key, bool uniform, float v)
{
    if (key->d_cv_table) { hipFree(key->d_cv_table); key->d_cv_table = nullptr; }
    if (!uniform) return NUFHE_OK;
    const long terms = (long)key->ext_size() * NUFHE_KS_T;
    std::vector<float> table(terms + 1);
    volatile float s = 0.0f;
    table[0] = 0.0f;
    for (long c = 1; c <= terms; c++) { s = s + v; table[c] = s; }
    HIP_TRY(hipMalloc((void **)&key->d_cv_table, table.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(key->d_cv_table, table.data(), table.size() * sizeof(float), hipMemcpyHostToDevice));
    return NUFHE_OK;
}

static int alloc_ks(nufhe_cloudkey *key)
{
    const long rows = (long)key->ext_size() * NUFHE_KS_T;
    if (!key->d_ks_a3) HIP_TRY(hipMalloc((void **)&key->d_ks_a3, (size_t)rows * 3 * key->n * sizeof(int32_t)));
    if (!key->d_ks_b) HIP_TRY(hipMalloc((void **)&key->d_ks_b, rows * 4 * sizeof(int32_t)));
    if (!key->d_ks_cv) HIP_TRY(hipMalloc((void **)&key->d_ks_cv, rows * 4 * sizeof(float)));
    return NUFHE_OK;
}

// (re)build the byte planes of the keyswitch key for the matrix-core kernel
static int build_ks_planes(nufhe_cloudkey *key)
{
    nufhe_ctx *ctx = key->ctx;
    if (key->n > KSM_COLS) return NUFHE_OK;          // wider keys keep the LDS-window kernel
    if (!key->d_ks_planes) HIP_TRY(hipMalloc((void **)&key->d_ks_planes, ks_planes_bytes(key->ext_size())));
    HIP_TRY(launch_ks_planes(key->d_ks_planes, key->d_ks_a3, key->ext_size(), key->n, ctx->stream));
    return NUFHE_OK;
}

int nufhe_ks_upload(nufhe_cloudkey *key, const int32_t *h_ks_a, const int32_t *h_ks_b, const float *h_ks_cv)
{
    NUFHE_API_BEGIN
    if (!key || !h_ks_a || !h_ks_b || !h_ks_cv) return fail(NUFHE_EINVAL, "null argument");
    nufhe_ctx *ctx = key->ctx;
    int rc = check_ctx(ctx);
    if (rc) return rc;
    const int n = key->n;
    const long rows = (long)key->ext_size() * NUFHE_KS_T;
    // base-0 slice must be zero (lwe_cpu.py:30-33): the kernels never read it
    for (long r = 0; r < rows; r++) {
        const int32_t *row0 = h_ks_a + (r * 4) * n;
        for (int i = 0; i < n; i++)
            if (row0[i] != 0) return fail(NUFHE_EINVAL, "keyswitch key: base-0 slice of ks_a is not zero (row %ld)", r);
        if (h_ks_b[r * 4] != 0 || h_ks_cv[r * 4] != 0.0f)
            return fail(NUFHE_EINVAL, "keyswitch key: base-0 slice of ks_b/ks_cv is not zero (row %ld)", r);
    }
    bool uniform = true;
    const float v0 = h_ks_cv[1];
    for (long r = 0; r < rows && uniform; r++)
        for (int h = 1; h < 4; h++)
            if (h_ks_cv[r * 4 + h] != v0) { uniform = false; break; }
    rc = set_cv_table(key, uniform && v0 >= 0.0f, v0);
    if (rc) return rc;
    std::vector<int32_t> packed((size_t)rows * 3 * n);
    for (long r = 0; r < rows; r++)
        memcpy(packed.data() + (size_t)r * 3 * n, h_ks_a + ((size_t)r * 4 + 1) * n, sizeof(int32_t) * 3 * n);
    rc = alloc_ks(key);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(key->d_ks_a3, packed.data(), packed.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(key->d_ks_b, h_ks_b, rows * 4 * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(key->d_ks_cv, h_ks_cv, rows * 4 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    rc = build_ks_planes(key);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return NUFHE_OK;
    NUFHE_API_END
}

// ---- device image of a whole cloud key (replication over RCCL: examples/multi_gpu.py) ----
// bootstrapping key in the wave layout | ks_a3 | ks_b | ks_cv, each part starting at a multiple of 256 bytes
static size_t image_part(size_t bytes) { return (bytes + 255) & ~(size_t)255; }

// the image starts with a 256-byte header that names what it holds; import refuses anything else
#define IMAGE_HEADER_BYTES 256
struct ImageHeader {
    char magic[8];            // "NUFHEIMG"
    int32_t abi, n, transform, mask_size;
    uint64_t total_bytes;
};
static_assert(sizeof(ImageHeader) <= IMAGE_HEADER_BYTES, "image header");

static ImageHeader image_header(const nufhe_cloudkey *key, size_t total)
{
    ImageHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, "NUFHEIMG", 8);
    h.abi = NUFHE_ABI_VERSION; h.n = key->n; h.transform = key->transform; h.mask_size = key->mask_size;
    h.total_bytes = total;
    return h;
}

static void image_layout(const nufhe_cloudkey *key, size_t off[5])
{
    const size_t rows = (size_t)key->ext_size() * NUFHE_KS_T;
    off[0] = IMAGE_HEADER_BYTES;
    off[1] = off[0] + image_part((size_t)key->bk_polys() * BK_POLY_ELEMS * sizeof(u64));
    off[2] = off[1] + image_part(rows * 3 * key->n * sizeof(int32_t));
    off[3] = off[2] + image_part(rows * 4 * sizeof(int32_t));
    off[4] = off[3] + image_part(rows * 4 * sizeof(float));
}

int nufhe_cloudkey_image_bytes(nufhe_cloudkey *key, size_t *bytes)
{
    NUFHE_API_BEGIN
    if (!key || !bytes) return fail(NUFHE_EINVAL, "null argument");
    size_t off[5];
    image_layout(key, off);
    *bytes = off[4];
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_cloudkey_export_image(nufhe_cloudkey *key, void *d_image)
{
    NUFHE_API_BEGIN
    if (!key || !d_image) return fail(NUFHE_EINVAL, "null argument");
    nufhe_ctx *ctx = key->ctx;
    int rc = check_key(ctx, key, true, true);
    if (rc) return rc;
    size_t off[5];
    image_layout(key, off);
    char *dst = (char *)d_image;
    const void *src[4] = {key->d_bk, key->d_ks_a3, key->d_ks_b, key->d_ks_cv};
    const size_t rows = (size_t)key->ext_size() * NUFHE_KS_T;
    const size_t len[4] = {(size_t)key->bk_polys() * BK_POLY_ELEMS * sizeof(u64), rows * 3 * key->n * sizeof(int32_t),
                           rows * 4 * sizeof(int32_t), rows * 4 * sizeof(float)};
    for (int i = 0; i < 4; i++)
        HIP_TRY(hipMemcpyAsync(dst + off[i], src[i], len[i], hipMemcpyDeviceToDevice, ctx->stream));
    unsigned char head[IMAGE_HEADER_BYTES];
    memset(head, 0, sizeof(head));
    const ImageHeader h = image_header(key, off[4]);
    memcpy(head, &h, sizeof(h));
    HIP_TRY(hipMemcpyAsync(dst, head, sizeof(head), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));      // `head` is a stack buffer
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_cloudkey_import_image(nufhe_cloudkey *key, const void *d_image)
{
    NUFHE_API_BEGIN
    if (!key || !d_image) return fail(NUFHE_EINVAL, "null argument");
    nufhe_ctx *ctx = key->ctx;
    int rc = check_ctx(ctx);
    if (rc) return rc;
    size_t off[5];
    image_layout(key, off);
    {
        // an image made for another transform / mask size / LWE size / library build has the same or a plausible size:
        // the header is what tells them apart
        ImageHeader got;
        HIP_TRY(hipMemcpyAsync(&got, d_image, sizeof(got), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        const ImageHeader want = image_header(key, off[4]);
        if (memcmp(got.magic, want.magic, 8) != 0) return fail(NUFHE_EINVAL, "not a cloud-key image (no NUFHEIMG header)");
        if (got.abi != want.abi)
            return fail(NUFHE_EINVAL, "cloud-key image of ABI version %d, this library is %d", (int)got.abi, (int)want.abi);
        if (got.n != want.n || got.transform != want.transform || got.mask_size != want.mask_size || got.total_bytes != want.total_bytes)
            return fail(NUFHE_EINVAL, "cloud-key image holds (n %d, transform %d, mask size %d), the key was created with (%d, %d, %d)",
                        (int)got.n, (int)got.transform, (int)got.mask_size, key->n, key->transform, key->mask_size);
    }
    if ((rc = alloc_bk(key)) || (rc = alloc_ks(key))) return rc;
    const char *src = (const char *)d_image;
    void *dst[4] = {key->d_bk, key->d_ks_a3, key->d_ks_b, key->d_ks_cv};
    const size_t rows = (size_t)key->ext_size() * NUFHE_KS_T;
    const size_t len[4] = {(size_t)key->bk_polys() * BK_POLY_ELEMS * sizeof(u64), rows * 3 * key->n * sizeof(int32_t),
                           rows * 4 * sizeof(int32_t), rows * 4 * sizeof(float)};
    for (int i = 0; i < 4; i++)
        HIP_TRY(hipMemcpyAsync(dst[i], src + off[i], len[i], hipMemcpyDeviceToDevice, ctx->stream));
    key->bk_half_valid = key->bkx_valid = false;
    if ((rc = build_ks_planes(key))) return rc;
    // the variance table of a uniform-variance key (what nufhe_ks_upload derives from the host arrays)
    std::vector<float> cv(rows * 4);
    HIP_TRY(hipMemcpyAsync(cv.data(), key->d_ks_cv, len[3], hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    bool uniform = true;
    const float v0 = cv[1];
    for (size_t r = 0; r < rows && uniform; r++)
        for (int h = 1; h < 4; h++)
            if (cv[r * 4 + h] != v0) { uniform = false; break; }
    return set_cv_table(key, uniform && v0 >= 0.0f, v0);
    NUFHE_API_END
}

int nufhe_ks_make(nufhe_cloudkey *key, const int32_t *d_noises_a, const int32_t *d_noises_b, const int32_t *d_in_key,
                  const int32_t *d_out_key, float variance)
{
    NUFHE_API_BEGIN
    if (!key || !d_noises_a || !d_noises_b || !d_in_key || !d_out_key) return fail(NUFHE_EINVAL, "null argument");
    nufhe_ctx *ctx = key->ctx;
    int rc = check_ctx(ctx);
    if (rc) return rc;
    rc = alloc_ks(key);
    if (rc) return rc;
    const long rows = (long)key->ext_size() * NUFHE_KS_T * 3;
    // the masks are the key's `a` as they are: one device-to-device copy into the key's own storage
    HIP_TRY(hipMemcpyAsync(key->d_ks_a3, d_noises_a, (size_t)rows * key->n * sizeof(int32_t), hipMemcpyDeviceToDevice,
                           ctx->stream));
    HIP_TRY(launch_ks_make(key->d_ks_b, key->d_ks_cv, key->d_ks_a3, d_noises_b, d_in_key, d_out_key, variance, rows,
                           key->n, ctx->stream));
    int rcp = build_ks_planes(key);
    if (rcp) return rcp;
    return set_cv_table(key, variance >= 0.0f, variance);
    NUFHE_API_END
}

int nufhe_ks_download_reference(nufhe_cloudkey *key, int32_t *h_ks_a, int32_t *h_ks_b, float *h_ks_cv)
{
    NUFHE_API_BEGIN
    if (!key || !h_ks_a || !h_ks_b || !h_ks_cv) return fail(NUFHE_EINVAL, "null argument");
    nufhe_ctx *ctx = key->ctx;
    int rc = check_key(ctx, key, false, true);
    if (rc) return rc;
    const long groups = (long)key->ext_size() * NUFHE_KS_T;
    const size_t bytes = (size_t)groups * 4 * key->n * sizeof(int32_t);
    int32_t *tmp = nullptr;
    HIP_TRY(hipMalloc((void **)&tmp, bytes));
    hipError_t e = launch_ks_to_reference(tmp, key->d_ks_a3, groups, key->n, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h_ks_a, tmp, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h_ks_b, key->d_ks_b, groups * 4 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h_ks_cv, key->d_ks_cv, groups * 4 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(tmp);
    if (e != hipSuccess) return fail(NUFHE_EHIP, "ks download: %s", hipGetErrorString(e));
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_tgsw_add_message(nufhe_ctx *ctx, int32_t *d_tgsw, const int32_t *d_messages, long count, int mask_size)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!d_tgsw || !d_messages) return fail(NUFHE_EINVAL, "null argument");
    if (count < 0 || mask_size < 1) return fail(NUFHE_EINVAL, "bad sizes");
    HIP_TRY(launch_tgsw_add_message(d_tgsw, d_messages, count, mask_size, ctx->stream));
    return NUFHE_OK;
    NUFHE_API_END
}

// ---- hot path ----------------------------------------------------------------------------

int nufhe_lwe_linear(nufhe_ctx *ctx, nufhe_lwe res, nufhe_lwe src, int32_t p, int add_result, long nbits, int size)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (nbits < 0 || size < 1) return fail(NUFHE_EINVAL, "bad sizes");
    if ((rc = check_lwe(res, size, "lwe_linear result", true, nbits, true)) || (rc = check_lwe(src, size, "lwe_linear source", true, nbits)))
        return rc;
    HIP_TRY(launch_lwe_linear(view(res), view(src), p, add_result, nbits, size, ctx->stream));
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_lwe_trivial_const(nufhe_ctx *ctx, nufhe_lwe res, int32_t mu, long nbits, int size)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (nbits < 0 || size < 1) return fail(NUFHE_EINVAL, "bad sizes");
    if ((rc = check_lwe(res, size, "lwe_trivial_const result", true, nbits, true))) return rc;
    HIP_TRY(launch_lwe_trivial_const(view(res), mu, nbits, size, ctx->stream));
    return NUFHE_OK;
    NUFHE_API_END
}

// exact-FFT engine: (re)build the split fp64 image of an NTT key from its device copy -- inverse NTT of every key
// polynomial back to its int32 coefficients (exact), 16-bit split, forward FFT of both halves (kernels_xfft.hip)
static int ensure_bkx(nufhe_ctx *ctx, nufhe_cloudkey *key)
{
    if (key->bkx_valid) return NUFHE_OK;
    if (stream_capturing(ctx))
        return fail(NUFHE_EINVAL, "the exact-FFT key image is built on first use: run one gate eagerly before capturing");
    const long polys = key->bk_polys();
    if (!key->d_bkx) HIP_TRY(hipMalloc((void **)&key->d_bkx, (size_t)polys * BKX_POLY_ELEMS * sizeof(cplx)));
    i32 *coeffs = nullptr;      // [polys][1024] coefficients, then one flag word
    HIP_TRY(hipMalloc((void **)&coeffs, ((size_t)polys * 1024 + 1) * sizeof(i32)));
    int *flag = (int *)(coeffs + (size_t)polys * 1024);
    int not_int32 = 0;
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), ctx->stream);
    if (e == hipSuccess) e = launch_bk_to_coeffs(coeffs, (const u64 *)key->d_bk, polys, ctx->d_tw1f, ctx->d_tw1i, flag, ctx->stream);
    if (e == hipSuccess) e = launch_bkx_from_coeffs(key->d_bkx, coeffs, polys, ctx->d_ftw1, ctx->d_ftw2, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&not_int32, flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(coeffs);
    if (e != hipSuccess) return fail(NUFHE_EHIP, "exact-FFT key image: %s", hipGetErrorString(e));
    // the engine's guarantee covers keys made of int32 torus polynomials (every key the scheme generates); a key of
    // arbitrary field elements (synthetic test data) has 64-bit coefficients and only the prime-field kernels compute with it
    if (not_int32)
        return fail(NUFHE_EINVAL, "the bootstrapping key is not the transform of int32 polynomials: the exact-FFT engine "
                    "cannot serve it (use NUFHE_ENGINE_NATIVE)");
    key->bkx_valid = true;
    return NUFHE_OK;
}

static bool uses_xfft(const nufhe_cloudkey *key)
{
    return key->engine == NUFHE_ENGINE_EXACT_FFT && key->transform == NUFHE_TRANSFORM_NTT;
}

// the context's tuning with the legacy per-switch overrides applied (nufhe_ctx_set_team_max_bits: an explicit value
// counts half for the FFT kernels; nufhe_ctx_set_pair_max_bits: the NTT k = 1 limit itself, for FFT / k = 2 an on-off switch)
static BrTuning effective_tuning(const nufhe_ctx *ctx)
{
    BrTuning T = ctx->tuning;
    if (ctx->team_max_bits >= 0) {
        T.team_max_bits = ctx->team_max_bits;
        T.team_max_bits_fft = ctx->team_max_bits / 2;
    }
    if (ctx->pair_max_bits >= 0) {
        T.pair_max_bits_ntt = ctx->pair_max_bits;
        if (ctx->pair_max_bits == 0) { T.pair_max_bits_fft = 0; T.ring_k2 = 0; }
    }
    return T;
}

// key, tables and kernel-family preparation common to every fused bootstrap launch of `total` rotations
// *transform receives the kernel family to launch (the key's transform, or BR_TRANSFORM_XFFT for the exact-FFT engine)
static int prepare_launch(nufhe_ctx *ctx, nufhe_cloudkey *key, BrLaunch &P, long total, i32 mu, int *transform)
{
    *transform = key->transform;
    if (uses_xfft(key)) {
        int rc = ensure_bkx(ctx, key);
        if (rc) return rc;
        P.park = nullptr;
        if (key->mask_size == 1) {          // (k = 2 keeps its accumulator in LDS: nothing to park)
            if ((rc = ensure(ctx, ctx->xfft_park, (size_t)total * 2048 * sizeof(u32)))) return rc;
            P.park = (u32 *)ctx->xfft_park.ptr;
        }
        P.nbits_total = total;
        P.bk = key->d_bkx;
        P.n = key->n;
        P.mu = mu;
        P.tw_a = ctx->d_ftw1;
        P.tw_b = ctx->d_ftw2;
        *transform = BR_TRANSFORM_XFFT;
        return NUFHE_OK;
    }
    P.nbits_total = total;
    P.bk = key->d_bk;
    P.bk_half = nullptr;
    P.tw_half = ctx->d_nth;
    {
        // smallest batches of the NTT / k = 1 path run the half-ring team kernel: it reads the key in its own layout (a
        // permutation of the same field elements), converted once per key on first use
        const long team_limit = effective_tuning(ctx).team_max_bits;
        if (ctx->team8 && key->transform == NUFHE_TRANSFORM_NTT && key->mask_size == 1 && total <= team_limit) {
            // (one-time, per key: 33 MB and one conversion launch.)  A failure here is not an error of the gate: the
            // launch falls back to the 4-wave team kernel, which reads the ordinary layout (P.bk_half stays NULL).
            if (!key->d_bk_half &&
                hipMalloc((void **)&key->d_bk_half, (size_t)key->bk_polys() * BK_POLY_ELEMS * sizeof(u64)) != hipSuccess) {
                key->d_bk_half = nullptr;
                (void)hipGetLastError();
            }
            if (key->d_bk_half && !key->bk_half_valid &&
                launch_bk_to_half(key->d_bk_half, (const u64 *)key->d_bk, key->bk_polys(), ctx->stream) == hipSuccess)
                key->bk_half_valid = true;
            if (key->d_bk_half && key->bk_half_valid) P.bk_half = key->d_bk_half;
        }
    }
    P.n = key->n;
    P.mu = mu;
    const bool fft = key->transform == NUFHE_TRANSFORM_FFT;
    P.tw_a = fft ? (const void *)ctx->d_ftw1 : (const void *)ctx->d_tw1x;
    P.tw_b = fft ? (const void *)ctx->d_ftw2 : (const void *)ctx->d_tw1i;
    return NUFHE_OK;
}

// shared by bootstrap / gate_binary / gate_mux: jobs -> (optional keyswitch) -> result
static int run_gate(nufhe_ctx *ctx, nufhe_cloudkey *key, const nufhe_lwe &result, const BrJob *jobs, int njobs,
                    i32 mu, long nbits, bool keyswitch, i32 ks_c0)
{
    if (nbits < 0) return fail(NUFHE_EINVAL, "negative batch size");
    if (nbits == 0) return NUFHE_OK;
    const long total = nbits * njobs;
    BrLaunch P;
    memset(&P, 0, sizeof(P));
    for (int j = 0; j < njobs; j++) P.job[j] = jobs[j];
    P.bits_per_job = nbits;
    int transform = 0;
    {
        int rcp = prepare_launch(ctx, key, P, total, mu, &transform);
        if (rcp) return rcp;
    }
    const bool prof = profiling_now(ctx);
    P.clock_probe = prof ? ctx->d_clock : nullptr;
    if (keyswitch) {
        const int ext = key->ext_size();
        int rc = ensure(ctx, ctx->ext_a, (size_t)total * ext * sizeof(i32));
        if (rc) return rc;
        rc = ensure(ctx, ctx->ext_b, (size_t)total * sizeof(i32));
        if (rc) return rc;
        P.out_a = (i32 *)ctx->ext_a.ptr; P.out_a_stride = ext;
        P.out_b = (i32 *)ctx->ext_b.ptr; P.out_b_stride = 1;
    } else {
        P.out_a = result.a; P.out_a_stride = result.a_stride;
        P.out_b = result.b; P.out_b_stride = result.b_stride;
    }
    hipEvent_t pe[3] = {nullptr, nullptr, nullptr};
    if (prof) {
        int rcp = profile_slot(ctx, pe);
        if (rcp) return rcp;
        HIP_TRY(hipMemsetAsync(ctx->d_clock, 0, CLOCK_PROBE_WORDS * sizeof(unsigned long long), ctx->stream));
        HIP_TRY(hipEventRecord(pe[0], ctx->stream));
    }
    HIP_TRY(launch_bootstrap(P, transform, key->mask_size, effective_tuning(ctx), ctx->stream));
    if (prof) HIP_TRY(hipEventRecord(pe[1], ctx->stream));
    if (keyswitch) {
        const i32 *ea = (const i32 *)ctx->ext_a.ptr, *eb = (const i32 *)ctx->ext_b.ptr;
        const int ext = key->ext_size();
        int rc = run_keyswitch(ctx, key, result, ea, ext, eb, 1, njobs == 2 ? ea + nbits * ext : nullptr, ext,
                               njobs == 2 ? eb + nbits : nullptr, 1, ks_c0, nbits);
        if (rc) return rc;
    }
    if (prof) {
        HIP_TRY(hipEventRecord(pe[2], ctx->stream));
        profile_commit(ctx, pe);
    }
    return NUFHE_OK;
}

int nufhe_bootstrap(nufhe_ctx *ctx, nufhe_cloudkey *key, nufhe_lwe result, nufhe_lwe x, int32_t mu, long nbits,
                    int no_keyswitch)
{
    NUFHE_API_BEGIN
    int rc = check_key(ctx, key, true, !no_keyswitch);
    if (rc) return rc;
    if ((rc = check_lwe(x, key->n, "bootstrap input", false, nbits)) ||
        (rc = check_lwe(result, no_keyswitch ? key->ext_size() : key->n, "bootstrap result", !no_keyswitch, nbits, true)))
        return rc;
    BrJob job;
    memset(&job, 0, sizeof(job));
    job.s0 = source(x, 1);
    job.c0 = 0;
    return run_gate(ctx, key, result, &job, 1, mu, nbits, !no_keyswitch, 0);
    NUFHE_API_END
}

int nufhe_keyswitch(nufhe_ctx *ctx, nufhe_cloudkey *key, nufhe_lwe result, nufhe_lwe src, long nbits)
{
    NUFHE_API_BEGIN
    int rc = check_key(ctx, key, false, true);
    if (rc) return rc;
    if (nbits < 0) return fail(NUFHE_EINVAL, "negative batch size");
    if ((rc = check_lwe(src, key->ext_size(), "keyswitch source", false, nbits)) ||
        (rc = check_lwe(result, key->n, "keyswitch result", true, nbits, true)))
        return rc;
    if (nbits == 0) return NUFHE_OK;
    const bool prof = profiling_now(ctx);
    hipEvent_t pe[3] = {nullptr, nullptr, nullptr};
    if (prof) {
        int rcp = profile_slot(ctx, pe);
        if (rcp) return rcp;
        HIP_TRY(hipEventRecord(pe[0], ctx->stream));
        HIP_TRY(hipEventRecord(pe[1], ctx->stream));
    }
    rc = run_keyswitch(ctx, key, result, src.a, src.a_stride, src.b, src.b_stride, nullptr, 0, nullptr, 0, 0, nbits);
    if (rc) return rc;
    if (prof) {
        HIP_TRY(hipEventRecord(pe[2], ctx->stream));
        profile_commit(ctx, pe);
    }
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_gate_binary(nufhe_ctx *ctx, nufhe_cloudkey *key, nufhe_lwe result, nufhe_lwe a, nufhe_lwe b, int32_t c,
                      int32_t pa, int32_t pb, int32_t mu, long nbits)
{
    NUFHE_API_BEGIN
    int rc = check_key(ctx, key, true, true);
    if (rc) return rc;
    if ((rc = check_lwe(result, key->n, "gate result", true, nbits, true)) || (rc = check_lwe(a, key->n, "gate operand a", false, nbits)) ||
        (rc = check_lwe(b, key->n, "gate operand b", false, nbits)))
        return rc;
    BrJob job;
    memset(&job, 0, sizeof(job));
    job.s0 = source(a, pa);
    job.s1 = source(b, pb);
    job.c0 = c;
    return run_gate(ctx, key, result, &job, 1, mu, nbits, true, 0);
    NUFHE_API_END
}

int nufhe_gate_mux(nufhe_ctx *ctx, nufhe_cloudkey *key, nufhe_lwe result, nufhe_lwe a, nufhe_lwe b, nufhe_lwe c,
                   long nbits)
{
    NUFHE_API_BEGIN
    int rc = check_key(ctx, key, true, true);
    if (rc) return rc;
    if ((rc = check_lwe(result, key->n, "mux result", true, nbits, true)) || (rc = check_lwe(a, key->n, "mux operand a", false, nbits)) ||
        (rc = check_lwe(b, key->n, "mux operand b", false, nbits)) || (rc = check_lwe(c, key->n, "mux operand c", false, nbits)))
        return rc;
    const i32 MU = (i32)(1u << 29);
    BrJob jobs[2];
    memset(jobs, 0, sizeof(jobs));
    jobs[0].s0 = source(a, 1);  jobs[0].s1 = source(b, 1);  jobs[0].c0 = -MU;   // (0,-1/8) + a + b, gates.py:639-641
    jobs[1].s0 = source(a, -1); jobs[1].s1 = source(c, 1);  jobs[1].c0 = -MU;   // (0,-1/8) - a + c, gates.py:648-650
    return run_gate(ctx, key, result, jobs, 2, MU, nbits, true, MU);             // (0,1/8) + u1 + u2 -> KS, :657-664
    NUFHE_API_END
}

int nufhe_gate_batch(nufhe_ctx *ctx, nufhe_cloudkey *key, const nufhe_gate_job *jobs, int n_jobs, int32_t mu)
{
    NUFHE_API_BEGIN
    int rc = check_key(ctx, key, true, true);
    if (rc) return rc;
    if (n_jobs < 0 || (n_jobs > 0 && !jobs)) return fail(NUFHE_EINVAL, "null job list");
    long out_bits = 0, mux_bits = 0;
    int live = 0;
    for (int j = 0; j < n_jobs; j++) {
        const nufhe_gate_job &g = jobs[j];
        char what[64];
        if (g.kind != NUFHE_JOB_BINARY && g.kind != NUFHE_JOB_MUX) return fail(NUFHE_EINVAL, "job %d: unknown kind %d", j, (int)g.kind);
        if (g.nbits < 0) return fail(NUFHE_EINVAL, "job %d: negative batch size", j);
        snprintf(what, sizeof(what), "job %d result", j);
        if ((rc = check_lwe(g.result, key->n, what, true, g.nbits, true))) return rc;
        snprintf(what, sizeof(what), "job %d operand a", j);
        if ((rc = check_lwe(g.a, key->n, what, false, g.nbits))) return rc;
        snprintf(what, sizeof(what), "job %d operand b", j);
        if ((rc = check_lwe(g.b, key->n, what, false, g.nbits))) return rc;
        if (g.kind == NUFHE_JOB_MUX) {
            snprintf(what, sizeof(what), "job %d operand c", j);
            if ((rc = check_lwe(g.c, key->n, what, false, g.nbits))) return rc;
            mux_bits += g.nbits;
        }
        out_bits += g.nbits;
        live += g.nbits > 0;
    }
    if (out_bits == 0) return NUFHE_OK;
    // the MUX fold and pre-combination are written for the gate constant 1/8 (gates.py:639-664)
    if (mux_bits > 0 && mu != (i32)(1u << 29))
        return fail(NUFHE_EINVAL, "a batch with NUFHE_JOB_MUX jobs needs mu = 2^29 (got %d)", (int)mu);
    // result views of two jobs must not overlap: their finalize blocks run concurrently.  Views are `nbits` rows of
    // `width` words `stride` apart; two views whose byte ranges meet are still disjoint when they interleave rows of
    // one array (equal strides, row offsets at least a row apart modulo the stride).
    auto rows_meet = [](const int32_t *p, long sp, long np, const int32_t *q, long sq, long nq, long width) {
        const long pb = (long)(intptr_t)p, qb = (long)(intptr_t)q, w = width * (long)sizeof(i32);
        const long pe = pb + (np - 1) * sp * (long)sizeof(i32) + w, qe = qb + (nq - 1) * sq * (long)sizeof(i32) + w;
        if (pe <= qb || qe <= pb) return false;
        if (sp == sq && sp > 0) {
            const long s = sp * (long)sizeof(i32);
            const long d = ((qb - pb) % s + s) % s;
            if (d >= w && s - d >= w) return false;
        }
        return true;
    };
    for (int j = 0; j < n_jobs; j++) {
        const nufhe_gate_job &gj = jobs[j];
        for (int i = 0; i < j && gj.nbits > 0; i++) {
            const nufhe_gate_job &gi = jobs[i];
            if (gi.nbits == 0) continue;
            if (rows_meet(gi.result.a, gi.result.a_stride, gi.nbits, gj.result.a, gj.result.a_stride, gj.nbits, key->n) ||
                rows_meet(gi.result.b, gi.result.b_stride, gi.nbits, gj.result.b, gj.result.b_stride, gj.nbits, 1))
                return fail(NUFHE_EINVAL, "jobs %d and %d: result views overlap", i, j);
        }
    }
    const long rows = out_bits + mux_bits;
    // tables: first rotations of all gates in job order (row = output bit), then the second rotations of the MUX gates
    std::vector<BatchRot> rots;
    std::vector<BatchOut> outs;
    rots.reserve(2 * (size_t)live);
    outs.reserve(live);
    auto rot = [](const nufhe_lwe &x, i32 p0, const nufhe_lwe &y, i32 p1, i32 c0, long start, long nbits) {
        BatchRot r;
        memset(&r, 0, sizeof(r));
        r.a0 = x.a; r.b0 = x.b; r.a0_stride = x.a_stride; r.b0_stride = x.b_stride; r.p0 = p0;
        r.a1 = y.a; r.b1 = y.b; r.a1_stride = y.a_stride; r.b1_stride = y.b_stride; r.p1 = p1;
        r.c0 = c0; r.start = start; r.nbits = nbits;
        return r;
    };
    const i32 MU8 = (i32)(1u << 29);
    long pos = 0, pos2 = out_bits;
    for (int j = 0; j < n_jobs; j++) {
        const nufhe_gate_job &g = jobs[j];
        if (g.nbits == 0) continue;
        BatchOut o;
        memset(&o, 0, sizeof(o));
        o.a = g.result.a; o.b = g.result.b; o.cv = g.result.cv;
        o.a_stride = g.result.a_stride; o.b_stride = g.result.b_stride;
        o.start = pos; o.nbits = g.nbits; o.second = -1;
        if (g.kind == NUFHE_JOB_MUX) {
            rots.push_back(rot(g.a, 1, g.b, 1, -MU8, pos, g.nbits));          // (0,-1/8) + a + b, gates.py:639-641
            o.second = pos2;
            pos2 += g.nbits;
        } else {
            rots.push_back(rot(g.a, g.pa, g.b, g.pb, g.c0, pos, g.nbits));
        }
        outs.push_back(o);
        pos += g.nbits;
    }
    for (int j = 0; j < n_jobs; j++) {
        const nufhe_gate_job &g = jobs[j];
        if (g.nbits == 0 || g.kind != NUFHE_JOB_MUX) continue;
        const long start = rots.empty() ? 0 : rots.back().start + rots.back().nbits;
        rots.push_back(rot(g.a, -1, g.c, 1, -MU8, start, g.nbits));           // (0,-1/8) - a + c, gates.py:648-650
    }
    const size_t rot_bytes = rots.size() * sizeof(BatchRot), out_bytes = outs.size() * sizeof(BatchOut);
    // the tables travel as kernel arguments (4 KiB per launch at most) into the next slot of the ring
    // (all four slots are sized together: a capture that follows one eager warm-up run must not meet an unallocated slot)
    const size_t table_bytes = rot_bytes + out_bytes < 4096 ? 4096 : 2 * (rot_bytes + out_bytes);
    for (Scratch &t : ctx->tables)
        if ((rc = ensure(ctx, t, table_bytes))) return rc;
    Scratch &slot = ctx->tables[ctx->table_next++ % 4];
    HIP_TRY(launch_write_table(slot.ptr, rots.data(), rot_bytes, ctx->stream));
    HIP_TRY(launch_write_table((char *)slot.ptr + rot_bytes, outs.data(), out_bytes, ctx->stream));
    const BatchRot *d_rots = (const BatchRot *)slot.ptr;
    const BatchOut *d_outs = (const BatchOut *)((const char *)slot.ptr + rot_bytes);

    const int ext = key->ext_size();
    if ((rc = ensure(ctx, ctx->batch_a, (size_t)rows * key->n * sizeof(i32))) || (rc = ensure(ctx, ctx->batch_b, (size_t)rows * sizeof(i32))) ||
        (rc = ensure(ctx, ctx->ext_a, (size_t)rows * ext * sizeof(i32))) || (rc = ensure(ctx, ctx->ext_b, (size_t)rows * sizeof(i32))))
        return rc;
    HIP_TRY(launch_batch_combine((i32 *)ctx->batch_a.ptr, (i32 *)ctx->batch_b.ptr, d_rots, (int)rots.size(), rows, key->n,
                                 ctx->stream));
    BrLaunch P;
    memset(&P, 0, sizeof(P));
    P.job[0].s0 = BrSource{(const i32 *)ctx->batch_a.ptr, (const i32 *)ctx->batch_b.ptr, (long)key->n, 1, 1};
    P.job[0].c0 = 0;
    P.bits_per_job = rows;
    int transform = 0;
    if ((rc = prepare_launch(ctx, key, P, rows, mu, &transform))) return rc;
    const bool prof = profiling_now(ctx);
    P.clock_probe = prof ? ctx->d_clock : nullptr;
    P.out_a = (i32 *)ctx->ext_a.ptr; P.out_a_stride = ext;
    P.out_b = (i32 *)ctx->ext_b.ptr; P.out_b_stride = 1;
    hipEvent_t pe[3] = {nullptr, nullptr, nullptr};
    if (prof) {
        if ((rc = profile_slot(ctx, pe))) return rc;
        HIP_TRY(hipMemsetAsync(ctx->d_clock, 0, CLOCK_PROBE_WORDS * sizeof(unsigned long long), ctx->stream));
        HIP_TRY(hipEventRecord(pe[0], ctx->stream));
    }
    HIP_TRY(launch_bootstrap(P, transform, key->mask_size, effective_tuning(ctx), ctx->stream));
    if (prof) HIP_TRY(hipEventRecord(pe[1], ctx->stream));
    if (mux_bits > 0)
        HIP_TRY(launch_batch_mux_fold((i32 *)ctx->ext_a.ptr, (i32 *)ctx->ext_b.ptr, d_outs, (int)outs.size(), out_bits, ext, MU8,
                                      ctx->stream));
    nufhe_lwe none;
    memset(&none, 0, sizeof(none));
    rc = run_keyswitch(ctx, key, none, (const i32 *)ctx->ext_a.ptr, ext, (const i32 *)ctx->ext_b.ptr, 1, nullptr, 0, nullptr, 0, 0,
                       out_bits, d_outs, (int)outs.size());
    if (rc) return rc;
    if (prof) {
        HIP_TRY(hipEventRecord(pe[2], ctx->stream));
        profile_commit(ctx, pe);
    }
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_ctx_pin_scratch(nufhe_ctx *ctx, int delta)
{
    NUFHE_API_BEGIN
    // (called from GateGraph.__del__, i.e. at whatever moment garbage collection runs: the caller's current device is
    // left as found, and while the stream is being captured nothing is synchronised or freed -- the retired buffers wait
    // for the next non-capturing ensure() / pin call)
    if (!ctx) return fail(NUFHE_EINVAL, "null context");
    if (ctx->scratch_pins + delta < 0) return fail(NUFHE_EINVAL, "scratch pin count would become negative");
    ctx->scratch_pins += delta;
    if (ctx->scratch_pins != 0 || ctx->retired.empty()) return NUFHE_OK;
    int caller_device = -1;
    HIP_TRY(hipGetDevice(&caller_device));
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess && !stream_capturing(ctx)) free_retired_if_unpinned(ctx);
    if (caller_device >= 0) (void)hipSetDevice(caller_device);
    if (e != hipSuccess) return fail(NUFHE_EHIP, "nufhe_ctx_pin_scratch: %s", hipGetErrorString(e));
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_lwe_phase(nufhe_ctx *ctx, int32_t *d_out, long out_stride, const int32_t *d_a, long a_stride,
                    const int32_t *d_base, long base_stride, const int32_t *d_key, int32_t sign, long count, int size)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (count < 0 || size < 1) return fail(NUFHE_EINVAL, "bad sizes");
    if (sign != 1 && sign != -1) return fail(NUFHE_EINVAL, "sign must be +1 or -1");
    HIP_TRY(launch_lwe_phase(d_out, out_stride, d_a, a_stride, d_base, base_stride, d_key, sign, count, size, ctx->stream));
    return NUFHE_OK;
    NUFHE_API_END
}

// ---- per-kernel entry points -------------------------------------------------------------

int nufhe_t32_to_phase(nufhe_ctx *ctx, int32_t *d_result, const int32_t *d_phase, long count, uint32_t mspace_size)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (mspace_size == 0) return fail(NUFHE_EINVAL, "mspace_size must be positive");
    HIP_TRY(launch_t32_to_phase(d_result, d_phase, count, mspace_size, ctx->stream));
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_shift_torus_polynomial(nufhe_ctx *ctx, int32_t *d_result, const int32_t *d_source, const int32_t *d_powers,
                                 long powers_stride, long powers_idx, long batch, int polys, int minus_one,
                                 int invert_powers)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    HIP_TRY(launch_shift_tp(d_result, d_source, d_powers, powers_stride, powers_idx, batch, polys, minus_one,
                            invert_powers, ctx->stream));
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_tlwe_extract(nufhe_ctx *ctx, int32_t *d_result_a, int32_t *d_result_b, const int32_t *d_tlwe, long batch,
                       int mask_size)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (mask_size < 1) return fail(NUFHE_EINVAL, "mask_size must be positive");
    HIP_TRY(launch_tlwe_extract(d_result_a, d_result_b, d_tlwe, batch, mask_size, ctx->stream));
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_tgsw_decompose(nufhe_ctx *ctx, int32_t *d_result, const int32_t *d_sample, long polys)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (polys < 0) return fail(NUFHE_EINVAL, "negative polynomial count");
    if (polys && (!d_result || !d_sample)) return fail(NUFHE_EINVAL, "null argument");
    HIP_TRY(launch_tgsw_decompose(d_result, d_sample, polys, ctx->stream));
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_tgsw_mac(nufhe_ctx *ctx, uint64_t *d_result, const uint64_t *d_sample, const uint64_t *d_bk, int bk_len,
                   int bk_row, long batch, int mask_size)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (mask_size != 1 && mask_size != 2) return fail(NUFHE_EINVAL, "mask_size must be 1 or 2");
    if (batch < 0) return fail(NUFHE_EINVAL, "negative batch size");
    if (bk_row < 0 || bk_row >= bk_len) return fail(NUFHE_EINVAL, "bk_row %d out of range [0, %d)", bk_row, bk_len);
    if (batch && (!d_result || !d_sample || !d_bk)) return fail(NUFHE_EINVAL, "null argument");
    HIP_TRY(launch_tgsw_mac(d_result, d_sample, d_bk, bk_row, batch, mask_size, ctx->stream));
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_ntt_forward_i32(nufhe_ctx *ctx, uint64_t *d_out, const int32_t *d_in, long batch)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    HIP_TRY(launch_ntt_forward(d_out, d_in, 0, batch, ctx->d_tw1f, ctx->d_tw1i, ctx->stream));
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_ntt_forward_u64(nufhe_ctx *ctx, uint64_t *d_out, const uint64_t *d_in, long batch)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    HIP_TRY(launch_ntt_forward(d_out, d_in, 1, batch, ctx->d_tw1f, ctx->d_tw1i, ctx->stream));
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_ntt_inverse_i32(nufhe_ctx *ctx, int32_t *d_out, const uint64_t *d_in, long batch)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    HIP_TRY(launch_ntt_inverse(d_out, d_in, 0, batch, ctx->d_tw1f, ctx->d_tw1i, ctx->stream));
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_ntt_inverse_u64(nufhe_ctx *ctx, uint64_t *d_out, const uint64_t *d_in, long batch)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    HIP_TRY(launch_ntt_inverse(d_out, d_in, 1, batch, ctx->d_tw1f, ctx->d_tw1i, ctx->stream));
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_fft_forward_i32(nufhe_ctx *ctx, double *d_out, const int32_t *d_in, long batch)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    HIP_TRY(launch_fft_forward((cplx *)d_out, d_in, batch, ctx->d_ftw1, ctx->d_ftw2, ctx->stream));
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_fft_inverse_i32(nufhe_ctx *ctx, int32_t *d_out, const double *d_in, long batch)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    HIP_TRY(launch_fft_inverse(d_out, (const cplx *)d_in, batch, ctx->d_ftw1, ctx->d_ftw2, ctx->stream));
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_ff_op(nufhe_ctx *ctx, uint64_t *d_out, const uint64_t *d_a, const uint64_t *d_b, const uint64_t *d_c,
                const uint64_t *d_d, const uint64_t *d_e, int op, int shift, long count)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (op < 0 || op > 8 || shift < 0 || shift >= 192) return fail(NUFHE_EINVAL, "bad ff op %d / shift %d", op, shift);
    HIP_TRY(launch_ff_op(d_out, d_a, d_b, d_c, d_d, d_e, op, shift, count, ctx->stream));
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_l4_op(nufhe_ctx *ctx, uint32_t *d_out, uint32_t *d_out2, const uint32_t *d_a, const uint32_t *d_b,
                const uint32_t *d_c, int op, int shift, long count)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (op < 0 || op > 10 || shift < 0 || shift >= 192) return fail(NUFHE_EINVAL, "bad limb op %d / shift %d", op, shift);
    if (!d_out || !d_out2 || !d_a) return fail(NUFHE_EINVAL, "null operand");
    HIP_TRY(launch_l4_op(d_out, d_out2, d_a, d_b, d_c, op, shift, count, ctx->stream));
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_poly_mul_i32(nufhe_ctx *ctx, int32_t *d_out, const int32_t *d_x, const int32_t *d_y, long batch, long y_batch)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (y_batch < 1) return fail(NUFHE_EINVAL, "y_batch must be positive");
    HIP_TRY(launch_poly_mul(d_out, d_x, d_y, nullptr, 1024, batch, y_batch, ctx->d_tw1f, ctx->d_tw1i, ctx->stream));
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_external_mul(nufhe_ctx *ctx, nufhe_cloudkey *key, int32_t *d_accum, int bk_row, long batch)
{
    NUFHE_API_BEGIN
    int rc = check_key(ctx, key, true, false);
    if (rc) return rc;
    if (bk_row < 0 || bk_row >= key->n) return fail(NUFHE_EINVAL, "bk_row %d out of range", bk_row);
    if (uses_xfft(key)) {
        if ((rc = ensure_bkx(ctx, key)) || (rc = ensure(ctx, ctx->xfft_park, (size_t)batch * 2048 * sizeof(u32)))) return rc;
        HIP_TRY(launch_blind_rotate_accum_xfft(d_accum, key->d_bkx, nullptr, 0, bk_row, 1, 1, batch, ctx->d_ftw1, ctx->d_ftw2,
                                               (u32 *)ctx->xfft_park.ptr, key->mask_size, ctx->stream));
        return NUFHE_OK;
    }
    const bool fft = key->transform == NUFHE_TRANSFORM_FFT;
    HIP_TRY(launch_blind_rotate_accum(d_accum, key->d_bk, nullptr, 0, bk_row, 1, 1, batch,
                                      fft ? (const void *)ctx->d_ftw1 : (const void *)ctx->d_tw1x,
                                      fft ? (const void *)ctx->d_ftw2 : (const void *)ctx->d_tw1i, key->transform,
                                      key->mask_size, ctx->stream));
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_blind_rotate(nufhe_ctx *ctx, nufhe_cloudkey *key, int32_t *d_accum, const int32_t *d_bara, long bara_stride,
                       int n_rows, long batch)
{
    NUFHE_API_BEGIN
    int rc = check_key(ctx, key, true, false);
    if (rc) return rc;
    if (n_rows < 0 || n_rows > key->n) return fail(NUFHE_EINVAL, "n_rows %d out of range", n_rows);
    if (uses_xfft(key)) {
        if ((rc = ensure_bkx(ctx, key)) || (rc = ensure(ctx, ctx->xfft_park, (size_t)batch * 2048 * sizeof(u32)))) return rc;
        HIP_TRY(launch_blind_rotate_accum_xfft(d_accum, key->d_bkx, d_bara, bara_stride, 0, n_rows, 0, batch, ctx->d_ftw1,
                                               ctx->d_ftw2, (u32 *)ctx->xfft_park.ptr, key->mask_size, ctx->stream));
        return NUFHE_OK;
    }
    const bool fft = key->transform == NUFHE_TRANSFORM_FFT;
    HIP_TRY(launch_blind_rotate_accum(d_accum, key->d_bk, d_bara, bara_stride, 0, n_rows, 0, batch,
                                      fft ? (const void *)ctx->d_ftw1 : (const void *)ctx->d_tw1x,
                                      fft ? (const void *)ctx->d_ftw2 : (const void *)ctx->d_tw1i, key->transform,
                                      key->mask_size, ctx->stream));
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_tlwe_encrypt_zero(nufhe_ctx *ctx, int32_t *d_result_a, const int32_t *d_key, const int32_t *d_noises1,
                            const int32_t *d_noises2, long batch, int mask_size)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (mask_size < 1) return fail(NUFHE_EINVAL, "mask_size must be positive");
    const int k = mask_size, k1 = mask_size + 1;
    // mask polynomials: result[:, i, :] = noises1[:, i, :] (tlwe_cpu.py:81)
    HIP_TRY(hipMemcpy2DAsync(d_result_a, (size_t)k1 * 1024 * sizeof(i32), d_noises1, (size_t)k * 1024 * sizeof(i32),
                             (size_t)k * 1024 * sizeof(i32), (size_t)batch, hipMemcpyDeviceToDevice, ctx->stream));
    // body: result[:, k, :] = noises2 + sum_i noises1[:, i, :] * key[i] (tlwe_cpu.py:76-84)
    i32 *body = d_result_a + (size_t)k * 1024;
    for (int i = 0; i < k; i++)
        HIP_TRY(launch_poly_mul_strided(body, (long)k1 * 1024, d_noises1 + (size_t)i * 1024, (long)k * 1024,
                                        d_key + (size_t)i * 1024, i == 0 ? d_noises2 : body,
                                        i == 0 ? 1024L : (long)k1 * 1024, batch, ctx->d_tw1f, ctx->d_tw1i, ctx->stream));
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_ctx_set_team_max_bits(nufhe_ctx *ctx, long bits)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    ctx->team_max_bits = bits;
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_ctx_set_keyswitch_mfma(nufhe_ctx *ctx, int enable)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (enable < 0 || enable > 2) return fail(NUFHE_EINVAL, "keyswitch mode must be 0, 1 or 2");
    ctx->ks_mfma = enable;
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_ctx_set_pair_max_bits(nufhe_ctx *ctx, long bits)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    ctx->pair_max_bits = bits;
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_ctx_get_tuning(nufhe_ctx *ctx, nufhe_tuning *out)
{
    NUFHE_API_BEGIN
    if (!ctx || !out) return fail(NUFHE_EINVAL, "null argument");
    const BrTuning &T = ctx->tuning;
    memset(out, 0, sizeof(*out));
    out->team_max_bits = T.team_max_bits; out->team_max_bits_fft = T.team_max_bits_fft;
    out->pair_max_bits_ntt = T.pair_max_bits_ntt; out->pair_max_bits_fft = T.pair_max_bits_fft;
    out->ks_mfma_min_bits = T.ks_mfma_min_bits;
    out->ring_k2 = T.ring_k2; out->k2_roomy_ratio_pct = T.k2_roomy_ratio_pct;
    out->measured = T.measured; out->num_cus = T.num_cus;
    snprintf(out->arch_name, sizeof(out->arch_name), "%s", ctx->arch_name);
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_ctx_set_tuning(nufhe_ctx *ctx, const nufhe_tuning *in)
{
    NUFHE_API_BEGIN
    if (!ctx) return fail(NUFHE_EINVAL, "null context");
    if (!in) {                                   // NULL: back to what the device table says
        ctx->tuning = br_tuning_for(ctx->arch_name, ctx->num_cus);
        return NUFHE_OK;
    }
    if (in->team_max_bits < 0 || in->team_max_bits_fft < 0 || in->pair_max_bits_ntt < 0 || in->pair_max_bits_fft < 0 ||
        in->ks_mfma_min_bits < 0 || in->k2_roomy_ratio_pct <= 0)
        return fail(NUFHE_EINVAL, "negative switch point");
    BrTuning &T = ctx->tuning;
    T.team_max_bits = in->team_max_bits; T.team_max_bits_fft = in->team_max_bits_fft;
    T.pair_max_bits_ntt = in->pair_max_bits_ntt; T.pair_max_bits_fft = in->pair_max_bits_fft;
    T.ks_mfma_min_bits = in->ks_mfma_min_bits;
    T.ring_k2 = in->ring_k2 != 0; T.k2_roomy_ratio_pct = in->k2_roomy_ratio_pct;
    T.measured = 0;                              // (num_cus and arch_name are the device's, not the caller's)
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_ctx_set_team8(nufhe_ctx *ctx, int enable)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    ctx->team8 = enable != 0;
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_profile_enable(nufhe_ctx *ctx, int enable)
{
    NUFHE_API_BEGIN
    if (!ctx) return fail(NUFHE_EINVAL, "null context");
    ctx->profile = enable != 0;
    ctx->ev_valid = false;
    ctx->ring_count = 0;
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_profile_history(nufhe_ctx *ctx, float *blind_rotate_ms, float *keyswitch_ms, int capacity, int *count)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!count || capacity < 0 || (capacity > 0 && (!blind_rotate_ms || !keyswitch_ms))) return fail(NUFHE_EINVAL, "null argument");
    const long have = ctx->ring_count < PROFILE_RING ? ctx->ring_count : PROFILE_RING;
    const long n = have < capacity ? have : capacity;
    if (have > 0) HIP_TRY(hipEventSynchronize(ctx->ev[2]));          // the last gate's end: everything before it is complete
    for (long i = 0; i < n; i++) {
        const long slot = (ctx->ring_count - n + i) % PROFILE_RING;      // the n most recent, oldest first
        HIP_TRY(hipEventElapsedTime(&blind_rotate_ms[i], ctx->ring[3 * slot], ctx->ring[3 * slot + 1]));
        HIP_TRY(hipEventElapsedTime(&keyswitch_ms[i], ctx->ring[3 * slot + 1], ctx->ring[3 * slot + 2]));
    }
    *count = (int)n;
    ctx->ring_count = 0;
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_profile_last(nufhe_ctx *ctx, float *blind_rotate_ms, float *keyswitch_ms)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!ctx->ev_valid) return fail(NUFHE_EINVAL, "no profiled launch recorded");
    HIP_TRY(hipEventSynchronize(ctx->ev[2]));
    float br = 0, ks = 0;
    HIP_TRY(hipEventElapsedTime(&br, ctx->ev[0], ctx->ev[1]));
    HIP_TRY(hipEventElapsedTime(&ks, ctx->ev[1], ctx->ev[2]));
    if (blind_rotate_ms) *blind_rotate_ms = br;
    if (keyswitch_ms) *keyswitch_ms = ks;
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_profile_clock(nufhe_ctx *ctx, double *shader_ghz, double *wave_ms)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!ctx->ev_valid) return fail(NUFHE_EINVAL, "no profiled launch recorded");
    HIP_TRY(hipEventSynchronize(ctx->ev[2]));
    unsigned long long h[2] = {0, 0};
    HIP_TRY(hipMemcpy(h, ctx->d_clock, sizeof(h), hipMemcpyDeviceToHost));
    if (h[1] == 0) return fail(NUFHE_EINVAL, "the last profiled launch did not run a wave-per-bit kernel");
    if (shader_ghz) *shader_ghz = (double)h[0] / ((double)h[1] * 10.0);     // 100 MHz ticks = 10 ns
    if (wave_ms) *wave_ms = (double)h[1] * 1e-5;
    return NUFHE_OK;
    NUFHE_API_END
}

int nufhe_profile_waves(nufhe_ctx *ctx, double *start_ms, double *end_ms, int *simd, int capacity, int *count)
{
    NUFHE_API_BEGIN
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!ctx->ev_valid) return fail(NUFHE_EINVAL, "no profiled launch recorded");
    if (!start_ms || !end_ms || !simd || !count || capacity < 1) return fail(NUFHE_EINVAL, "null argument");
    HIP_TRY(hipEventSynchronize(ctx->ev[2]));
    unsigned long long h[CLOCK_PROBE_WORDS];
    HIP_TRY(hipMemcpy(h, ctx->d_clock, sizeof(h), hipMemcpyDeviceToHost));
    unsigned long long first = ~0ull;
    for (int w = 0; w < 8; w++)
        if (h[4 + 3 * w] && h[2 + 3 * w] < first) first = h[2 + 3 * w];
    int n = 0;
    for (int w = 0; w < 8 && n < capacity; w++) {
        if (!h[4 + 3 * w]) continue;
        start_ms[n] = (double)(h[2 + 3 * w] - first) * 1e-5;      // 100 MHz ticks
        end_ms[n] = (double)(h[3 + 3 * w] - first) * 1e-5;
        simd[n] = (int)h[4 + 3 * w] - 1;
        n++;
    }
    *count = n;
    if (n == 0) return fail(NUFHE_EINVAL, "the last profiled launch did not run a wave-per-bit kernel");
    return NUFHE_OK;
    NUFHE_API_END
}

}  // extern "C"
