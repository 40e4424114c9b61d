/*
 * c_abi_gate.c -- the bootstrapped-gate hot path driven from plain C through include/nufhe_hip.h: no Python, no torch, no
 * HIP headers.  This is what a host in another language binds; tests/test_gpu_c_host.py builds it with
 * gcc, feeds it a cloud key in the REFERENCE's array formats and two encrypted batches, and compares the words it writes
 * with the CPU oracle.
 *
 *   gcc -std=c99 -O2 -I include examples/c_abi_gate.c -L nufhe_amd -lnufhe_hip -Wl,-rpath,$PWD/nufhe_amd -o c_abi_gate
 *   ./c_abi_gate in.bin out.bin
 *
 * in.bin  : int32 n, int32 nbits, int32 transform (0 NTT / 1 FFT), int32 gate (0 NAND, 1 MUX, 2 = ONE nufhe_gate_batch call:
 *           NAND on the first half of the bits and MUX on the second half, two jobs of different kinds in one launch)
 *           bootstrapping key, reference format: 8 bytes x [n][2][2][2][1024]   (uint64 NTT / complex128 [..][512] FFT)
 *           keyswitch key, reference format: int32 a[1024][8][4][n], int32 b[1024][8][4], float cv[1024][8][4]
 *           three ciphertext batches: int32 a[nbits][n], int32 b[nbits]  (the third is only read by MUX)
 * out.bin : int32 a[nbits][n], int32 b[nbits], float cv[nbits]
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#include "nufhe_hip.h"

#define CHECK(call)                                                                    \
    do {                                                                               \
        int rc_ = (call);                                                              \
        if (rc_ != NUFHE_OK) {                                                         \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, nufhe_last_error());         \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

static void *read_block(FILE *f, size_t bytes)
{
    void *p = malloc(bytes ? bytes : 1);
    if (!p || fread(p, 1, bytes, f) != bytes) {
        fprintf(stderr, "short read (%zu bytes)\n", bytes);
        exit(2);
    }
    return p;
}

int main(int argc, char **argv)
{
    if (argc != 3) {
        fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]);
        return 2;
    }
    if (nufhe_abi_version() != NUFHE_ABI_VERSION) {
        fprintf(stderr, "library ABI %d, header ABI %d\n", nufhe_abi_version(), NUFHE_ABI_VERSION);
        return 2;
    }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int32_t head[4];
    if (fread(head, sizeof(head), 1, f) != 1) return 2;
    const int n = head[0], transform = head[2], gate = head[3];
    const long nbits = head[1];
    const size_t bk_bytes = (size_t)n * 8 * 1024 * 8, rows = (size_t)1024 * 8 * 4;
    void *bk = read_block(f, bk_bytes);
    int32_t *ks_a = read_block(f, rows * n * 4);
    int32_t *ks_b = read_block(f, rows * 4);
    float *ks_cv = read_block(f, rows * 4);
    int32_t *in_a[3], *in_b[3];
    for (int i = 0; i < 3; i++) {
        in_a[i] = read_block(f, (size_t)nbits * n * 4);
        in_b[i] = read_block(f, (size_t)nbits * 4);
    }
    fclose(f);

    nufhe_ctx *ctx = NULL;
    nufhe_cloudkey *key = NULL;
    CHECK(nufhe_ctx_create(0, NULL, 1 /* a stream of the library's own */, &ctx));
    CHECK(nufhe_cloudkey_create(ctx, n, transform, 1, &key));
    CHECK(nufhe_bk_upload_reference(key, bk));
    CHECK(nufhe_ks_upload(key, ks_a, ks_b, ks_cv));

    /* device ciphertexts: three inputs and the result, each a | b | cv in allocations of their own */
    nufhe_lwe ct[4];
    for (int i = 0; i < 4; i++) {
        void *a, *b, *cv;
        CHECK(nufhe_alloc(ctx, (size_t)nbits * n * 4 + 4, &a));
        CHECK(nufhe_alloc(ctx, (size_t)nbits * 4 + 4, &b));
        CHECK(nufhe_alloc(ctx, (size_t)nbits * 4 + 4, &cv));
        ct[i].a = a; ct[i].b = b; ct[i].cv = cv;
        ct[i].a_stride = n; ct[i].b_stride = 1; ct[i].size = n;
        if (i < 3 && nbits) {
            CHECK(nufhe_h2d(ctx, a, in_a[i], (size_t)nbits * n * 4));
            CHECK(nufhe_h2d(ctx, b, in_b[i], (size_t)nbits * 4));
        }
    }
    const int32_t MU = (int32_t)1 << 29;
    if (gate == 0)
        CHECK(nufhe_gate_binary(ctx, key, ct[3], ct[0], ct[1], MU, -1, -1, MU, nbits));     /* NAND, gates.py:81-121 */
    else if (gate == 1)
        CHECK(nufhe_gate_mux(ctx, key, ct[3], ct[0], ct[1], ct[2], nbits));                  /* a ? b : c, gates.py:600-664 */
    else {
        /* two independent gates of different kinds as one launch: views of the same buffers, shifted to each job's bits */
        const long half = nbits / 2;
        nufhe_gate_job jobs[2];
        nufhe_lwe v[4];
        for (int i = 0; i < 4; i++) {
            v[i] = ct[i];
            v[i].a += half * n; v[i].b += half; v[i].cv += half;
        }
        jobs[0].kind = NUFHE_JOB_BINARY; jobs[0].c0 = MU; jobs[0].pa = -1; jobs[0].pb = -1; jobs[0].nbits = half;
        jobs[0].result = ct[3]; jobs[0].a = ct[0]; jobs[0].b = ct[1]; jobs[0].c = ct[2];
        jobs[1].kind = NUFHE_JOB_MUX; jobs[1].c0 = 0; jobs[1].pa = 0; jobs[1].pb = 0; jobs[1].nbits = nbits - half;
        jobs[1].result = v[3]; jobs[1].a = v[0]; jobs[1].b = v[1]; jobs[1].c = v[2];
        CHECK(nufhe_gate_batch(ctx, key, jobs, 2, MU));
        nufhe_tuning t;
        CHECK(nufhe_ctx_get_tuning(ctx, &t));
        printf("switch points of %s (%d CUs, measured %d): team %ld, pair %ld, matrix-core keyswitch above %ld bits\n",
               t.arch_name, (int)t.num_cus, (int)t.measured, t.team_max_bits, t.pair_max_bits_ntt, t.ks_mfma_min_bits);
    }
    /* a second, chained gate on the device results: NOT(result) = lwe_negate, then back (exercises nufhe_lwe_linear) */
    CHECK(nufhe_lwe_linear(ctx, ct[0], ct[3], -1, 0, nbits, n));
    CHECK(nufhe_lwe_linear(ctx, ct[3], ct[0], -1, 0, nbits, n));
    CHECK(nufhe_ctx_synchronize(ctx));

    int32_t *out_a = malloc((size_t)nbits * n * 4 + 4), *out_b = malloc((size_t)nbits * 4 + 4);
    float *out_cv = malloc((size_t)nbits * 4 + 4);
    if (nbits) {
        CHECK(nufhe_d2h(ctx, out_a, ct[3].a, (size_t)nbits * n * 4));
        CHECK(nufhe_d2h(ctx, out_b, ct[3].b, (size_t)nbits * 4));
        CHECK(nufhe_d2h(ctx, out_cv, ct[3].cv, (size_t)nbits * 4));
    }
    f = fopen(argv[2], "wb");
    if (!f) { perror(argv[2]); return 2; }
    fwrite(out_a, 4, (size_t)nbits * n, f);
    fwrite(out_b, 4, (size_t)nbits, f);
    fwrite(out_cv, 4, (size_t)nbits, f);
    fclose(f);

    /* a malformed call is refused with a message, not a crash: result descriptor with the wrong LWE size */
    nufhe_lwe bad = ct[3];
    bad.size = n + 1;
    if (nufhe_gate_binary(ctx, key, bad, ct[0], ct[1], MU, -1, -1, MU, nbits) != NUFHE_EINVAL) {
        fprintf(stderr, "a wrong descriptor was accepted\n");
        return 1;
    }
    for (int i = 0; i < 4; i++) {
        CHECK(nufhe_free(ctx, ct[i].a));
        CHECK(nufhe_free(ctx, ct[i].b));
        CHECK(nufhe_free(ctx, ct[i].cv));
    }
    CHECK(nufhe_cloudkey_destroy(key));
    CHECK(nufhe_ctx_destroy(ctx));
    printf("c_abi_gate OK: %s on %ld bits through %s (refused the malformed call: %s)\n",
           gate == 0 ? "NAND" : gate == 1 ? "MUX" : "NAND | MUX batch", nbits,
           nufhe_version(), nufhe_last_error());
    return 0;
}
