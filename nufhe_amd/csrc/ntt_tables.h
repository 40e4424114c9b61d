// ntt_tables.h -- host-side construction of the twiddle tables used by ntt1024.h.
// Replaces the host setup of the reference's NTT (nufhe/transform/ntt.py:63-120, gen_twiddle_ref):
// same root of unity (nufhe/transform/ntt_cpu.py:96-109), different factorisation, so only ONE
// 1024-entry table per direction (the "twiddle 1" layer) is needed.
#pragma once
#include "ff.h"

#define NTT_ROOT_2_32 0xa70dc47e4cbdf43fULL /* 2^32-th root of unity, ntt_cpu.py:109 */

static inline u64 ntt_host_pow(u64 x, u64 e)
{
    u64 r = 1;
    while (e) {
        if (e & 1) r = ff_mul(r, x);
        x = ff_mul(x, x);
        e >>= 1;
    }
    return r;
}

// tw1f[k2 * 64 + j1] = psi^((2 k2 + 1) j1);  tw1i[k2 * 64 + j1] = psi^(-(2 k2 + 1) j1) / 1024
static inline void ntt_make_tables(u64 *tw1f, u64 *tw1i)
{
    const u64 psi = ntt_host_pow(NTT_ROOT_2_32, (1ULL << 32) / 2048);
    const u64 psi_inv = ntt_host_pow(psi, FF_P - 2);
    const u64 n_inv = ntt_host_pow(1024, FF_P - 2);
    for (int k2 = 0; k2 < 16; k2++)
        for (int j1 = 0; j1 < 64; j1++) {
            const u64 e = (u64)(2 * k2 + 1) * (u64)j1;
            tw1f[k2 * 64 + j1] = ntt_host_pow(psi, e);
            tw1i[k2 * 64 + j1] = ff_mul(ntt_host_pow(psi_inv, e), n_inv);
        }
}

// Forward table as the limb-form transform reads it (ntt1024_l4.h): after exchange 1 lane (k2, q) holds
// j1 = q + 4 r in register r, so tw1x[r * 64 + lane] = tw1f[k2 * 64 + q + 4 r]
static inline void ntt_make_tw1x(u64 *tw1x, const u64 *tw1f)
{
    for (int r = 0; r < 16; r++)
        for (int lane = 0; lane < 64; lane++) tw1x[r * 64 + lane] = tw1f[(lane >> 2) * 64 + (lane & 3) + 4 * r];
}
