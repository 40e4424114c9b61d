import os, sys, ctypes
import numpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import nufhe_amd
from nufhe_amd import _lib
from nufhe_amd.device import DeviceThread, ptr
from oracle import oracle as orc
import gpu_helpers as H
import golden_inputs as gi

thr = DeviceThread(0)
lwe_key, tlwe_key, ck = orc.make_key_pair(orc.DeterministicRNG(123))
cloud_key = H.cloud_key_from_arrays(thr, ck)
secret_key = H.secret_key_from_array(thr, lwe_key)
ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(456), thread=thr)
vm = ctx.make_virtual_machine(cloud_key)

for B in (2, 8, 9, 32, 64):
    rng = orc.DeterministicRNG(456)
    ms = [rng.uniform_bool((B,)).astype(bool) for _ in range(2)]
    cs = [orc.encrypt(rng, lwe_key, m) for m in ms]
    ds = [H.ciphertext_from_arrays(thr, c) for c in cs]
    r = vm.gate_nand(ds[0], ds[1])
    exp = orc.gate('gate_nand', ck, cs[0], cs[1])
    ra, rb, rcv = H.ct_arrays(r)
    bad_bits = numpy.where((ra != exp[0]).any(axis=1) | (rb != exp[1]))[0]
    print("NAND B=%d: mismatching bits %s; cv ok %s" % (B, bad_bits.tolist(), (rcv == exp[2]).all()))
    # bootstrap without keyswitch
    from nufhe_amd.bootstrap import bootstrap
    from nufhe_amd import lwe as L
    params = cloud_key.params
    MU = 2**29
    ta = (-cs[0][0] - cs[1][0]).astype(numpy.int32); tb = (numpy.int32(MU) - cs[0][1] - cs[1][1]).astype(numpy.int32)
    ea, eb = orc.bootstrap_extract(ck.bk, ta, tb, MU)
    tmp = H.ciphertext_from_arrays(thr, (ta, tb, numpy.zeros(B, numpy.float32)))
    ext = L.LweSampleArray.empty(thr, params.tgsw_params.tlwe_params.extracted_lweparams, (B,))
    bootstrap(thr, ext, cloud_key.bootstrap_key, cloud_key.keyswitch_key, MU, tmp, no_keyswitch=True)
    ga, gb = H.host(ext.a), H.host(ext.b)
    bad = numpy.where((ga != ea).any(axis=1) | (gb != eb))[0]
    print("   bootstrap(no KS): mismatching bits", bad.tolist())
    # keyswitch alone on the oracle's extracted samples
    ksr = L.LweSampleArray.empty(thr, params.in_out_params, (B,))
    src = H.ciphertext_from_arrays(thr, (ea, eb, numpy.zeros(B, numpy.float32)))
    L.lwe_keyswitch(thr, ksr, cloud_key.keyswitch_key, src)
    ka, kb, kcv = orc.lwe_keyswitch(ck.ks_a, ck.ks_b, ck.ks_cv, ea, eb)
    xa, xb, xcv = H.ct_arrays(ksr)
    bad = numpy.where((xa != ka).any(axis=1) | (xb != kb))[0]
    print("   keyswitch alone: mismatching bits", bad.tolist(), "mismatching columns (first bad bit):",
          (numpy.where(xa[bad[0]] != ka[bad[0]])[0][:10].tolist() if len(bad) else []))

# blind rotate hook
acc0, bk, bara = gi.blind_rotate_inputs()
from nufhe_amd.bootstrap import NativeCloudKey
native = NativeCloudKey(thr, bk.shape[0])
bkc = numpy.ascontiguousarray(bk, numpy.uint64)
_lib.call("nufhe_bk_upload_reference", native.handle, bkc.ctypes.data_as(ctypes.c_void_p))
acc = H.dev(thr, acc0)
_lib.call("nufhe_blind_rotate", thr.handle, native.handle, ptr(acc), ptr(H.dev(thr, bara)), bara.shape[1], bk.shape[0], acc0.shape[0])
exp = orc.blind_rotate(acc0, bk, bara)
got = H.host(acc)
print("blind_rotate hook: mismatches", int((got != exp).sum()), "of", exp.size)
for it in range(1, 4):
    acc = H.dev(thr, acc0)
    _lib.call("nufhe_blind_rotate", thr.handle, native.handle, ptr(acc), ptr(H.dev(thr, bara)), bara.shape[1], it, acc0.shape[0])
    exp = orc.blind_rotate(acc0, bk, bara, n_iter=it)
    print("  rows=%d mismatches %d" % (it, int((H.host(acc) != exp).sum())))
accum, bk2, row = gi.extmul_inputs()
native2 = NativeCloudKey(thr, bk2.shape[0])
bkc2 = numpy.ascontiguousarray(bk2, numpy.uint64)
_lib.call("nufhe_bk_upload_reference", native2.handle, bkc2.ctypes.data_as(ctypes.c_void_p))
a2 = H.dev(thr, accum)
_lib.call("nufhe_external_mul", thr.handle, native2.handle, ptr(a2), row, 6)
e2 = orc.tgsw_external_mul(accum, bk2, row)
g2 = H.host(a2)
print("external_mul hook: mismatches", int((g2 != e2).sum()), "of", e2.size, "per bit:", (g2 != e2).reshape(6, -1).sum(1).tolist())
