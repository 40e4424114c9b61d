"""circuit_batch.py -- what the heterogeneous gate batch (VirtualMachine.gate_batch / nufhe_gate_batch) buys on
width-parallel circuits: independent small gates that, gate by gate, each cost a full 500-step blind rotation while
occupying a fraction of the chip (VERDICT r4 "missing" item 3; the reference's circuit is the gate-by-gate comparator of
nufhe/operators_integer.py:64-95).

  alu4        four DIFFERENT bitwise operations (AND, OR, XOR, NAND) of the same two 64-bit operands: 4 calls vs 1 batch
  mixed3      a MUX, an XNOR and an ANDNY on unrelated operands of 48 / 100 / 17 bits: 3 calls vs 1 batch
  comparators four independent uint_min circuits on (4, 16)-bit operands (18 dependent gates each): one after the other
              (the reference's schedule, per circuit) vs in lock step, every step of the four circuits as ONE batch

Median of 5 timed runs each, results compared word for word.  Prints one JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
import nufhe_amd as nufhe
from nufhe_amd.operators_integer import uint_min, uint_min_many, uintarray_to_bitarray, bitarray_to_uintarray

ctx = nufhe.Context(rng=nufhe.DeterministicRNG(123))
secret, cloud = ctx.make_key_pair()
if len(sys.argv) > 1:
    cloud.set_engine(sys.argv[1])          # `python tools/circuit_batch.py exact-fft`
vm = ctx.make_virtual_machine(cloud)
thr = ctx.thread
rs = numpy.random.RandomState(7)


def enc(shape):
    return ctx.encrypt(secret, rs.randint(0, 2, shape).astype(bool))


def timed(fn, reps=5):
    fn(); thr.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); out = fn(); thr.synchronize(); ts.append(1e3 * (time.perf_counter() - t))
    return round(sorted(ts)[reps // 2], 3), out


out = {}

# ---- alu4
a, b = enc(64), enc(64)
names = ['gate_and', 'gate_or', 'gate_xor', 'gate_nand']
t_seq, r_seq = timed(lambda: [getattr(vm, n)(a, b) for n in names])
t_bat, r_bat = timed(lambda: vm.gate_batch([(n, a, b) for n in names]))
assert all(x == y for x, y in zip(r_seq, r_bat))
out['alu4_64bit'] = {'gate_by_gate_ms': t_seq, 'one_batch_ms': t_bat, 'speedup': round(t_seq / t_bat, 2)}

# ---- mixed3
s, x, y = enc(48), enc(48), enc(48)
p, q = enc(100), enc(100)
u, v = enc(17), enc(17)
t_seq, r_seq = timed(lambda: [vm.gate_mux(s, x, y), vm.gate_xnor(p, q), vm.gate_andny(u, v)])
t_bat, r_bat = timed(lambda: vm.gate_batch([('gate_mux', s, x, y), ('gate_xnor', p, q), ('gate_andny', u, v)]))
assert all(x_ == y_ for x_, y_ in zip(r_seq, r_bat))
out['mixed3_mux48_xnor100_andny17'] = {'gate_by_gate_ms': t_seq, 'one_batch_ms': t_bat, 'speedup': round(t_seq / t_bat, 2)}

# ---- four independent comparators
C, M, W = 4, 4, 16
nums = [(rs.randint(0, 2 ** W, M).astype(numpy.uint16), rs.randint(0, 2 ** W, M).astype(numpy.uint16)) for _ in range(C)]
cts = [(ctx.encrypt(secret, uintarray_to_bitarray(p_)), ctx.encrypt(secret, uintarray_to_bitarray(q_))) for p_, q_ in nums]


def one_after_the_other():
    res = []
    for ca, cb in cts:
        ans = vm.empty_ciphertext((M, W))
        uint_min(thr, cloud, ans, ca, cb)
        res.append(ans)
    return res


def lock_step():
    # the schedule of uint_min, the same step of all C circuits as one batch (nufhe_amd.operators_integer.uint_min_many)
    res = [vm.empty_ciphertext((M, W)) for _ in range(C)]
    uint_min_many(thr, cloud, res, [ca for ca, _ in cts], [cb for _, cb in cts])
    return res


t_seq, r_seq = timed(one_after_the_other, reps=3)
t_bat, r_bat = timed(lock_step, reps=3)
for k in range(C):
    assert r_seq[k] == r_bat[k]
    assert (bitarray_to_uintarray(ctx.decrypt(secret, r_bat[k])) == numpy.minimum(*nums[k])).all()
out['four_uint_min_4x16'] = {'one_after_the_other_ms': t_seq, 'lock_step_batches_ms': t_bat, 'speedup': round(t_seq / t_bat, 2),
                             'gates_per_circuit': W + 2}
print(json.dumps(out))
