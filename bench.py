"""
bench.py -- bootstrapped-gate throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 10 --warmup 2            # config 2: NAND, 4096-bit batch
    python bench.py --gate mux                                # config 3
    python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8   # config 4 (weak scaling)

A "step" is one gate over one batch of synthetic ciphertexts that already live in HBM (keys and
ciphertexts are generated from fixed seeds before the timed region).  One process per GPU; the
batch shards over ranks as independent bits (no data-path collective); with N > 1 every step ends
with the result gather (RCCL all_gather of the output ciphertexts), as in the reference's
examples/multi_gpu.py.  Rank 0 prints ONE JSON line.
"""

import argparse
import json
import os
import sys
import time

import numpy
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes per bit, streaming model (BASELINE.md §3 / SURVEY §8d)
A_BK = 32_768_000
A_KS = 16_384_000 + 32_768
A_LWE = 2008
A_EXT = 4100
A_NAND = A_BK + A_KS + 3 * A_LWE                 # 49,190,792
A_MUX = 2 * A_BK + A_KS + 4 * A_LWE              # 81,960,800
A_BR_NAND = A_BK + 2 * A_LWE + A_EXT             # bootstrap kernel alone, one blind rotate per bit
HBM_PEAK_GBS = 8000.0                            # MI355X_MICROARCH.md: 8.0 TB/s spec


def pmc_traffic(transform, gate, bits):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this same command,
    corrected as MI355X_MICROARCH.md prescribes); None for configurations that were not measured."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            entry = json.load(f)[transform]
    except Exception:
        return None
    if gate != "nand" or bits != 4096:
        return None
    return entry["hbm_bytes_per_launch"]


def cpu_baseline(gate, sample_bits):
    """Times the CPU oracle (a C restatement of the reference's *_cpu.py composition, OpenMP over
    bits) on a bounded sample of the same workload, on this host's cores."""
    from oracle import oracle as orc
    lwe_key, tlwe_key, ck = orc.make_key_pair(orc.DeterministicRNG(123))
    rng = orc.DeterministicRNG(456)
    ms = [rng.uniform_bool((sample_bits,)).astype(bool) for _ in range(3)]
    cs = [orc.encrypt(rng, lwe_key, m) for m in ms]
    t0 = time.time()
    if gate == 'mux':
        r = orc.gate_mux(ck, cs[0], cs[1], cs[2])
        ok = (orc.decrypt(lwe_key, r) == numpy.where(ms[0], ms[1], ms[2])).all()
    else:
        r = orc.gate('gate_nand', ck, cs[0], cs[1])
        ok = (orc.decrypt(lwe_key, r) == ~(ms[0] & ms[1])).all()
    dt = time.time() - t0
    assert ok
    return dict(value=sample_bits / dt, unit="gates/s", cores=orc.num_threads(), kind="port",
                sample="%d-bit %s, full n=500 bootstrap + keyswitch, %.1f s" % (sample_bits, gate.upper(), dt),
                ms_per_bit=1000.0 * dt / sample_bits)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--gate", choices=["nand", "mux"], default="nand")
    ap.add_argument("--bits", type=int, default=4096, help="bits per GPU")
    ap.add_argument("--transform", choices=["NTT", "FFT"], default="NTT",
                    help="NTT = BASELINE configs 2-4 (bit-exact path); FFT = config 5 (fp64, tolerance path)")
    ap.add_argument("--cpu-sample-bits", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the secondary measurements (other gate / FFT transform) reported under 'other_configs'")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("NUFHE_BENCH_BACKEND", "nccl")   # "gloo": test-only, ranks may share a GPU
        if backend == "gloo":
            local_rank = local_rank % max(1, torch.cuda.device_count())
            torch.cuda.set_device(local_rank)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import nufhe_amd
    from nufhe_amd import _lib
    from nufhe_amd.device import DeviceThread

    thr = DeviceThread(local_rank)
    ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(123), thread=thr)
    secret_key, cloud_key = ctx.make_key_pair(transform_type=args.transform)   # replicated on every rank (98.6 MB)
    vm = ctx.make_virtual_machine(cloud_key)

    B = args.bits
    data_rng = numpy.random.RandomState(456 + rank)
    ms = [data_rng.randint(0, 2, size=(B,)).astype(bool) for _ in range(3)]
    ctx.rng = nufhe_amd.DeterministicRNG(1000 + rank)
    cs = [ctx.encrypt(secret_key, m) for m in ms]
    out = vm.empty_ciphertext((B,))
    from nufhe_amd import multi_gpu

    def step():
        if args.gate == "mux":
            vm.gate_mux(cs[0], cs[1], cs[2], dest=out)
        else:
            vm.gate_nand(cs[0], cs[1], dest=out)
        if world > 1:
            # the result gather of examples/multi_gpu.py, here one RCCL all_gather per array
            multi_gpu.gather_ciphertext(out, world * B)

    import ctypes
    lib = _lib.lib()
    for _ in range(args.warmup):
        step()
    lib.nufhe_profile_enable(thr.handle, 1)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    br_ms, ks_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        if rank == 0:
            # HIP-event timing of the kernels of this step (events recorded on the launch stream)
            a = ctypes.c_float(); b = ctypes.c_float()
            _lib.check(lib.nufhe_profile_last(thr.handle, ctypes.byref(a), ctypes.byref(b)))
            br_ms.append(a.value); ks_ms.append(b.value)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device=thr.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the result gather alone (reported separately; it is also part of every timed step)
    gather_ms = None
    if world > 1:
        multi_gpu.gather_ciphertext(out, world * B)
        torch.cuda.synchronize(); dist.barrier()
        t1 = time.perf_counter()
        for _ in range(5):
            multi_gpu.gather_ciphertext(out, world * B)
        torch.cuda.synchronize(); dist.barrier()
        gather_ms = 1e3 * (time.perf_counter() - t1) / 5

    # secondary measurements, OUTSIDE the timed region: the other BASELINE configurations on the same
    # ciphertexts (3 steps each after 1 warm-up); reported under "other_configs", never in "value"
    other = {}
    if not args.no_extra:
        def measure(fn, nsteps=3):
            fn()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t1 = time.perf_counter()
            for _ in range(nsteps):
                fn()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            return (time.perf_counter() - t1) / nsteps
        out2 = vm.empty_ciphertext((B,))
        other_gate = "nand" if args.gate == "mux" else "mux"
        if other_gate == "mux":
            dt = measure(lambda: vm.gate_mux(cs[0], cs[1], cs[2], dest=out2))
            ok = bool((ctx.decrypt(secret_key, out2) == numpy.where(ms[0], ms[1], ms[2])).all())
        else:
            dt = measure(lambda: vm.gate_nand(cs[0], cs[1], dest=out2))
            ok = bool((ctx.decrypt(secret_key, out2) == ~(ms[0] & ms[1])).all())
        other["gate_%s_%s" % (other_gate, args.transform)] = {
            "ms_per_step_per_gpu": 1e3 * dt, "ms_per_bit": 1e3 * dt / B, "gates_per_s_per_gpu": B / dt, "correct": ok}
        other_tr = "FFT" if args.transform == "NTT" else "NTT"
        ctx_o = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(123), thread=thr)
        sk_o, ck_o = ctx_o.make_key_pair(transform_type=other_tr)     # same secret key bits (same seed)
        vm_o = ctx_o.make_virtual_machine(ck_o)
        dt = measure(lambda: vm_o.gate_nand(cs[0], cs[1], dest=out2))
        ok = bool((ctx_o.decrypt(sk_o, out2) == ~(ms[0] & ms[1])).all())
        other["gate_nand_%s" % other_tr] = {
            "ms_per_step_per_gpu": 1e3 * dt, "ms_per_bit": 1e3 * dt / B, "gates_per_s_per_gpu": B / dt, "correct": ok}
        del vm_o, ck_o
        if world == 1:
            # the reference's own report (test/test_gates.py:62-75,252-314): 1 warm-up + 10 calls, each
            # bracketed by a synchronize, at sizes B and B/2 -> overall / scaled ms per bit, fixed overhead
            def ten_calls(nbits):
                a, b, c = cs[0][:nbits], cs[1][:nbits], cs[2][:nbits]
                dest = vm.empty_ciphertext((nbits,))
                fn = ((lambda: vm.gate_mux(a, b, c, dest=dest)) if args.gate == "mux"
                      else (lambda: vm.gate_nand(a, b, dest=dest)))
                fn()
                ts = []
                for _ in range(10):
                    thr.synchronize()
                    t1 = time.perf_counter()
                    fn()
                    thr.synchronize()
                    ts.append(time.perf_counter() - t1)
                ts = numpy.array(ts)
                return ts.mean(), ts.std(ddof=1) / numpy.sqrt(ts.size)
            (m1, e1), (m2, e2) = ten_calls(B), ten_calls(B // 2)
            other["reference_style_report"] = {
                "gate": args.gate, "transform": args.transform, "sizes": [B, B // 2], "calls": 10,
                "mean_ms": [1e3 * m1, 1e3 * m2], "stderr_ms": [1e3 * e1, 1e3 * e2],
                "overall_ms_per_bit": 1e3 * m1 / B,
                "scaled_ms_per_bit": 1e3 * (m1 - m2) / (B - B // 2),
                "fixed_overhead_ms": 1e3 * (m1 - (m1 - m2) / (B - B // 2) * B)}

    # correctness of what was timed: every decrypted bit equals the truth table
    dec = ctx.decrypt(secret_key, out)
    expect = numpy.where(ms[0], ms[1], ms[2]) if args.gate == "mux" else ~(ms[0] & ms[1])
    correct = bool((dec == expect).all())

    if rank == 0:
        total_bits = world * B * args.steps
        gates_per_s = total_bits / elapsed
        ms_per_step = 1000.0 * elapsed / args.steps
        n_rot = 2 if args.gate == "mux" else 1
        br_avg = float(numpy.mean(br_ms)); ks_avg = float(numpy.mean(ks_ms))
        a_kernel = A_BR_NAND * B * n_rot            # algorithmic bytes of one bootstrap-kernel launch
        achieved = a_kernel / (br_avg * 1e-3) / 1e9
        a_gate = (A_MUX if args.gate == "mux" else A_NAND) * B
        result = {
            "metric": "bootstrapped gates/sec (%s), %d-bit batch per GPU, %s, n=500 N=1024 k=1 l=2" % (
                args.gate.upper(), B, args.transform),
            "value": gates_per_s,
            "unit": "gates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "ms_per_bit": ms_per_step / (world * B),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("u64 mod 2^64-2^32+1 (NTT) / int32 torus" if args.transform == "NTT"
                      else "f64 complex (folded FFT-512) / int32 torus"),
            "data": "synthetic (seeded keys and ciphertexts, resident in HBM)",
            "correct": correct,
            "config": {"workload": "gate_%s, %d-bit batch per GPU (BASELINE config %s)" % (
                args.gate, B, "5" if args.transform == "FFT" else (
                    "3" if args.gate == "mux" else ("2" if world == 1 else "4"))),
                "bits_per_gpu": B, "transform": args.transform, "parallelism": "bits sharded over %d GPU(s), keys replicated" % world},
            "roofline": {
                "bound": "hbm", "kernel": "k_bootstrap%s (fused mod-switch + blind rotate + extract)" % (
                    "_fft" if args.transform == "FFT" else ""),
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(args.transform, args.gate, B),
                "kernel_ms": br_avg, "keyswitch_ms": ks_avg,
                "algorithmic_bytes_per_launch": a_kernel,
                "gate_streaming_model_GBs": a_gate / (ms_per_step * 1e-3) / 1e9,
            },
        }
        if args.transform == "NTT":
            # what actually binds K1: VALU issue.  Instruction count per bit-iteration from
            # the ISA of the loop body (confirmed by PMC SQ_INSTS_VALU), 4 cycles per wave64 instruction,
            # 1024 SIMDs, sustained clock from PMC SQ_WAVE_CYCLES (2.1 GHz).
            valu_per_iter, simds, clock = 14126, 1024, 2.1e9
            issued = B * n_rot * 500 * valu_per_iter
            result["roofline"]["valu_issue"] = {
                "instructions_per_launch": issued,
                "frac_of_simd_cycles": issued * 4 / (simds * clock * br_avg * 1e-3),
                "note": "wave64 VALU instructions x 4 cycles / (1024 SIMDs x 2.1 GHz x kernel time)"}
        if gather_ms is not None:
            result["gather_ms"] = gather_ms
            result["gather_bytes_per_rank"] = B * 2008
        if other:
            result["other_configs"] = other
        if world == 1 and not args.no_cpu_baseline:
            nthreads = os.cpu_count() or 1
            sample = args.cpu_sample_bits or max(16, min(256, 4 * nthreads))
            result["cpu_baseline"] = cpu_baseline(args.gate, sample)
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
