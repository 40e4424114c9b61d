"""
bench.py -- bootstrapped-gate throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 10 --warmup 2            # config 2: NAND, 4096-bit batch
    python bench.py --gate mux                                # config 3
    python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8   # config 4 (weak scaling)

A "step" is one gate over one batch of synthetic ciphertexts that already live in HBM (keys and
ciphertexts are generated from fixed seeds before the timed region).  One process per GPU; the
batch shards over ranks as independent bits (no data-path collective); every step launched through
torch.distributed.run (also with ONE rank: that is how the RCCL route is exercised on a 1-GPU box) ends
with the result gather to rank 0 (RCCL over xGMI), as in the reference's examples/multi_gpu.py.
Rank 0 prints ONE JSON line.
"""

import argparse
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (the host driver has no legacy IPC): RCCL would fail with
# `hipIpcGetMemHandle: invalid argument` without it; must be in the environment before the HIP runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

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


def _profile_json(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except Exception:
        return None


def pmc_traffic(transform, gate, bits):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this same command,
    corrected as MI355X_MICROARCH.md prescribes: NAND at 4096 bits from tools/profile.sh, MUX at 4096 and NAND at 2048
    bits from tools/pmc_traffic_configs.sh); None for configurations that were not measured."""
    table = _profile_json("pmc_traffic.json") or {}
    entry = (table.get("configs") or {}).get("%s/%s/%d" % (transform, gate, bits))      # tools/pmc_traffic_configs.sh
    if entry is None and gate == "nand" and bits == 4096:
        entry = table.get(transform)                                                     # tools/profile.sh
    return None if entry is None else entry["hbm_bytes_per_launch"]


FP64_VECTOR_PEAK_TFLOPS = 78.6      # MI355X vector fp64: 256 CUs x 128 flop/clk x 2.4 GHz (v_fma_f64 measured
                                    # at 4.1-4.4 cycles per wave64 instruction: profiles/r03_microbench_issue.txt)
MFMA_I8_PEAK_TOPS = 5000.0          # dense int8 / fp8 matrix peak (MI355X_MICROARCH.md: ~5 PF dense)
SIMDS = 1024                        # 256 CUs x 4

# arithmetic one blind-rotate iteration of ONE bit needs, whatever the representation (k = 1, l = 2, N = 1024):
# 6 negacyclic NTT-1024 = 6 x 10 stages x 512 butterflies, a butterfly = 1 add + 1 sub in GF(P); general
# multiplications: 6 x 1024 (the one table layer per transform) + 8 x 1024 (the key products)
ALG_ADDSUB_PER_BIT_ITER = 6 * 10 * 512 * 2         # 61,440
ALG_MODMUL_PER_BIT_ITER = 6 * 1024 + 8 * 1024      # 14,336


# `roofline.frac` is BASELINE's HBM axis (streaming model) since round 5.  `roofline.valu_issue.frac` is FROZEN at the
# round-3 definition ("vs_measured_class_rates": issued instructions x the best rate each issue class reaches on this chip
# at any occupancy, over the SIMD cycles of the launch) so that rounds compare -- it was the headline `frac`
# of rounds 3 and 4 --; the nominal-2-cycle fraction rides beside it under `valu_issue.fractions.vs_nominal_2_cycle_issue`.
FRAC_DEFINITION = "r03"


def issue_roofline(transform, bits, rotations, n_iter, kernel_ms, live_clock_ghz=None):
    """What bounds the bootstrap kernel: VALU issue.  Everything comes from TRACKED files:
    profiles/isa_mix.json (tools/isa_mix.py: instructions per blind-rotate iteration by issue class, counted in the
    ISA of the loop), profiles/valu_issue_costs.json (tools/microbench_issue -> tools/valu_issue_costs.py: cycles per
    class as a function of occupancy), profiles/pmc_<transform>.json (tools/profile.sh), and the live HIP-event
    duration / in-kernel clock of the timed launches.  Every fraction names its denominator."""
    mix = _profile_json("isa_mix.json")
    costs = _profile_json("valu_issue_costs.json")
    pmc = _profile_json("pmc_%s.json" % transform)
    if mix is None:
        return None
    k = mix.get({"NTT": "k_bootstrap<1>", "FFT": "k_bootstrap_fft", "XFFT": "k_bootstrap_xfft"}[transform])
    if k is None:
        return None
    iters = bits * rotations * n_iter
    out = {"isa_mix_per_iteration": {c: k.get(c, 0) for c in (
        "valu", "valu_plain", "valu_other", "valu_f64", "s_nop", "lds", "vmem", "scratch")},
        "vgprs": k.get("vgprs"), "scratch_bytes": k.get("scratch_bytes")}
    clock = None
    if pmc is not None:
        d = pmc.get("derived", {})
        clock = d.get("grbm_gui_active_per_xcd_ghz") or d.get("shader_clock_ghz_from_wave_cycles")
        out["pmc"] = {"valu_instructions_per_wave_iteration": (d.get("valu_instructions_per_wave") or 0) / n_iter,
                      "sustained_clock_ghz": clock,
                      "frac_wave_cycles_issuing": d.get("frac_SQ_ACTIVE_INST_ANY"),
                      "frac_wave_cycles_waiting_for_issue": d.get("frac_SQ_WAIT_INST_ANY"),
                      "frac_wave_cycles_at_waitcnt": d.get("frac_SQ_WAIT_ANY"),
                      "note": "rocprofv3 --pmc passes of tools/profile.sh on a separate run of this workload "
                              "(profiles/pmc_*.json, see profiles/README.md), not of this process; the instruction count is "
                              "re-derived from the current ISA above"}
    if live_clock_ghz:
        out["clock_ghz_in_kernel"] = live_clock_ghz     # s_memtime / s_memrealtime of one wave inside the timed launches
        clock = live_clock_ghz
    if costs is None or not clock:
        return out
    avail = SIMDS * clock * 1e9 * kernel_ms * 1e-3            # SIMD cycles the launch had
    per_iter = avail / iters                                  # SIMD cycles spent per bit-iteration
    n_plain, n_other, n_valu = k.get("valu_plain", 0), k.get("valu_other", 0), k.get("valu", 0)
    cyc_machine = n_plain * costs["machine_plain_cycles"] + n_other * costs["machine_other_cycles"]
    cyc_nominal = n_valu * costs["nominal_cycles_microarch_guide"]
    cyc_slots = n_valu * costs["two_wave_other_cycles"]
    out["simd_cycles_per_bit_iteration"] = per_iter
    out["cycles_per_valu_instruction_per_simd"] = per_iter / n_valu
    fracs = {
        "vs_measured_class_rates": {
            "frac": cyc_machine / per_iter, "cycles_per_bit_iteration": cyc_machine,
            "denominator": "SIMD cycles of the launch (1024 SIMDs x in-kernel clock x kernel time)",
            "numerator": "issued instructions x the best rate each class reaches on this chip at ANY occupancy "
                         "(plain VOP1/VOP2 %.2f cycles -- two waves sharing a slot --, everything else %.2f; "
                         "profiles/valu_issue_costs.json): %d x %.2f + %d x %.2f" % (
                             costs["machine_plain_cycles"], costs["machine_other_cycles"], n_plain,
                             costs["machine_plain_cycles"], n_other, costs["machine_other_cycles"])},
        "vs_nominal_2_cycle_issue": {
            "frac": cyc_nominal / per_iter, "cycles_per_bit_iteration": cyc_nominal,
            "denominator": "SIMD cycles of the launch",
            "numerator": "issued instructions x the 2 cycles per wave64 instruction MI355X_MICROARCH.md quotes for SIMD-32 "
                         "(measured: only the plain class reaches it, and only paired across two waves)"},
        "vs_one_slot_per_instruction": {
            "frac": cyc_slots / per_iter, "cycles_per_bit_iteration": cyc_slots,
            "denominator": "SIMD cycles of the launch",
            "numerator": "issued instructions x %.2f cycles = one issue slot each, what a MIXED stream pays at 2 waves per SIMD "
                         "(blend lines of the microbenchmark); above 1 means plain instructions of the two waves shared "
                         "slots" % costs["two_wave_other_cycles"]},
    }
    if transform == "NTT":
        # minimal-instruction arithmetic: a GF(P) add / sub = 4 plain limb instructions at the paired rate; a modular
        # multiplication = 4 v_mad_u64_u32 + 4 instructions of reduction / split at the "other" rate
        alg = (ALG_ADDSUB_PER_BIT_ITER * 4 * costs["machine_plain_cycles"]
               + ALG_MODMUL_PER_BIT_ITER * 8 * costs["machine_other_cycles"]) / 64.0
        fracs["algorithmic"] = {
            "frac": alg / per_iter, "cycles_per_bit_iteration": alg,
            "denominator": "SIMD cycles of the launch",
            "numerator": "per bit-iteration %d field add/sub x 4 plain instructions + %d field multiplications x 8 instructions "
                         "(4 v_mad_u64_u32 + 4 to reduce / split), per lane (/ 64), at the machine rates: the arithmetic "
                         "the transforms cannot avoid; the rest of the issued cycles is representation overhead (packing to "
                         "64-bit words, limb splits, per-lane shifts, zero-extension moves) and unshared issue slots" % (
                             ALG_ADDSUB_PER_BIT_ITER, ALG_MODMUL_PER_BIT_ITER)}
        out.update({"bound": "valu-issue", "achieved": cyc_machine * iters / (kernel_ms * 1e-3) / 1e9,
                    "peak": SIMDS * clock, "unit": "G SIMD issue cycles/s",
                    "frac": fracs["vs_measured_class_rates"]["frac"],
                    "frac_is": "vs_measured_class_rates", "frac_definition": FRAC_DEFINITION})
    else:
        flops = iters * 64.0 * k.get("f64_flops_per_lane", 0)
        fracs["fp64_fma_peak"] = {
            "frac": flops / (kernel_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
            "denominator": "vector fp64 peak %.1f TFLOP/s (fma = 2 flops per lane)" % FP64_VECTOR_PEAK_TFLOPS,
            "numerator": "fp64 flops issued: 64 lanes x %d per bit-iteration; three quarters of the fp64 instructions are "
                         "add / sub / mul (1 flop), so 0.5 is the most this instruction mix can show" % k.get("f64_flops_per_lane", 0)}
        out.update({"bound": "valu-issue (fp64)", "achieved": cyc_machine * iters / (kernel_ms * 1e-3) / 1e9,
                    "peak": SIMDS * clock, "unit": "G SIMD issue cycles/s",
                    "frac": fracs["vs_measured_class_rates"]["frac"], "frac_is": "vs_measured_class_rates",
                    "frac_definition": FRAC_DEFINITION,
                    "fp64_tflops": flops / (kernel_ms * 1e-3) / 1e12})
    out["fractions"] = fracs
    return out


def keyswitch_roofline(bits, ks_ms, mfma):
    """K2 (keyswitch stage = digit transpose + matrix-core product + finalize, HIP events around the three kernels)."""
    if not ks_ms:
        return None
    if not mfma:
        return {"kernel": "k_keyswitch_a (LDS row window; batches <= 2 x CUs bits) + finalize", "stage_ms": ks_ms,
                "bound": "lds", "note": "small-batch kernel: no matrix-core line"}
    ops = 2.0 * bits * 1024 * 8 * 4 * 500 * 4        # one-hot K = 1024 x 8 x 4 slots, 500 columns, 4 byte planes; MAC = 2
    return {"kernel": "k_ks_digits_t + k_keyswitch_mfma (v_mfma_i32_16x16x64_i8, one-hot A) + k_keyswitch_finalize",
            "stage_ms": ks_ms, "bound": "mfma", "achieved": ops / (ks_ms * 1e-3) / 1e12, "peak": MFMA_I8_PEAK_TOPS,
            "unit": "TOP/s (int8)", "frac": ops / (ks_ms * 1e-3) / 1e12 / MFMA_I8_PEAK_TOPS,
            "ops_per_launch": ops,
            "note": "issued int8 multiply-adds of the one-hot product (3 of 4 K-slots multiply zeros by construction) over "
                    "the WHOLE keyswitch stage incl. the pre-pass and finalize; the MFMA kernel alone is ~70 % of the stage "
                    "(profiles/*_kernel_stats.csv)"}


def cpu_baseline_and_parity(gate, sample_bits, cs_host, gpu_out):
    """Runs the CPU oracle (a C restatement of the reference's *_cpu.py composition, OpenMP over bits)
    on the first `sample_bits` bits of the SAME ciphertexts the GPU just processed, on this host's
    cores: its wall time is the reported CPU baseline, its output words are the parity check.

    cs_host: three (a, b, cv) host triples; gpu_out: {label: (gate, exact, (a, b, cv))} host arrays of
    the GPU results to compare (NTT path: every word must match; FFT path: stated tolerance).
    The oracle regenerates the cloud key from the bench's key seed (GPU key generation == oracle key
    generation from one seed: tests/test_gpu_kernels.py::test_gpu_keygen_matches_oracle_keygen)."""
    from oracle import oracle as orc
    lwe_key, tlwe_key, ck = orc.make_key_pair(orc.DeterministicRNG(123))
    n = sample_bits
    cs = [tuple(x[:n] for x in c) for c in cs_host]
    results, times = {}, {}
    needed = {g for g, _, _ in gpu_out.values()} | {gate}
    for g in sorted(needed):
        t0 = time.time()
        results[g] = orc.gate_mux(ck, cs[0], cs[1], cs[2]) if g == 'mux' else orc.gate('gate_nand', ck, cs[0], cs[1])
        times[g] = time.time() - t0
    parity = {}
    for label, (g, exact, arrs) in gpu_out.items():
        ref = results[g]
        da = (arrs[0][:n].astype(numpy.int64) - ref[0].astype(numpy.int64) + 2**31) % 2**32 - 2**31
        db = (arrs[1][:n].astype(numpy.int64) - ref[1].astype(numpy.int64) + 2**31) % 2**32 - 2**31
        differing = int((da != 0).sum() + (db != 0).sum())
        entry = {"bits": n, "words": int(da.size + db.size), "differing": differing,
                 "variances_differing": int((arrs[2][:n] != ref[2]).sum())}
        if not exact:
            entry["max_abs_diff_lsb"] = int(max(abs(da).max(), abs(db).max()))
            entry["tolerance_lsb"] = 16
        parity[label] = entry
    dt = times[gate]
    base = dict(value=n / dt, unit="gates/s", cores=orc.num_threads(), kind="port",
                sample="first %d bits of the benchmarked %s batch, full n=500 bootstrap + keyswitch, %.1f s" % (
                    n, gate.upper(), dt),
                ms_per_bit=1000.0 * dt / n,
                where="this host, in this run (the C restatement of the reference's *_cpu.py composition, OpenMP over bits)")
    # north_star's other baseline: the reference's OWN Python CPU functions.  They cannot run on the GPU box (the
    # reference is not shipped there), so the figure is the one measured in the build container, carried with its
    # provenance (tools/time_reference_cpu.py -> profiles/reference_python_cpu_timing.json); it was NOT timed in this run
    ref_py = reference_python_entry(gate)
    if ref_py is not None:
        base["reference_python"] = ref_py
    return base, parity


def reference_python_entry(gate):
    ref = _profile_json("reference_python_cpu_timing.json")
    if ref is None:
        return None
    mult = 2 if gate == "mux" else 1
    return {
        "ms_per_bit": ref["extrapolated_ms_per_bit"] * mult,
        "value": 1000.0 / (ref["extrapolated_ms_per_bit"] * mult), "unit": "gates/s",
        "cores": ref["cores_used"], "kind": "reference", "measured_in_this_run": False,
        "where": "build container (%s), 1 core, K<=5 blind-rotate iterations at B=32 x (500/K); %s" % (
            ref["cpu"], ref["label"]),
        "source": "profiles/reference_python_cpu_timing.json (tools/time_reference_cpu.py)"}


def shard_parity(gate, transform, sample_bits, cs_host, out_host, dist, dev, world, group=None):
    """N > 1: EVERY rank compares the first `sample_bits` ciphertexts of its own shard (its own inputs, its own
    output of the last timed step) with the CPU oracle and the counts are all-reduced, so that a multi-GPU line
    carries a parity verdict over all ranks (the N = 1 line compares a larger sample, cpu_baseline_and_parity)."""
    from oracle import oracle as orc
    # the ranks of this node share its host cores (torch.distributed.run exports OMP_NUM_THREADS=1 by default)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    orc.set_num_threads(max(1, (os.cpu_count() or 1) // max(1, local_world)))
    lwe_key, tlwe_key, ck = orc.make_key_pair(orc.DeterministicRNG(123))
    n = sample_bits
    cs = [tuple(x[:n] for x in c) for c in cs_host]
    dist.barrier(group=group)       # all ranks start their oracle run together: the host cores are shared evenly
    t_cpu = time.time()
    ref = orc.gate_mux(ck, cs[0], cs[1], cs[2]) if gate == 'mux' else orc.gate('gate_nand', ck, cs[0], cs[1])
    t_cpu = time.time() - t_cpu
    da = (out_host[0][:n].astype(numpy.int64) - ref[0].astype(numpy.int64) + 2**31) % 2**32 - 2**31
    db = (out_host[1][:n].astype(numpy.int64) - ref[1].astype(numpy.int64) + 2**31) % 2**32 - 2**31
    sums = torch.tensor([int((da != 0).sum() + (db != 0).sum()), int(da.size + db.size),
                         int((out_host[2][:n] != ref[2]).sum()), 1], dtype=torch.int64, device=dev)
    worst = torch.tensor([int(max(abs(da).max(), abs(db).max()))], dtype=torch.int64, device=dev)
    dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(worst, op=dist.ReduceOp.MAX, group=group)
    entry = {"bits_per_rank": n, "ranks_reporting": int(sums[3].item()), "words": int(sums[1].item()),
             "differing": int(sums[0].item()), "variances_differing": int(sums[2].item()),
             "note": "every rank: first %d ciphertexts of its shard (a[500] and b) vs the CPU oracle on the same inputs, "
                     "counts all-reduced" % n}
    if transform != "NTT":
        entry["max_abs_diff_lsb"] = int(worst.item())
        entry["tolerance_lsb"] = 16
    assert entry["ranks_reporting"] == world
    # the CPU baseline of an N > 1 line: THIS rank's oracle run (its share of the host cores, the other ranks of the
    # node running theirs at the same time), scaled to the node: `world` ranks finished `world * n` bits in the
    # slowest rank's time
    slowest = torch.tensor([t_cpu], dtype=torch.float64, device=dev)
    dist.all_reduce(slowest, op=dist.ReduceOp.MAX, group=group)
    dt = float(slowest.item())
    base = dict(value=world * n / dt, unit="gates/s", cores=orc.num_threads() * local_world, kind="port",
                sample="first %d bits of EVERY rank's %s shard (%d ranks x %d threads at the same time on this host), full "
                       "n=500 bootstrap + keyswitch, slowest rank %.1f s" % (n, gate.upper(), world, orc.num_threads(), dt),
                ms_per_bit=1000.0 * dt / (world * n), threads_per_rank=orc.num_threads(), ranks=world,
                where="this host, in this run (the C restatement of the reference's *_cpu.py composition, OpenMP over bits)")
    ref_py = reference_python_entry(gate)
    if ref_py is not None:
        base["reference_python"] = ref_py
    return entry, base


def peer_main(args):
    """`--gather-backend peer`: the RCCL-free route of the N > 1 line.  ONE process drives all N GPUs through one
    DeviceThread each -- the reference's own scheme (examples/multi_gpu.py:46-114: a thread and a Thread object per GPU,
    the main thread collects) -- and the slices reach GPU 0 by hipMemcpyPeerAsync on the source streams (nufhe_gather,
    multi_gpu.gather_threads).  Launches are asynchronous (a gate is tens of milliseconds, a launch microseconds), so one
    host thread keeps every GPU busy.  Same shard sizes, same timed region (K steps between device-wide synchronisations,
    the gather of every step inside it), same JSON shape, fewer diagnostics."""
    import nufhe_amd
    from nufhe_amd import multi_gpu
    from nufhe_amd.device import DeviceThread
    n = args.gpus
    if torch.cuda.device_count() < n:
        print(json.dumps({"error": "--gather-backend peer needs %d GPUs in one process, %d visible" % (n, torch.cuda.device_count())}))
        return
    B = args.bits
    thrs, vms, css, mss, outs, sks = [], [], [], [], [], []
    for d in range(n):
        thr = DeviceThread(d)
        ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(123), thread=thr)
        secret_key, cloud_key = ctx.make_key_pair(transform_type=args.transform)
        if args.engine != "native":
            cloud_key.set_engine(args.engine)
        vm = ctx.make_virtual_machine(cloud_key)
        rs = numpy.random.RandomState(456 + d)
        ms = [rs.randint(0, 2, size=(B,)).astype(bool) for _ in range(3)]
        ctx.rng = nufhe_amd.DeterministicRNG(1000 + d)
        with torch.cuda.device(thr.device):
            cs = [ctx.encrypt(secret_key, m) for m in ms]
            out = vm.empty_ciphertext((B,))
        thrs.append(thr); vms.append((ctx, vm)); css.append(cs); mss.append(ms); outs.append(out); sks.append(secret_key)

    def step():
        for d in range(n):
            with torch.cuda.device(thrs[d].device):
                if args.gate == "mux":
                    vms[d][1].gate_mux(*css[d], dest=outs[d])
                else:
                    vms[d][1].gate_nand(css[d][0], css[d][1], dest=outs[d])
        return multi_gpu.gather_threads(thrs[0], [(thrs[d], outs[d]) for d in range(n)])

    def sync_all():
        for thr in thrs:
            with torch.cuda.device(thr.device):
                thr.synchronize()
    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full = step()
    sync_all()
    elapsed = time.perf_counter() - t0
    ok = True
    for d in range(n):
        with torch.cuda.device(thrs[d].device):
            dec = vms[d][0].decrypt(sks[d], outs[d])
        exp = numpy.where(mss[d][0], mss[d][1], mss[d][2]) if args.gate == "mux" else ~(mss[d][0] & mss[d][1])
        ok = ok and bool((dec == exp).all())
        ok = ok and bool((full.a[d * B:(d + 1) * B].to(thrs[d].device) == outs[d].a).all())       # slice d sits at its place
    ms_per_step = 1e3 * elapsed / args.steps
    print(json.dumps({
        "metric": "bootstrapped gates/sec (%s), %d-bit batch per GPU, %s%s, n=500 N=1024 k=1 l=2" % (
            args.gate.upper(), B, args.transform, " (exact-fft engine)" if args.engine != "native" else ""),
        "value": n * B * args.steps / elapsed, "unit": "gates/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "ms_per_bit": ms_per_step / (n * B), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64 mod 2^64-2^32+1 (NTT) / int32 torus" if args.transform == "NTT" and args.engine == "native"
        else "f64 complex (folded FFT-512) / int32 torus", "engine": args.engine,
        "data": "synthetic (seeded keys and ciphertexts, resident in HBM)", "correct": ok,
        "config": {"workload": "gate_%s, %d-bit batch per GPU" % (args.gate, B), "bits_per_gpu": B, "transform": args.transform,
                   "parallelism": "bits sharded over %d GPU(s) driven by ONE process, keys generated per GPU from one seed" % n},
        "gather": {"backend": "peer (hipMemcpyPeerAsync on the source streams, nufhe_gather)", "dst": 0, "verified": ok,
                   "collectives_per_step": 0, "inside_timed_region": True},
        "roofline": None, "cpu_baseline": None,
        "note": "RCCL-free route of the N > 1 line (--gather-backend peer): no roofline / cpu_baseline objects -- run the default "
                "route, or N = 1, for those"}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--gate", choices=["nand", "mux"], default="nand")
    ap.add_argument("--bits", type=int, default=4096, help="bits per GPU")
    ap.add_argument("--transform", choices=["NTT", "FFT"], default="NTT",
                    help="NTT = BASELINE configs 2-4 (bit-exact path); FFT = config 5 (fp64, tolerance path)")
    ap.add_argument("--engine", choices=["native", "exact-fft"], default="native",
                    help="arithmetic of the NTT path: native = u64 prime-field NTT kernels (the BASELINE headline); "
                         "exact-fft = fp64 folded FFT on a 16-bit split key, bit-identical by construction "
                         "(nufhe_cloudkey_set_engine)")
    ap.add_argument("--gather-backend", choices=["rccl", "gloo", "peer"], default=None,
                    help="N > 1, how the result slices reach rank 0: rccl = one async RCCL gather per step (default); gloo = the "
                         "same collective staged through the host (also the automatic fallback when RCCL cannot start); peer = "
                         "ONE process drives all N GPUs and collects with hipMemcpyPeerAsync (nufhe_gather), no RCCL at all")
    ap.add_argument("--cpu-sample-bits", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the secondary measurements (other gate / FFT transform) reported under 'other_configs'")
    args = ap.parse_args()

    if args.gather_backend is None:
        args.gather_backend = "gloo" if os.environ.get("NUFHE_BENCH_BACKEND") == "gloo" else "rccl"
    if args.gather_backend == "peer":
        # one process, one DeviceThread per GPU (the reference's own model, examples/multi_gpu.py:46-114).  Under a
        # torch.distributed.run launch only rank 0 works; the other ranks leave at once.
        if int(os.environ.get("RANK", "0")) == 0:
            peer_main(args)
        return
    if "RANK" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` is a complete command: it starts its own N ranks on this node (one per GPU,
        # torch.distributed.run, free port on 127.0.0.1) the way the reference's example starts its own per-GPU
        # workers (examples/multi_gpu.py:86-114); rank 0 of the children prints the JSON line.  N = 1 never gets here.
        from nufhe_amd import multi_gpu as _mg
        try:
            sys.exit(_mg.launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus,
                                      backend="gloo" if args.gather_backend == "gloo" else "nccl"))
        except RuntimeError as e:
            sys.exit("bench.py: " + str(e))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "RANK" in os.environ and world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # launched by torch.distributed.run (RANK set): the process group is created for ANY world size, so
    # a 1-GPU box still drives device-side RCCL gathers; a plain `python bench.py` has no group
    use_dist = world > 1 or ("RANK" in os.environ and os.environ.get("NUFHE_BENCH_NO_DIST") != "1")
    errors = []          # what went wrong on the way to a working gather (reported in the line, never a traceback)
    ctl = None           # control-plane group: barriers and scalar reductions always travel over gloo on the host
    data_backend = None  # what carries the result slices
    if use_dist:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # RCCL has never run here with more than one rank before the driver's own 8-GPU run: make its first contact loud
        # (NCCL_DEBUG=WARN prints the failing call) and keep the dmabuf IPC mode the host driver needs
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = "gloo" if args.gather_backend == "gloo" else "nccl"
        if backend == "nccl" and torch.cuda.device_count() <= local_rank:
            # (not a software failure to fall back from: a line that says n_gpus = N must come from N GPUs)
            sys.exit("bench.py: rank %d (local %d) has no GPU of its own: %d visible; RCCL needs one per rank" % (
                rank, local_rank, torch.cuda.device_count()))
        local_rank = local_rank % max(1, torch.cuda.device_count())      # (gloo test route: ranks may share a GPU)
        torch.cuda.set_device(local_rank)
        fail_hook = os.environ.get("NUFHE_BENCH_FAIL_RCCL", "")          # tests only: "init" / "gather" raise at that point
        if backend == "nccl":
            try:
                if fail_hook == "init":
                    raise RuntimeError("NUFHE_BENCH_FAIL_RCCL=init (test hook)")
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank),
                                        timeout=datetime.timedelta(seconds=300))
                ctl = dist.new_group(backend="gloo")
                data_backend = "nccl"
            except Exception as e:                  # RCCL could not start: the run still yields N kernel times
                errors.append("RCCL process group: %s: %s" % (type(e).__name__, str(e)[:500]))
                if dist.is_initialized():
                    dist.destroy_process_group()
                backend = "gloo"
        if backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
            data_backend = "gloo"

    import nufhe_amd
    from nufhe_amd import _lib
    from nufhe_amd.device import DeviceThread

    thr = DeviceThread(local_rank)
    ctx = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(123), thread=thr)
    # cloud-key generation on the GPU, replicated on every rank (98.6 MB): host RNG draws in
    # the reference's order + upload + TLWE encryptions / transforms / keyswitch-key assembly in HIP kernels
    keygen_ms = {}

    def timed_key_pair(c, label, **kw):
        thr.synchronize()
        t_k = time.perf_counter()
        pair = c.make_key_pair(**kw)
        thr.synchronize()
        keygen_ms[label] = 1e3 * (time.perf_counter() - t_k)
        return pair
    secret_key, cloud_key = timed_key_pair(ctx, args.transform, transform_type=args.transform)
    if args.engine != "native":
        if args.transform != "NTT":
            sys.exit("bench.py: --engine %s applies to --transform NTT" % args.engine)
        cloud_key.set_engine(args.engine)
    xfft = args.engine == "exact-fft"
    vm = ctx.make_virtual_machine(cloud_key)

    B = args.bits
    data_rng = numpy.random.RandomState(456 + rank)
    ms = [data_rng.randint(0, 2, size=(B,)).astype(bool) for _ in range(3)]
    ctx.rng = nufhe_amd.DeterministicRNG(1000 + rank)
    cs = [ctx.encrypt(secret_key, m) for m in ms]
    from nufhe_amd import multi_gpu
    # Result buffers.  With a process group: TWO packed buffers (a | b | variances of the slice in one int32
    # allocation) that alternate, so the gather of step i (ONE RCCL collective, async) overlaps the gate of step i + 1;
    # a buffer is reused only after the collective that read it has been waited for (examples/multi_gpu.py:104-107 is
    # the reference's blocking collection of pickled slices).
    if use_dist:
        packs = [multi_gpu.PackedCiphertext(cs[0].params, B, thr.device) for _ in range(2)]
        outs = [p.ciphertext for p in packs]
        recv = [torch.empty(world * B * (cs[0].params.size + 2), dtype=torch.int32, device=thr.device)
                if rank == 0 and data_backend == "nccl" else None for _ in range(2)]
    else:
        packs, recv = None, None
        outs = [vm.empty_ciphertext((B,))]
    pending = [None, None]
    step_no = [0]

    def step():
        k = step_no[0] % len(outs)
        step_no[0] += 1
        if pending[k] is not None:
            pending[k].wait(unpack=False)        # orders this stream behind the collective that read buffer k
            pending[k] = None
        if args.gate == "mux":
            vm.gate_mux(cs[0], cs[1], cs[2], dest=outs[k])
        else:
            vm.gate_nand(cs[0], cs[1], dest=outs[k])
        if use_dist:
            pending[k] = multi_gpu.gather_packed_async(packs[k], world * B, dst=0, recv=recv[k], group=data_group)
        return k

    def drain():
        for k in range(2):
            if pending[k] is not None:
                pending[k].wait(unpack=False)
                pending[k] = None

    import ctypes
    lib = _lib.lib()
    data_group = None          # default group (RCCL, or gloo when that is the default); the gloo side group after a failure
    if use_dist and data_backend == "nccl":
        # first contact with RCCL on more than one rank: one gather, checked on every rank.  If it raises anywhere, ALL
        # ranks switch to the host-staged gather over the gloo group (agreed over gloo) and the line says why.
        failed = 0
        try:
            if os.environ.get("NUFHE_BENCH_FAIL_RCCL") == "gather":
                raise RuntimeError("NUFHE_BENCH_FAIL_RCCL=gather (test hook)")
            step(); drain(); torch.cuda.synchronize()
        except Exception as e:
            failed = 1
            errors.append("first RCCL gather on rank %d: %s: %s" % (rank, type(e).__name__, str(e)[:500]))
            pending[0] = pending[1] = None
        flag = torch.tensor([failed], dtype=torch.int64)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=ctl)
        if int(flag.item()):
            if not failed:
                errors.append("the first RCCL gather failed on another rank")
            data_backend, data_group = "gloo", ctl
            recv = [None, None]
    for _ in range(args.warmup):
        step()
    drain()
    lib.nufhe_profile_enable(thr.handle, 1)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier(group=ctl)
    br_ms, ks_ms, clock_ghz, wave_ms = [], [], [], []
    t0 = time.perf_counter()
    last = 0
    for _ in range(args.steps):
        last = step()               # nothing in the loop waits for the device: the library records HIP events around the
                                    # kernels of every gate (on the launch stream) and keeps them (nufhe_profile_history)
    drain()                                         # every gather of the timed steps completes inside the timed region
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0          # this rank alone (a straggler shows up here)
    if use_dist:
        dist.barrier(group=ctl)
    elapsed = time.perf_counter() - t0
    # HIP-event durations of the bootstrap kernel / the keyswitch stage of EVERY timed step, read after the timed region
    hist_n = min(args.steps, 256)
    hist_br = (ctypes.c_float * hist_n)(); hist_ks = (ctypes.c_float * hist_n)(); got = ctypes.c_int(0)
    _lib.check(lib.nufhe_profile_history(thr.handle, hist_br, hist_ks, hist_n, ctypes.byref(got)))
    br_ms = [hist_br[i] for i in range(got.value)]; ks_ms = [hist_ks[i] for i in range(got.value)]
    g = ctypes.c_double(); w = ctypes.c_double()
    if lib.nufhe_profile_clock(thr.handle, ctypes.byref(g), ctypes.byref(w)) == 0:      # in-kernel clock of the last step
        clock_ghz.append(g.value); wave_ms.append(w.value)
    per_rank_ms = [1e3 * own_elapsed / args.steps]
    per_rank_kernel = None
    if use_dist:
        dev = "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=ctl)
        elapsed = float(t.item())
        own = torch.tensor([1e3 * own_elapsed / args.steps], dtype=torch.float64, device=dev)
        every = torch.zeros(world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(every, own, group=ctl)
        per_rank_ms = [float(x) for x in every.cpu()]
        # the dominant kernel on every rank: mean HIP-event duration of its timed launches and its in-kernel clock, so
        # that a slow GPU shows in the roofline figure (rank 0's kernel is the one `roofline` is computed from)
        mine = torch.tensor([float(numpy.mean(br_ms)) if br_ms else 0.0, float(numpy.mean(ks_ms)) if ks_ms else 0.0,
                             clock_ghz[0] if clock_ghz else 0.0], dtype=torch.float64, device=dev)
        allk = torch.zeros(3 * world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allk, mine, group=ctl)
        allk = allk.cpu().view(world, 3)
        per_rank_kernel = {"kernel_ms": [float(x) for x in allk[:, 0]], "keyswitch_ms": [float(x) for x in allk[:, 1]],
                           "clock_ghz_in_kernel": [float(x) for x in allk[:, 2]]}

    # the result gather alone (reported separately; it is also part of every timed step)
    gather_ms = None
    gathered_ok = None
    out = outs[last]                                # the result of the last timed step
    if use_dist:
        full = multi_gpu.gather_packed_async(packs[last], world * B, dst=0, group=data_group).wait()
        torch.cuda.synchronize(); dist.barrier(group=ctl)
        t1 = time.perf_counter()
        for _ in range(5):
            multi_gpu.gather_packed_async(packs[last], world * B, dst=0, recv=recv[last], group=data_group).wait(unpack=False)
        torch.cuda.synchronize(); dist.barrier(group=ctl)
        gather_ms = 1e3 * (time.perf_counter() - t1) / 5
        if rank == 0:
            # rank 0's own slice must sit at the head of the gathered arrays
            gathered_ok = bool((full[0][:B] == out.a).all() and (full[1][:B] == out.b).all()
                               and (full[2][:B] == out.current_variances).all() and full[0].shape[0] == world * B)

    # secondary measurements, OUTSIDE the timed region: the other BASELINE configurations on the same
    # ciphertexts (3 steps each after 1 warm-up); reported under "other_configs", never in "value"
    other = {}
    if not args.no_extra:
        def measure(fn, nsteps=3):
            fn()
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier(group=ctl)
            t1 = time.perf_counter()
            for _ in range(nsteps):
                fn()
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier(group=ctl)
            return (time.perf_counter() - t1) / nsteps
        out2 = vm.empty_ciphertext((B,))
        out3 = vm.empty_ciphertext((B,))
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
        sk_o, ck_o = timed_key_pair(ctx_o, other_tr, transform_type=other_tr)     # same secret key bits (same seed)
        vm_o = ctx_o.make_virtual_machine(ck_o)
        dt = measure(lambda: vm_o.gate_nand(cs[0], cs[1], dest=out3))
        ok = bool((ctx_o.decrypt(sk_o, out3) == ~(ms[0] & ms[1])).all())
        other["gate_nand_%s" % other_tr] = {
            "ms_per_step_per_gpu": 1e3 * dt, "ms_per_bit": 1e3 * dt / B, "gates_per_s_per_gpu": B / dt, "correct": ok}
        out3 = out3.copy()                               # (the buffer is reused below; this copy goes to the parity leg)
        del vm_o, ck_o
        out4 = None
        if args.transform == "NTT":
            # the other engine of the NTT path on the SAME key and ciphertexts (its output words go through the same
            # parity leg below: differing must be 0)
            other_engine = "native" if xfft else "exact-fft"
            out4 = vm.empty_ciphertext((B,))
            cloud_key.set_engine(other_engine)
            try:
                t_img = time.perf_counter()
                vm.gate_nand(cs[0][:1], cs[1][:1])                     # builds the split key image on first use
                torch.cuda.synchronize()
                keygen_ms["%s_engine_first_gate" % other_engine] = 1e3 * (time.perf_counter() - t_img)
                dt = measure(lambda: vm.gate_nand(cs[0], cs[1], dest=out4))
                out5 = vm.empty_ciphertext((B,))
                dtm = measure(lambda: vm.gate_mux(cs[0], cs[1], cs[2], dest=out5))
                del out5
            finally:
                cloud_key.set_engine(args.engine)
            ok = bool((ctx.decrypt(secret_key, out4) == ~(ms[0] & ms[1])).all())
            other["gate_nand_%s_engine_NTT" % other_engine.replace("-", "_")] = {
                "engine": other_engine, "ms_per_step_per_gpu": 1e3 * dt, "ms_per_bit": 1e3 * dt / B,
                "gates_per_s_per_gpu": B / dt, "correct": ok, "mux_ms_per_step_per_gpu": 1e3 * dtm,
                "note": "same NTT key, same ciphertexts, the other engine of nufhe_cloudkey_set_engine; bit-identical "
                        "outputs by construction (include/nufhe_hip.h), checked under `parity`"}
        if world == 1 and args.transform == "NTT" and B >= 256:
            # small batches (one dependent gate of a circuit at a time: latency, not throughput): the first 1 / 256 bits of
            # the same ciphertexts on both engines of the NTT key, 10 calls each
            small = {}
            for engine in ("native", "exact-fft"):
                cloud_key.set_engine(engine)
                try:
                    for nb in (1, 256):
                        a_s, b_s = cs[0][:nb], cs[1][:nb]
                        dest_s = vm.empty_ciphertext((nb,))
                        dts = measure(lambda: vm.gate_nand(a_s, b_s, dest=dest_s), nsteps=10)
                        small["%s_%d_bits" % (engine.replace("-", "_"), nb)] = {
                            "ms_per_gate_call": 1e3 * dts,
                            "correct": bool((ctx.decrypt(secret_key, dest_s) == ~(ms[0][:nb] & ms[1][:nb])).all())}
                finally:
                    cloud_key.set_engine(args.engine)
            small["note"] = ("NAND on the first 1 / 256 bits: native = 8 waves per bit on the u64 kernels (k_bootstrap_team8), "
                             "exact-fft = 4 waves per bit (k_bootstrap_xfft_quad); identical output words")
            other["small_batch_latency"] = small
        if world == 1:
            ctx_k = nufhe_amd.Context(rng=nufhe_amd.DeterministicRNG(123), thread=thr)
            k2 = {}
            for tr in ("NTT", "FFT"):
                sk_k, ck_k = timed_key_pair(ctx_k, tr + "_k2", transform_type=tr, tlwe_mask_size=2)
                # the same NAND with tlwe_mask_size = 2 (test/test_gates.py:96-100), 3 timed calls per engine
                vm_k = ctx_k.make_virtual_machine(ck_k)
                ck2 = [ctx_k.encrypt(sk_k, m) for m in ms[:2]]
                outk = vm_k.empty_ciphertext((B,))
                for engine in (("native", "exact-fft") if tr == "NTT" else ("native",)):
                    ck_k.set_engine(engine) if tr == "NTT" else None
                    vm_k.gate_nand(ck2[0], ck2[1], dest=outk)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(3):
                        vm_k.gate_nand(ck2[0], ck2[1], dest=outk)
                    torch.cuda.synchronize()
                    dtk = (time.perf_counter() - t1) / 3
                    k2["gate_nand_%s%s" % (tr, "" if engine == "native" else "_exact_fft_engine")] = {
                        "ms_per_step_per_gpu": 1e3 * dtk, "ms_per_bit": 1e3 * dtk / B,
                        "correct": bool((ctx_k.decrypt(sk_k, outk) == ~(ms[0] & ms[1])).all())}
                del vm_k, ck_k, sk_k, ck2, outk
            other["tlwe_mask_size_2"] = k2
        other["keygen_ms"] = dict(keygen_ms, note="Context.make_key_pair wall time incl. host random numbers "
                                  "(numpy RandomState, reference draw order) and their upload; the first entry "
                                  "also pays the one-time kernel loading")
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

    # correctness of what was timed: every decrypted bit equals the truth table -- on EVERY rank (each rank holds
    # the secret key of the shared seed and its own plaintexts); the verdicts are combined below
    dec = ctx.decrypt(secret_key, out)
    expect = numpy.where(ms[0], ms[1], ms[2]) if args.gate == "mux" else ~(ms[0] & ms[1])
    correct = bool((dec == expect).all())
    multi_parity = multi_base = None
    if use_dist:
        dev = "cpu"
        okt = torch.tensor([1 if correct else 0], dtype=torch.int64, device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN, group=ctl)
        correct = bool(okt.item())
        if world > 1 and not args.no_cpu_baseline:
            def host_(ct):
                return tuple(x.detach().cpu().numpy() for x in (ct.a, ct.b, ct.current_variances))
            multi_parity, multi_base = shard_parity(args.gate, args.transform, min(B, args.cpu_sample_bits or 256), [host_(c) for c in cs], host_(out),
                                        dist, dev, world, group=ctl)

    if use_dist:
        every_err = [None] * world
        dist.all_gather_object(every_err, errors, group=ctl)
        errors = sorted({e for lst in every_err for e in (lst or [])})
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
            "metric": "bootstrapped gates/sec (%s), %d-bit batch per GPU, %s%s, n=500 N=1024 k=1 l=2" % (
                args.gate.upper(), B, args.transform, " (exact-fft engine)" if xfft else ""),
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
            "dtype": ("f64 complex (folded FFT-512 on a 16-bit split key: exact integer results) / int32 torus" if xfft
                      else "u64 mod 2^64-2^32+1 (NTT) / int32 torus" if args.transform == "NTT"
                      else "f64 complex (folded FFT-512) / int32 torus"),
            "engine": args.engine,
            "data": "synthetic (seeded keys and ciphertexts, resident in HBM)",
            "correct": correct,
            "config": {"workload": "gate_%s, %d-bit batch per GPU (BASELINE config %s)" % (
                args.gate, B, "5" if args.transform == "FFT" else (
                    "3" if args.gate == "mux" else ("2" if world == 1 else "4"))),
                "bits_per_gpu": B, "transform": args.transform, "parallelism": "bits sharded over %d GPU(s), keys replicated" % world},
            "roofline": {},
        }
        # `roofline` IS BASELINE.json's axis (the contract of the bench line): the ALGORITHMIC bytes of one launch of the
        # dominant kernel -- SURVEY 8d: every bit streams its 32.8 MB key once, no reuse -- over the HIP-event duration of
        # the launch, against the 8 TB/s HBM peak; `traffic` = the HBM bytes the counters saw.  What actually bounds
        # the kernel (VALU issue) is priced under `roofline.valu_issue`, with every fraction naming
        # its denominator.  (Until round 4 the issue-model fraction was the headline `frac` and this object sat under
        # `streaming_model`; the numbers are the same, the one BASELINE names now comes first.)
        live_clock = sum(clock_ghz) / len(clock_ghz) if clock_ghz else None
        kernel_name = "k_bootstrap_xfft" if xfft else "k_bootstrap_fft" if args.transform == "FFT" else "k_bootstrap<1>"
        issue = issue_roofline("XFFT" if xfft else args.transform, B, n_rot, 500, br_avg, live_clock) or {}
        traffic = pmc_traffic("XFFT" if xfft else args.transform, args.gate, B)
        if xfft:
            a_kernel = 2 * a_kernel                 # the split key image is two fp64 planes: 65.5 MB per bit and rotation
            achieved = a_kernel / (br_avg * 1e-3) / 1e9
        roof = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": None if traffic is None else
                              "tracked rocprofv3 --pmc pass of this same command (profiles/pmc_traffic.json), not this run",
            "kernel": "%s (fused mod-switch + blind rotate + extract)" % kernel_name,
            "kernel_ms": br_avg, "keyswitch_ms": ks_avg,
            "algorithmic_bytes_per_launch": a_kernel,
            "gate_streaming_model_GBs": a_gate / (ms_per_step * 1e-3) / 1e9,
            "frac_is": "algorithmic bytes per launch (streaming model, SURVEY 8d) / HIP-event duration / 8 TB/s",
            "note": "BASELINE.json's HBM axis.  The kernel shares each key row between all resident waves through L2, so "
                    "`achieved` is a model figure, not traffic (`traffic` = measured HBM bytes per launch, ~0.4 % of the "
                    "algorithmic bytes: no wasted re-reads); the kernel is bound by VALU issue -- see `valu_issue`",
            "valu_issue": issue,
            "keyswitch": keyswitch_roofline(B, ks_avg, B > 2 * 256)}
        if args.transform == "FFT" or xfft:
            # the streaming-model axis carries no information for the fp64 kernels: key rows are shared by the resident
            # waves through L2, so the model figure exceeds the HBM peak; the second axis is the vector fp64 rate
            fp64 = (issue.get("isa_mix_per_iteration") or {}).get("valu_f64")
            if fp64:
                tflops = 2.0 * fp64 * 64 * B * n_rot * 500 / (br_avg * 1e-3) / 1e12      # an fp64 VALU instruction = 64 lanes x (fma = 2 flop; adds counted alike)
                roof["fp64_vector"] = {"achieved_tflops_upper": tflops, "peak_tflops": FP64_VECTOR_PEAK_TFLOPS,
                                       "frac_upper": tflops / FP64_VECTOR_PEAK_TFLOPS,
                                       "note": "fp64 VALU instructions per iteration (profiles/isa_mix.json) x 64 lanes x 2 flop over the "
                                               "kernel time: an upper figure (adds and multiplies count as 2 like an fma)"}
            roof["frac_note"] = ("streaming-model frac > 1 = L2 reuse of key rows between resident waves, not skipped work (the "
                                 "parity leg checks the timed output); read `valu_issue` and `fp64_vector` for what bounds this kernel")
        if "clock_ghz_in_kernel" in issue:
            roof["clock_ghz_in_kernel"] = issue["clock_ghz_in_kernel"]
        if wave_ms:
            # one wave's blind rotation (start to end) against the kernel: rounds x wave life time ~ kernel time when
            # the waves of every SIMD finish together (pacing)
            roof["wave_ms_in_kernel"] = sum(wave_ms) / len(wave_ms)
        # self-check of the SIMD-partner pacing: the two waves of a SIMD must finish together (rounds x one
        # wave's life time ~ kernel time) and the clock must not have sagged; a line that fails this was measured on a
        # box / in a state where the kernel does not run as designed
        rounds = -(-(B * n_rot) // (256 * 8))
        checks = {}
        if wave_ms:
            checks["rounds"] = rounds
            checks["wave_ms_x_rounds_over_kernel_ms"] = roof["wave_ms_in_kernel"] * rounds / br_avg
            checks["pacing_ok"] = bool(checks["wave_ms_x_rounds_over_kernel_ms"] >= 0.9)
        if live_clock:
            checks["clock_ok"] = bool(live_clock >= 2.2 or args.transform == "FFT" and live_clock >= 2.0)
        roof["self_check"] = checks
        if per_rank_kernel is not None:
            km = per_rank_kernel["kernel_ms"]
            # (the in-kernel clock exists for the wave-per-bit kernels only: batches of more than 4 x CUs bits per GPU)
            cg = per_rank_kernel["clock_ghz_in_kernel"] = [c if c > 0 else None for c in per_rank_kernel["clock_ghz_in_kernel"]]
            have = [c for c in cg if c is not None]
            per_rank_kernel.update({
                "kernel_ms_min": min(km), "kernel_ms_max": max(km),
                "clock_ghz_min": min(have) if have else None, "clock_ghz_max": max(have) if have else None,
                "streaming_frac_min": a_kernel / (max(km) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "streaming_frac_max": a_kernel / (min(km) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "note": "one entry per rank, rank order; `roofline` above is rank 0's kernel; the slowest rank sets `value`"})
            roof["per_rank"] = per_rank_kernel
        result["roofline"] = roof
        result["per_rank_ms_per_step"] = per_rank_ms
        if gather_ms is not None:
            result["gather_ms"] = gather_ms
            result["gather_bytes_per_rank"] = B * 2008
            result["gather"] = {"backend": data_backend, "dst": 0, "verified": gathered_ok,
                                "requested": args.gather_backend, "control_plane": "gloo (barriers, timings, verdicts)",
                                "env": {k: os.environ.get(k) for k in ("NCCL_DEBUG", "HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_SOCKET_IFNAME",
                                                                        "MASTER_ADDR", "OMP_NUM_THREADS")},
                                "env_note": "HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver only supports dmabuf IPC, without it RCCL "
                                            "fails with hipIpcGetMemHandle: invalid argument; NCCL_DEBUG=WARN names a failing call",
                                "collectives_per_step": 1, "overlapped": True,
                                "note": "a | b | variances of a slice share one buffer: ONE gather per step, started "
                                        "async after the gate and overlapped with the next gate (two result buffers "
                                        "alternate); every gather of the timed steps completes inside the timed region; "
                                        "gather_ms = the same collective alone, blocking"}
        if multi_parity is not None:
            result["parity"] = multi_parity
            result["cpu_baseline"] = multi_base
            bad_multi = (multi_parity["differing"] != 0 if args.transform == "NTT"
                         else multi_parity["max_abs_diff_lsb"] > multi_parity["tolerance_lsb"])
            result["correct"] = bool(result["correct"] and not bad_multi)
        if world > 1:
            result["config"]["note"] = ("weak scaling: %d bits per GPU; config 4 (32768 bits over 8 GPUs) is this line at "
                                        "N = 8 -- the same run read as a strong-scaling split of 32768 bits" % B)
        if errors:
            result["error"] = "; ".join(errors)
            result["error_note"] = ("the requested gather route did not work; the line was produced over the fallback named in "
                                    "gather.backend -- kernel times, parity and value are real, gather_ms is the fallback's")
        if other:
            result["other_configs"] = other
            oe = next((v for k, v in other.items() if k.startswith("gate_nand_") and k.endswith("_engine_NTT")), None)
            if oe and args.gate == "nand":
                # both engines of the NTT key side by side (same key, same ciphertexts, identical output words: `parity`);
                # `value` above is the engine named in `engine` -- by default the u64 prime-field kernels BASELINE names
                result["engines_side_by_side"] = {
                    args.engine: {"ms_per_step": ms_per_step, "gates_per_s": gates_per_s},
                    oe["engine"]: {"ms_per_step": oe["ms_per_step_per_gpu"], "gates_per_s": world * oe["gates_per_s_per_gpu"]},
                    "small_batch_ms_per_gate": {k: v["ms_per_gate_call"] for k, v in other.get("small_batch_latency", {}).items()
                                                if isinstance(v, dict)}}
        if world == 1 and not args.no_cpu_baseline:
            nthreads = os.cpu_count() or 1
            # ~10-20 s of oracle time for the headline gate on the box's 128 host threads (about 17 ms per bit)
            sample = min(B, args.cpu_sample_bits or max(16, min(768, 6 * nthreads)))

            def host(ct):
                return tuple(x.detach().cpu().numpy() for x in (ct.a, ct.b, ct.current_variances))
            main_label = "gate_%s_%s" % (args.gate, args.transform)
            gpu_out = {main_label: (args.gate, args.transform == "NTT", host(out))}
            if not args.no_extra:
                gpu_out["gate_%s_%s" % (other_gate, args.transform)] = (other_gate, args.transform == "NTT", host(out2))
                gpu_out["gate_nand_%s" % other_tr] = ("nand", other_tr == "NTT", host(out3))
                if out4 is not None:
                    gpu_out["gate_nand_%s_engine_NTT" % other_engine.replace("-", "_")] = ("nand", True, host(out4))
            result["cpu_baseline"], result["parity"] = cpu_baseline_and_parity(
                args.gate, sample, [host(c) for c in cs], gpu_out)
            result["parity"]["note"] = ("GPU output words (a[500] and b per bit) vs the CPU oracle on the same input "
                                        "ciphertexts; NTT legs must show differing = 0, the FFT leg is held to "
                                        "max_abs_diff_lsb <= tolerance_lsb (tests/test_gpu_fft.py)")
            bad = [k for k, v in result["parity"].items() if isinstance(v, dict) and (
                v["differing"] != 0 if k.endswith("NTT") else v.get("max_abs_diff_lsb", 0) > v.get("tolerance_lsb", 0))]
            result["correct"] = bool(result["correct"] and not bad)
            try:
                # the every-word runs of tools/extended_parity.py are too slow for this command (minutes of oracle time):
                # the builder's last ones travel with the line, labelled as what they are
                ext = _profile_json("extended_parity_latest.json")
                if ext:
                    result["parity"]["extended"] = dict(ext, measured_in_this_run=False)
            except (TypeError, ValueError):
                pass
        print(json.dumps(result))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
