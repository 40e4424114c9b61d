#!/bin/bash
# closing_bench.sh TAG -- on the MI355X box: the bench lines a round closes with (gpurun_out/closing_TAG/): the driver's
# default command, MUX / NTT, NAND / FFT, MUX / FFT, the RCCL route at world size 1, `--gpus 2` WITHOUT a launcher (two ranks
# sharing the GPU over gloo), the 8-rank rehearsal of the driver's multi-GPU command (8 ranks sharing the GPU over gloo, 256
# bits each), independent small gates one by one against one gate_batch launch, the small-batch latencies and the circuit
# timings.
TAG=${1:-run}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/closing_$TAG
mkdir -p "$OUT"
cd "$ROOT"
python bench.py --steps 10 --warmup 2 2>/dev/null | tail -1 > "$OUT/bench_default_full.json"
python bench.py --steps 10 --warmup 2 --gate mux --no-extra --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/lines_other_configs.jsonl"
python bench.py --steps 10 --warmup 2 --transform FFT --no-extra 2>/dev/null | tail -1 >> "$OUT/lines_other_configs.jsonl"
python bench.py --steps 10 --warmup 2 --transform FFT --gate mux --no-extra --no-cpu-baseline 2>/dev/null | tail -1 >> "$OUT/lines_other_configs.jsonl"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 1 \
    --steps 10 --warmup 2 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 >> "$OUT/lines_other_configs.jsonl"
NUFHE_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 6 --warmup 2 --no-extra 2>/dev/null | tail -1 > "$OUT/bench_gpus2_self_launched_gloo_one_gpu.json"
NUFHE_BENCH_BACKEND=gloo python bench.py --gpus 8 --bits 256 --steps 3 --warmup 1 --no-extra 2>/dev/null | grep '^{' | tail -1 > "$OUT/bench_gpus8_rehearsal_gloo_one_gpu.json"
python tools/circuit_batch.py 2>/dev/null | tail -1 > "$OUT/circuit_batch.json"
python tools/latency_small.py 2>/dev/null | tail -1 > "$OUT/latency_small.json"
python tools/latency_sweep.py NTT 2>/dev/null | tail -1 > "$OUT/latency_sweep_ntt.json"
python tools/latency_sweep.py FFT 2>/dev/null | tail -1 > "$OUT/latency_sweep_fft.json"
python - "$OUT" <<'PY'
import json, sys, os
out = sys.argv[1]
d = json.load(open(os.path.join(out, "bench_default_full.json")))
print("default: %.3f ms/step, K1 %.3f ms, hbm-model frac %.3f, valu-issue frac %.3f, correct %s, parity %s" % (
    d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"]["valu_issue"].get("frac", 0), d["correct"],
    {k: v.get("differing") for k, v in d["parity"].items() if isinstance(v, dict)}))
for k, v in d.get("other_configs", {}).items():
    if isinstance(v, dict) and "ms_per_step_per_gpu" in v:
        print("  %s: %.3f ms" % (k, v["ms_per_step_per_gpu"]))
for line in open(os.path.join(out, "lines_other_configs.jsonl")):
    e = json.loads(line)
    print("%s | n_gpus %d: %.3f ms/step, K1 %.3f" % (e["config"]["workload"], e["n_gpus"], e["ms_per_step"], e["roofline"]["kernel_ms"]))
e = json.load(open(os.path.join(out, "bench_gpus2_self_launched_gloo_one_gpu.json")))
print("--gpus 2 self-launched: n_gpus %d, %.3f ms/step, parity %s" % (e["n_gpus"], e["ms_per_step"], e["parity"]))
e = json.load(open(os.path.join(out, "bench_gpus8_rehearsal_gloo_one_gpu.json")))
print("--gpus 8 rehearsal: n_gpus %d, parity %s, cpu_baseline %.1f gates/s on %d cores" % (
    e["n_gpus"], {k: e["parity"][k] for k in ("ranks_reporting", "differing")}, e["cpu_baseline"]["value"], e["cpu_baseline"]["cores"]))
print(open(os.path.join(out, "circuit_batch.json")).read().strip())
print(open(os.path.join(out, "latency_small.json")).read().strip())
PY
