#!/bin/bash
# sample_clocks.sh -- on the GPU box: rocm-smi sclk / power / temperature samples while bench.py runs (150 steps)
cd "$(dirname "$0")/.."
python bench.py --steps ${STEPS:-150} --warmup 2 --no-extra --no-cpu-baseline "$@" > gpurun_out/clk_bench.json 2>/dev/null &
BP=$!
sleep 6
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power|Temperature \(Sensor (junction|edge)|mclk|fclk|socclk" | tr '\n' ';'
  echo
  sleep 1
done
wait $BP
echo idle:
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ';'
echo
rocm-smi --showmaxpower --showperflevel 2>/dev/null | grep -v "^$" | head
tail -c 600 gpurun_out/clk_bench.json
