#!/bin/bash
# run_reference_tests.sh -- the reference's OWN pytest files, byte for byte, against the drop-in (`import nufhe` = the alias
# package of this repo).  Step 1 (here, where /root/reference exists): copies the files into tools/scratch/reftests/ (git-ignored,
# shipped to the GPU box by gpurun) beside tests/reference_suite/conftest.py and records their hashes.  Step 2 (on the GPU
# box): `tools/run_reference_tests.sh run` executes them and writes gpurun_out/r06_reference_tests_unmodified.log.
#   tools/run_reference_tests.sh stage && gpurun --timeout 1800 -- 'tools/run_reference_tests.sh run'
# `NUFHE_NTT_ENGINE=exact-fft tools/run_reference_tests.sh run exact_fft`: the same files with every NTT key on the exact-FFT
# engine (log ..._unmodified_exact_fft.log).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
DIR="$ROOT/tools/scratch/reftests"
FILES="test_api_high_level.py test_api_low_level.py test_gates.py utils.py"
if [ "$1" = "stage" ]; then
    rm -rf "$DIR"; mkdir -p "$DIR"
    for f in $FILES; do cp /root/reference/test/$f "$DIR/$f"; done
    cp -r "$ROOT/tests/reference_suite/conftest.py" "$ROOT/tests/reference_suite/reikna" "$DIR/"
    (cd "$DIR" && sha256sum $FILES > SHA256SUMS && cat SHA256SUMS)
    exit 0
fi
mkdir -p "$ROOT/gpurun_out"
LOG="$ROOT/gpurun_out/r06_reference_tests_unmodified${2:+_$2}.log"
cd "$DIR" || exit 1
{
    echo "== the reference's test files, unmodified (sha256 below; staged from /root/reference/test by tools/run_reference_tests.sh),"
    echo "== run with PYTHONPATH=<repo root>: 'import nufhe' is the alias package nufhe -> nufhe_amd; conftest = tests/reference_suite/conftest.py"
    echo "== NUFHE_NTT_ENGINE=${NUFHE_NTT_ENGINE:-native} (engine of every NTT key: native u64 kernels | exact-fft)"
    sha256sum -c SHA256SUMS
    cat SHA256SUMS
    export PYTHONPATH="$ROOT:$DIR"
    for run in "test_api_high_level.py" "test_api_low_level.py" "test_gates.py -m 'not perf'"; do
        echo; echo "== python -m pytest -p no:cacheprovider -q -rs $run"
        eval python -m pytest -p no:cacheprovider -q -rs $run 2>&1 | tail -25
    done
    # the reference's own performance tests (the source of its README table, README.md:48-66): 4096-bit NAND / MUX and
    # a (128, 32) uint_min, both transforms, fused and multi-kernel bootstrap, in the reference's own report format
    echo; echo "== python -m pytest -p no:cacheprovider -q -rs -s test_gates.py -m perf --heavy-performance-load"
    python -m pytest -p no:cacheprovider -v -rs -s test_gates.py -m perf --heavy-performance-load 2>&1 | grep -v "^$" | tail -150
} > "$LOG" 2>&1
tail -60 "$LOG"
