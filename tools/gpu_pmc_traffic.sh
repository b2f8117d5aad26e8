#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes only (separate rocprofv3 runs, --kernel-trace, per the guide) for a bench mode:
#   BENCH_ARGS="--mode train" bash tools/gpu_pmc_traffic.sh train   ->  gpurun_out/pmc_<name>/summary.txt
set -uo pipefail
name="${1:-inference}"
out="$PWD/gpurun_out/pmc_$name"
mkdir -p "$out"
export TMPDIR=/tmp
repo="$PWD"
cd /tmp
args="--steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-}"
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/$c" -- \
        python "$repo/bench.py" $args > "$out/$c.log" 2>&1 || { echo "pass $c failed"; tail -n 5 "$out/$c.log"; exit 1; }
done
cd "$repo"
python tools/pmc_summary.py "$out" | tee "$out/summary.txt"
