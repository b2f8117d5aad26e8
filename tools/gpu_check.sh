#!/bin/bash
# One GPU-box session: parity tests -> smoke -> bench -> rocprofv3 kernel trace.
# Usage (from the repo root on the GPU box): bash tools/gpu_check.sh [tests|bench|prof|all]
set -uo pipefail
what="${1:-all}"
out=gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
run_tests() {
    timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout 300 --maxfail 40 -p no:cacheprovider \
        > "$out/pytest_gpu.log" 2>&1
    rc=$?
    tail -n 60 "$out/pytest_gpu.log"
    echo "pytest exit code: $rc"
    # 0 = green, 1 = test failures (keep going, we want the bench too); anything else = stop
    [ $rc -le 1 ]
}
run_smoke() {
    timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee "$out/smoke.log"
}
run_bench() {
    timeout -k 10 600 python bench.py --steps 10 --warmup 3 2> "$out/bench.err" | tee "$out/bench.json"
}
run_prof() {
    rm -rf "$out/prof"
    cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$out/prof" -- \
        python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$OLDPWD/$out/prof_bench.log" 2>&1
    rc=$?
    cd "$OLDPWD"
    find "$out/prof" -name "*kernel_stats.csv" | head -n 1 | xargs -r head -n 25
    return $rc
}
case "$what" in
    tests) run_tests ;;
    bench) run_bench ;;
    prof) run_prof ;;
    all) run_tests && run_smoke && run_bench && run_prof ;;
esac
