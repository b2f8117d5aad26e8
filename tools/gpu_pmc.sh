#!/bin/bash
# PMC passes for the dominant kernel (separate rocprofv3 runs, --kernel-trace only, per the guide).
set -uo pipefail
out="$PWD/gpurun_out/pmc"
mkdir -p "$out"
export TMPDIR=/tmp
repo="$PWD"
cd /tmp
args="--steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-}"
pass() {
    name="$1"; shift
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/$name" -- \
        python "$repo/bench.py" $args > "$out/$name.log" 2>&1 || { echo "pass $name failed"; tail -n 5 "$out/$name.log"; return 1; }
}
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT &&
pass sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM &&
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT &&
pass fetch FETCH_SIZE &&
pass write WRITE_SIZE
cd "$repo"
python tools/pmc_summary.py "$out" | tee "$out/summary.txt"
