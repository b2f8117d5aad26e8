#!/bin/bash
# instruction-mix PMC pass for the fused kernel
set -uo pipefail
out="$PWD/gpurun_out/pmc2"; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
repo="$PWD"; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES --output-format csv -d "$out/mix" -- python "$repo/bench.py" --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > "$out/mix.log" 2>&1 || tail -5 "$out/mix.log"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d "$out/act" -- python "$repo/bench.py" --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > "$out/act.log" 2>&1 || tail -5 "$out/act.log"
cd "$repo"; python tools/pmc_summary.py "$out" | grep -A20 fused_block
