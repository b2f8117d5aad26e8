#!/bin/bash
# HBM traffic of the unet_laplacian kernels: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (per the guide).
set -uo pipefail
out="$PWD/gpurun_out/pmc_unet"
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
repo="$PWD"
cd /tmp
pass() {
    name="$1"; shift
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/$name" -- \
        python "$repo/bench.py" --mode unet --steps 2 --warmup 1 --no-cpu-baseline > "$out/$name.log" 2>&1 || { echo "pass $name failed"; tail -n 5 "$out/$name.log"; return 1; }
}
pass fetch FETCH_SIZE && pass write WRITE_SIZE && pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT &&
pass sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM && pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
cd "$repo"
python tools/pmc_summary.py "$out" > "$out/summary.txt"
grep -A22 "uh_enc32\|uh_mlp_kernel<32" "$out/summary.txt" | head -60
