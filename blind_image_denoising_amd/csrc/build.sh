#!/bin/bash
# Builds libbfcnn_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).  The translation units are compiled in
# parallel (device code never crosses a unit: no -fgpu-rdc) and linked into one shared library.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../lib"
obj="$out/obj"
mkdir -p "$out" "$obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
units=(conv3x3_c16 fused_h3 edge_layers train_ops pyramid augment loss_terms unet_ops unet_h3 unet_h3_enc unet_h3_first engine)
pids=()
for u in "${units[@]}"; do
    "$HIPCC" -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-pass-failed "$@" \
        -c "$here/$u.hip" -o "$obj/$u.o" &
    pids+=($!)
done
fail=0
for p in "${pids[@]}"; do wait "$p" || fail=1; done
[ "$fail" -eq 0 ] || { echo "compilation failed" >&2; exit 1; }
objs=()
for u in "${units[@]}"; do objs+=("$obj/$u.o"); done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$out/libbfcnn_hip.so"
echo "built $out/libbfcnn_hip.so"
