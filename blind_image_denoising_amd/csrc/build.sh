#!/bin/bash
# Builds libbfcnn_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).  The translation units are compiled in
# parallel (device code never crosses a unit: no -fgpu-rdc) and linked into one shared library.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
# BF_BUILD_OUT / BF_BUILD_NAME: another directory / file name for the result (tools/ablate.sh builds timing variants with
# extra -D flags passed as arguments; the unit list below is the only one)
out="${BF_BUILD_OUT:-$here/../lib}"
name="${BF_BUILD_NAME:-libbfcnn_hip.so}"
obj="$out/obj${BF_BUILD_NAME:+_${BF_BUILD_NAME%.so}}"
mkdir -p "$out" "$obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
units=(conv3x3_c16 fused_h3 fused_h3v fused_h3w train_bwd_h3 train_bwd2_h3 train_fwd_h3t train_bwd_h3t edge_layers base_rows train_ops pyramid augment loss_terms unet_ops unet_h3 unet_h3_chain unet_h3_enc unet_h3_first generic_h3 train_prims train_generic collective engine)
# incremental: a unit is recompiled when its source, any header or the flag set is newer than / differs from its object
flags_sig="$*"
[ -f "$obj/.flags" ] && [ "$(cat "$obj/.flags")" = "$flags_sig" ] || { rm -f "$obj"/*.o; printf '%s' "$flags_sig" > "$obj/.flags"; }
newest_hdr="$(ls -t "$here"/*.h "$here/../../include/bfcnn_hip.h" "$here/../../include/bfcnn_hip_debug.h" "${BASH_SOURCE[0]}" | head -1)"
pids=()
for u in "${units[@]}"; do
    if [ -f "$obj/$u.o" ] && [ "$obj/$u.o" -nt "$here/$u.hip" ] && [ "$obj/$u.o" -nt "$newest_hdr" ]; then continue; fi
    "$HIPCC" -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-pass-failed "$@" \
        -c "$here/$u.hip" -o "$obj/$u.o" &
    pids+=($!)
done
fail=0
for p in "${pids[@]}"; do wait "$p" || fail=1; done
[ "$fail" -eq 0 ] || { echo "compilation failed" >&2; exit 1; }
objs=()
for u in "${units[@]}"; do objs+=("$obj/$u.o"); done
printf '%s\n' "${units[@]}" > "$obj/.units"          # tools/ablate_unit.sh links exactly these objects (never a stale one)
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -ldl -o "$out/$name"
echo "built $out/$name"
