#!/bin/bash
# Timing-only variants of ONE translation unit: recompiles <unit>.hip with -D<MACRO>=<v> and links it with the current
# objects of the other units (blind_image_denoising_amd/lib/obj, i.e. run csrc/build.sh first).
#   tools/ablate_unit.sh fused_h3v H3V_ABLATE 1 2 3 12 32   ->  lib/variants/libbfcnn_hip_H3V_ABLATE<v>.so
set -euo pipefail
cd "$(dirname "$0")/.."
unit="$1"; macro="$2"; shift 2
src=blind_image_denoising_amd/csrc
obj=blind_image_denoising_amd/lib/obj
out=blind_image_denoising_amd/lib/variants
mkdir -p "$out"
# the unit list csrc/build.sh wrote with its last build: an object left behind by a renamed or removed unit is never linked
[ -f "$obj/.units" ] || { echo "run blind_image_denoising_amd/csrc/build.sh first ($obj/.units is missing)" >&2; exit 1; }
others=()
while IFS= read -r u; do [ "$u" = "$unit" ] || others+=("$obj/$u.o"); done < "$obj/.units"
pids=()
for v in "$@"; do
    (
        /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-pass-failed \
            "-D${macro}=${v}" -c "$src/$unit.hip" -o "$out/${unit}_${macro}${v}.o"
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "${others[@]}" "$out/${unit}_${macro}${v}.o" -ldl \
            -o "$out/libbfcnn_hip_${macro}${v}.so"
    ) &
    pids+=($!)
done
fail=0
for p in "${pids[@]}"; do wait "$p" || fail=1; done
[ "$fail" -eq 0 ] || { echo "a variant failed to build" >&2; exit 1; }
ls -la "$out"/*.so
