#!/bin/bash
# Builds libbfcnn_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../lib"
mkdir -p "$out"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared \
    -Wall -Wno-unused-function \
    "$here/conv3x3_c16.hip" "$here/fused_h3.hip" "$here/edge_layers.hip" "$here/train_ops.hip" "$here/pyramid.hip" "$here/augment.hip" "$here/unet_ops.hip" "$here/unet_h3.hip" "$here/engine.hip" \
    -o "$out/libbfcnn_hip.so" "$@"
echo "built $out/libbfcnn_hip.so"
