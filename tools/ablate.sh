#!/bin/bash
# Builds timing-only ablation variants of the library into blind_image_denoising_amd/lib/variants/.
set -euo pipefail
cd "$(dirname "$0")/.."
src=blind_image_denoising_amd/csrc
out=blind_image_denoising_amd/lib/variants
mkdir -p "$out"
for v in "$@"; do
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -D${ABLATE_MACRO:-BF_ABLATE}=$v \
        $src/conv3x3_c16.hip $src/fused_h3.hip $src/edge_layers.hip $src/train_ops.hip $src/pyramid.hip $src/augment.hip $src/unet_ops.hip $src/unet_h3.hip $src/unet_h3_enc.hip $src/engine.hip \
        -o "$out/libbfcnn_hip_${ABLATE_MACRO:-BF_ABLATE}$v.so" 2>/dev/null &
done
wait
ls -la "$out"
