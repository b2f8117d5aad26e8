#!/bin/bash
# Builds timing-only ablation variants of the library into blind_image_denoising_amd/lib/variants/:
#   ABLATE_MACRO=H3_ABLATE tools/ablate.sh 1 2 4     ->  lib/variants/libbfcnn_hip_H3_ABLATE<v>.so
# Goes through csrc/build.sh (ONE unit list); a failed compile fails the script, nothing is hidden.
set -euo pipefail
cd "$(dirname "$0")/.."
macro="${ABLATE_MACRO:-BF_ABLATE}"
out="$PWD/blind_image_denoising_amd/lib/variants"
mkdir -p "$out"
for v in "$@"; do
    BF_BUILD_OUT="$out" BF_BUILD_NAME="libbfcnn_hip_${macro}${v}.so" bash blind_image_denoising_amd/csrc/build.sh "-D${macro}=${v}"
done
ls -la "$out"
