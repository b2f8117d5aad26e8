#!/bin/bash
# Builds the C restatement (and nothing of the reference: it is Python on TensorFlow and has no
# compilable sources, so there is no oracle/_ref) into oracle/_build/libbfcnn_port.so.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
mkdir -p "$here/_build"
march="${BFCNN_PORT_MARCH:-native}"
gcc -O3 -march="$march" -ffp-contract=fast -fopenmp -fPIC -shared "$here/bfcnn_port.c" -o "$here/_build/libbfcnn_port.so" -lm
echo "built $here/_build/libbfcnn_port.so (-march=$march)"
