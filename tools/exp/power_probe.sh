#!/bin/bash
# Socket power and shader clock (rocm-smi, sampled every 0.2 s) under a few steady workloads: which part of the fused block's
# power budget is matrix work, which is memory traffic.  Usage (GPU box): bash tools/exp/power_probe.sh > gpurun_out/power.log
V=blind_image_denoising_amd/lib/variants
watch() {   # watch <label> <command...>
    label="$1"; shift
    "$@" > /tmp/pp_out.txt 2>&1 &
    pid=$!
    sleep 1.0
    : > /tmp/pp_samples.txt
    while kill -0 $pid 2>/dev/null; do
        /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)" | sed 's/.*: //' | tr '\n' ' ' >> /tmp/pp_samples.txt
        echo >> /tmp/pp_samples.txt
        sleep 0.2
    done
    python3 - "$label" <<'PY'
import re, sys, statistics
rows = []
for l in open('/tmp/pp_samples.txt'):
    m = re.search(r'\((\d+)Mhz\)\s+([\d.]+)', l)
    if m: rows.append((int(m.group(1)), float(m.group(2))))
rows = rows[1:-1] if len(rows) > 4 else rows
if rows:
    print(f"{sys.argv[1]:44s} samples {len(rows):3d}  sclk median {statistics.median(r[0] for r in rows):6.0f} MHz  power median {statistics.median(r[1] for r in rows):7.1f} W  max {max(r[1] for r in rows):7.1f} W")
else:
    print(f"{sys.argv[1]:44s} no samples")
PY
    tail -n 1 /tmp/pp_out.txt | cut -c1-200
}
watch "MFMA only (16x16x32 f16, 12 waves/CU)" tools/exp/mfma_mix 0 160
watch "MFMA + VALU + LDS mix" tools/exp/mfma_mix 7 120
watch "device copy 537 MB tensors" python -c "
import torch
x = torch.randn(128 * 256 * 256 * 16, device='cuda'); y = torch.empty_like(x)
for _ in range(16000): y.copy_(x)
torch.cuda.synchronize(); print('copy done')"
watch "device read (sum) 537 MB" python -c "
import torch
x = torch.randn(128 * 256 * 256 * 16, device='cuda')
for _ in range(24000): x.sum()
torch.cuda.synchronize(); print('sum done')"
watch "device fill 537 MB" python -c "
import torch
x = torch.empty(128 * 256 * 256 * 16, device='cuda')
for _ in range(30000): x.fill_(1.0)
torch.cuda.synchronize(); print('fill done')"
watch "bench default (two blocks per launch)" python bench.py --steps 700 --warmup 5 --no-cpu-baseline --no-sub-records
watch "bench h3_pair=0 (one block per launch)" python bench.py --steps 700 --warmup 5 --no-cpu-baseline --no-sub-records --opt h3_pair=0
BFCNN_HIP_LIB=$PWD/$V/libbfcnn_hip_H3V_ABLATE3.so watch "two blocks, no DMA / stores" python bench.py --steps 900 --warmup 5 --no-cpu-baseline --no-sub-records
BFCNN_HIP_LIB=$PWD/$V/libbfcnn_hip_H3V_ABLATE12.so watch "two blocks, no MFMA" python bench.py --steps 1200 --warmup 5 --no-cpu-baseline --no-sub-records
