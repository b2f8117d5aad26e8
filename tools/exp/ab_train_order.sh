# A/B of the XCD-contiguous tile order of bwd_block_h3t_kernel inside bench.py --mode train (ms per step, us per launch)
for i in 1 2; do
  timeout -k 10 200 python bench.py --mode train --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('xcd order  ', round(d['ms_per_step'],3), round(d['roofline']['launch_us'],1))"
  BFCNN_HIP_LIB=blind_image_denoising_amd/lib/variants/libbfcnn_hip_H3U_XCD_ORDER0.so timeout -k 10 200 python bench.py --mode train --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round robin', round(d['ms_per_step'],3), round(d['roofline']['launch_us'],1))"
done
