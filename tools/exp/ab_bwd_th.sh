set -e
V=$PWD/blind_image_denoising_amd/lib/variants/libbfcnn_hip_BWD_H3_TH8.so
BFCNN_HIP_LIB=$V timeout -k 10 300 python -m pytest tests/test_gpu_training.py -q -x -k "matches_oracle and not random" 2>&1 | tail -n 2
for i in 1 2; do
echo -n "TH=16 "; timeout -k 10 200 python bench.py --mode train --no-cpu-baseline --steps 30 --warmup 5 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
echo -n "TH=8  "; BFCNN_HIP_LIB=$V timeout -k 10 200 python bench.py --mode train --no-cpu-baseline --steps 30 --warmup 5 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
