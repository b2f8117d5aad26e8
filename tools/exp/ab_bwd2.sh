set -e
for i in 1 2; do
for o in 0 1; do
echo "== fused_bwd2=$o"; timeout -k 10 200 python bench.py --mode train --no-cpu-baseline --steps 30 --warmup 5 --opt train_fused_bwd2=$o | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
echo "== variant A0"; BFCNN_HIP_LIB=$PWD/blind_image_denoising_amd/lib/variants/libbfcnn_hip_BWD2_PREFETCH_A0.so timeout -k 10 200 python bench.py --mode train --no-cpu-baseline --steps 30 --warmup 5 --opt train_fused_bwd2=1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
