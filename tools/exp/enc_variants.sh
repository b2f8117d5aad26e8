set -e
V=$PWD/blind_image_denoising_amd/lib/variants
timeout -k 10 600 python -m pytest tests/test_gpu_unet.py -q -x 2>&1 | tail -2
for i in 1 2; do
echo -n "enc new "; timeout -k 10 120 python tools/exp/enc_ablate.py 2>&1 | tail -1
echo -n "enc old "; BFCNN_HIP_LIB=$V/libenc_old.so timeout -k 10 120 python tools/exp/enc_ablate.py 2>&1 | tail -1
echo -n "unet new "; timeout -k 10 200 python bench.py --mode unet --no-cpu-baseline --steps 20 --warmup 5 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
echo -n "unet mlp-old "; BFCNN_HIP_LIB=$V/libmlp_old.so timeout -k 10 200 python bench.py --mode unet --no-cpu-baseline --steps 20 --warmup 5 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
