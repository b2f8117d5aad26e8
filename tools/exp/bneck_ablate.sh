for v in "" UG_ABLATE1 UG_ABLATE2 UG_ABLATE4 UG_ABLATE8 UG_ABLATE6; do
  if [ -n "$v" ]; then export BFCNN_HIP_LIB=blind_image_denoising_amd/lib/variants/libbfcnn_hip_$v.so; fi
  timeout -k 10 120 python bench.py --mode generic --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['roofline']['launch_us'],1), round(d['ms_per_step'],3))"
done
