# LayerNorm lane sums of the split-f16 MLP / chain kernels: v_permlane swaps (default) against ds_bpermute (UH_LN_PERMLANE=0); same box
for g in v5; do
for i in 1 2; do
for v in "" UH_LN_PERMLANE0; do
  if [ -n "$v" ]; then export BFCNN_HIP_LIB=$PWD/blind_image_denoising_amd/lib/variants/libbfcnn_hip_$v.so; else unset BFCNN_HIP_LIB; fi
  echo "$g ${v:-permlane} $(timeout -k 10 200 python bench.py --mode unet --unet-graph $g --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))') ms"
done
done
done
