# (results, one box: 1 -> 11.00 ms, 2 -> 10.89-10.92, 3 (default) -> 10.84, 4 -> 10.90-10.92, 6 -> 10.91-10.93; placing the weight-fragment
# ds_reads in the pattern as well -- one per 2 / 3 MFMAs -- 11.50-11.59 against 11.23-11.28, and 20 minutes of compile time per unit: not kept)
# uh_mlp_core: vector instructions the scheduler places per MFMA (sched_group_barrier) in the MLP / chain kernels of unet_h3.hip (the encoder
# kernels of unet_h3_enc.hip keep the default 3); bench.py --mode unet, ms per forward, same box
for i in 1 2; do
for v in "" UH_MLP_VALU_PER_MFMA1 UH_MLP_VALU_PER_MFMA2 UH_MLP_VALU_PER_MFMA4 UH_MLP_VALU_PER_MFMA6; do
  if [ -n "$v" ]; then export BFCNN_HIP_LIB=$PWD/blind_image_denoising_amd/lib/variants/libbfcnn_hip_$v.so; else unset BFCNN_HIP_LIB; fi
  echo "${v:-default(3)} $(timeout -k 10 200 python bench.py --mode unet --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))') ms"
done
done
