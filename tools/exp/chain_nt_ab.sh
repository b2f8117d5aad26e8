# uh_chain32_kernel without the node (fuse_up_block=0: uo_upsample_act_add_kernel + the plain chain) at 512 / 768 / 1024 threads per
# workgroup (2 / 3 / 4 waves per SIMD; ~125 registers) against the node-forming chain (512 threads, 196 registers); ms per forward
for i in 1 2; do
echo "node-forming chain (default)  $(timeout -k 10 200 python bench.py --mode unet --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))') ms"
for v in "" UH_CHAIN_NT768 UH_CHAIN_NT1024; do
  if [ -n "$v" ]; then export BFCNN_HIP_LIB=$PWD/blind_image_denoising_amd/lib/variants/libbfcnn_hip_$v.so; else unset BFCNN_HIP_LIB; fi
  echo "plain chain ${v:-UH_CHAIN_NT512} + upsample kernel  $(timeout -k 10 200 python bench.py --mode unet --no-cpu-baseline --opt fuse_up_block=0 2>/dev/null | tail -n 1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))') ms"
done
unset BFCNN_HIP_LIB
done
