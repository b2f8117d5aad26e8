# uo_dwconv_ln_rows_kernel: rows requested ahead (UO_DWR_PD builds), rocprofv3 average of the 64-channel 5x5 instance inside bench.py --mode unet
export TMPDIR=/tmp
repo="$PWD"
for v in "" UO_DWR_PD1 UO_DWR_PD3; do
  if [ -n "$v" ]; then export BFCNN_HIP_LIB=$repo/blind_image_denoising_amd/lib/variants/libbfcnn_hip_$v.so; fi
  rm -rf $repo/gpurun_out/dwr_prof
  ( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $repo/gpurun_out/dwr_prof -- python $repo/bench.py --mode unet --steps 10 --warmup 3 --no-cpu-baseline > $repo/gpurun_out/dwr.json 2>/dev/null )
  echo "${v:-default(PD2)} $(grep -h 'uo_dwconv_ln_rows_kernel<64, 5, 32>' $repo/gpurun_out/dwr_prof/*/*kernel_stats.csv | sed 's/(float const[^\"]*\"/\"/' | cut -d, -f1-4) ms_per_step $(tail -n 1 $repo/gpurun_out/dwr.json | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))')"
done
