# (the two-channel kernel this script compared was not kept: see profiles/r04_dwconv_rows_variants.txt and git history)
# uo_dwconv_ln_rows(2)_kernel variants: two channels per lane with rows requested 4 (default) / 3 / 2 ahead, and the four-channel form
# (UO_DWR2=0, rows two ahead): rocprofv3 average of the 64-channel 5x5 instance inside bench.py --mode unet, and ms per forward
export TMPDIR=/tmp
repo="$PWD"
for v in "" UO_DWR2_PD3 UO_DWR2_PD2 UO_DWR20; do
  if [ -n "$v" ]; then export BFCNN_HIP_LIB=$repo/blind_image_denoising_amd/lib/variants/libbfcnn_hip_$v.so; fi
  rm -rf $repo/gpurun_out/dwr_prof
  ( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $repo/gpurun_out/dwr_prof -- python $repo/bench.py --mode unet --steps 10 --warmup 3 --no-cpu-baseline > $repo/gpurun_out/dwr.json 2>/dev/null )
  echo "${v:-default(2ch,PD4)} $(grep -h 'uo_dwconv_ln_rows' $repo/gpurun_out/dwr_prof/*/*kernel_stats.csv | grep '64, 5, 32' | sed 's/(float const[^\"]*\"/\"/' | awk -F'",' '{print $2}' | cut -d, -f1-3) ms_per_step $(tail -n 1 $repo/gpurun_out/dwr.json | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))')"
done
