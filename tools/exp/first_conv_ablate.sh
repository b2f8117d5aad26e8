# uf_first_conv_tile_kernel<5>, timing-only ablations (UF_TILE_ABLATE builds): 1 LDS gathers from conflict-free addresses, 2 no stores, 4 no
# MFMAs, 7 all three; rocprofv3 median of the full-size calls inside bench.py --mode unet, one box
export TMPDIR=/tmp
repo="$PWD"
for v in "" UF_TILE_ABLATE1 UF_TILE_ABLATE2 UF_TILE_ABLATE4 UF_TILE_ABLATE7; do
  if [ -n "$v" ]; then export BFCNN_HIP_LIB="$repo/blind_image_denoising_amd/lib/variants/libbfcnn_hip_$v.so"; else unset BFCNN_HIP_LIB; fi
  rm -rf $repo/gpurun_out/fc_prof
  ( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $repo/gpurun_out/fc_prof -- python $repo/bench.py --mode unet --steps 10 --warmup 3 --no-cpu-baseline > $repo/gpurun_out/fc.json 2>/dev/null )
  python - "$repo" "${v:-as built}" <<'PY'
import csv, glob, sys
repo, name = sys.argv[1], sys.argv[2]
f = glob.glob(repo + '/gpurun_out/fc_prof/*/*kernel_trace.csv')[0]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(f)) if 'first_conv' in r['Kernel_Name']]
big = sorted(x for x in d if x > 60)
print(f"{name:20s} first conv median {big[len(big)//2]:.0f} us  min {big[0]:.0f}")
PY
done
