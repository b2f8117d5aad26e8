# (timing-only ablations of the one-tile kernel on one box: as built 457 us; without the max reduction and its eight barriers 432 us; with the
# weight fragments as 16-byte loads on top of that 380-522 us, bimodal -- the workgroup preamble is at most a tenth of the kernel: no
# prepared-operand entry point was added)
# uf_first_conv_kernel<5> (the unet's first convolution): one tile per workgroup (default) against 512 persistent workgroups with the next
# tile's patch requested a tile ahead (-D'UF_GRID(k)=512'), SAME box, alternating: rocprofv3 average of the kernel's full-size calls inside
# bench.py --mode unet, and ms per forward
export TMPDIR=/tmp
repo="$PWD"
for i in 1 2 3; do
for v in "" ORIG; do
  if [ -n "$v" ]; then export BFCNN_HIP_LIB="$repo/blind_image_denoising_amd/lib/variants/libbfcnn_hip_$v.so"; else unset BFCNN_HIP_LIB; fi
  rm -rf $repo/gpurun_out/fc_prof
  ( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $repo/gpurun_out/fc_prof -- python $repo/bench.py --mode unet --steps 10 --warmup 3 --no-cpu-baseline > $repo/gpurun_out/fc.json 2>/dev/null )
  python - "$repo" "${v:-one tile per workgroup}" <<'PY'
import csv, glob, sys, json
repo, name = sys.argv[1], sys.argv[2]
f = glob.glob(repo + '/gpurun_out/fc_prof/*/*kernel_trace.csv')[0]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(f)) if 'first_conv' in r['Kernel_Name']]
big = sorted(x for x in d if x > 100)
ms = json.loads(open(repo + '/gpurun_out/fc.json').read().strip().split('\n')[-1])['ms_per_step']
print(f"{(name if name != chr(0) else name):28s} first conv median {big[len(big)//2]:.0f} us  min {big[0]:.0f}  max {big[-1]:.0f}   forward {ms:.3f} ms")
PY
done
done
