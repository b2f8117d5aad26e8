# the decoder chain kernel (uh_chain32_kernel: the three level-0 decoder blocks and their node in one launch) on and off, same box,
# bench.py --mode unet, both graphs; ms per forward of batch 32 x 512 x 512
for g in v5 v5.6; do
for i in 1 2; do
for v in 1 0; do
  echo "$g fuse_chain=$v $(timeout -k 10 200 python bench.py --mode unet --unet-graph $g --no-cpu-baseline --opt fuse_chain=$v 2>/dev/null | tail -n 1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))') ms"
done
done
done
