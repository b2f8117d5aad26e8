# bench.py --mode train with the BatchNorm finalisations folded into the block kernels' prologues (default) and as kernels of their own
for i in 1 2; do
  for v in 1 0; do
    timeout -k 10 200 python bench.py --mode train --no-cpu-baseline --opt train_fold_finalize=$v 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train_fold_finalize=$v', round(d['ms_per_step'],3), 'ms/step', round(d['roofline']['launch_us'],1), 'us per bwd launch')"
  done
done
