set -e
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/t_all.log 2>&1; tail -n 3 gpurun_out/t_all.log
BF_BENCH_REHEARSE=1 timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-sub-records > gpurun_out/reh_inf.json 2> gpurun_out/reh_inf.err; tail -n 1 gpurun_out/reh_inf.json | cut -c1-400
BF_BENCH_REHEARSE=1 timeout -k 10 300 python bench.py --gpus 2 --mode train --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/reh_train.json 2> gpurun_out/reh_train.err; tail -n 1 gpurun_out/reh_train.json | cut -c1-400
