set -e
timeout -k 10 600 python -m pytest tests/test_gpu_inference.py -q -x 2>&1 | tail -n 2
sed -i 's/for cfg in .*; do/for cfg in "0 0" "4 0" "2 0" "1 0"; do/' tools/exp/base_rows_abl.sh
bash tools/exp/base_rows_abl.sh 2>&1 | tail -n 4
for i in 1 2; do timeout -k 10 200 python bench.py --no-cpu-baseline --no-sub-records --steps 50 --warmup 10 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
