set -e
timeout -k 10 600 python -m pytest tests/test_gpu_inference.py -q -x 2>&1 | tail -n 2
for cfg in "32 512" "8 1024" "16 384"; do set -- $cfg
for p in 1 0; do echo -n "B=$1 size=$2 h3_pair=$p: "; timeout -k 10 300 python bench.py --batch $1 --size $2 --no-cpu-baseline --no-sub-records --steps 30 --warmup 5 --opt h3_pair=$p 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'])"; done; done
