set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for o in 0 1; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bwd2_$o -o p -- python3 $R/bench.py --mode train --no-cpu-baseline --steps 20 --warmup 5 --opt train_fused_bwd2=$o > $R/gpurun_out/prof_bwd2_$o.log 2>&1
done
