#!/bin/bash
# One GPU-box session that produces the files of profiles/ for a round: for every bench mode the JSON line of a plain run
# and the rocprofv3 --kernel-trace --stats summary of the SAME command; the kernel-stats CSV is picked by CONTENT (it must
# name the mode's dominant kernel), never by pid or directory order.  Usage: bash tools/gpu_profile_round.sh r02 [modes...]
set -uo pipefail
tag="${1:-rXX}"; shift || true
modes=("$@"); [ ${#modes[@]} -gt 0 ] || modes=(inference train unet unet56 pyramid)
repo="$PWD"
out="$repo/gpurun_out/profiles_$tag"
mkdir -p "$out"
export TMPDIR=/tmp
# fp32_b128 / fp32_b64: the exact-fp32 arithmetic at the bench batch and at the north star's batch (the ">= 60 % MFMA roofline" figure)
declare -A ARGS=( [inference]="" [train]="--mode train" [unet]="--mode unet" [unet56]="--mode unet --unet-graph v5.6" [pyramid]="--mode pyramid" [generic]="--mode generic"
                  [fp32_b128]="--arith 0 --no-sub-records" [fp32_b64]="--arith 0 --batch 64 --no-sub-records" [onepair0]="--opt h3_pair=0 --no-sub-records" )
declare -A KERNEL=( [inference]="fused_block2_h3w_kernel" [train]="bwd_block_h3t_kernel" [unet]="uh_enc32u_kernel" [unet56]="uh_enc32u_kernel" [pyramid]="lap_split_kernel" [generic]="ug_bneck_kernel"
                    [fp32_b128]="fused_block_v4_kernel" [fp32_b64]="fused_block_v4_kernel" [onepair0]="fused_block_h3v_kernel" )
rc_all=0
for m in "${modes[@]}"; do
    a="${ARGS[$m]}"
    echo "== $m: bench"
    timeout -k 10 400 python bench.py $a --steps 30 --warmup 5 > "$out/${tag}_${m}_bench.json" 2> "$out/${tag}_${m}_bench.err" || { echo "bench $m failed"; tail -n 5 "$out/${tag}_${m}_bench.err"; rc_all=1; continue; }
    tail -c 600 "$out/${tag}_${m}_bench.json"; echo
    echo "== $m: rocprofv3 --kernel-trace --stats"
    rm -rf "$out/prof_$m"
    ( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_$m" -- \
        python "$repo/bench.py" $a --steps 10 --warmup 3 --no-cpu-baseline --no-sub-records > "$out/${tag}_${m}_profiled_bench.json" 2> "$out/prof_$m.err" ) || { echo "profile $m failed"; tail -n 5 "$out/prof_$m.err"; rc_all=1; continue; }
    picked=""
    while IFS= read -r f; do
        if grep -q "${KERNEL[$m]}" "$f"; then picked="$f"; break; fi
    done < <(find "$out/prof_$m" -name "*kernel_stats.csv" | sort)
    if [ -z "$picked" ]; then echo "no kernel_stats.csv names ${KERNEL[$m]}"; rc_all=1; continue; fi
    cp "$picked" "$out/${tag}_${m}_kernel_stats.csv"
    head -n 6 "$out/${tag}_${m}_kernel_stats.csv" | cut -c1-160
    # the summary's AverageNs includes the warm-up calls (cold caches, clocks ramping): the same kernel over the TIMED calls only --
    # the last (steps x launches per step) of the trace -- with its median, next to it
    trace="${picked%_kernel_stats.csv}_kernel_trace.csv"
    [ -f "$trace" ] && python "$repo/tools/kstats_timed.py" "$trace" "${KERNEL[$m]}" > "$out/${tag}_${m}_kernel_timed.json" && cat "$out/${tag}_${m}_kernel_timed.json"
    rm -rf "$out/prof_$m"
done
exit $rc_all
