# base_conv_rows_kernel average duration (rocprofv3) for the timing switches BF_BASE_ROWS_ABL (1 no stores, 2 no loads) and BF_BASE_ROWS_BAND
set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp
for cfg in "0 0" "1 0" "2 0" "3 0" "0 8" "0 32" "0 64"; do
  set -- $cfg
  export BF_BASE_ROWS_ABL=$1
  if [ "$2" != "0" ]; then export BF_BASE_ROWS_BAND=$2; else unset BF_BASE_ROWS_BAND; fi
  rm -rf /tmp/brp
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/brp -o p -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sub-records > /dev/null 2>&1
  f=$(find /tmp/brp -name "*kernel_stats.csv" | sort | while read f; do if grep -q base_conv_rows "$f"; then echo $f; break; fi; done)
  echo "abl=$1 band=$2: $(python3 -c "import csv,sys; [print(r['Calls'], r['AverageNs']) for r in csv.DictReader(open('$f')) if 'base_conv_rows' in r['Name']]")"
done
