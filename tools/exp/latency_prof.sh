R=$PWD; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/lp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp -o p -- python3 $R/tools/exp/small_batch_variants.py > /dev/null 2>&1
f=$(find /tmp/lp -name "*kernel_stats.csv" | head -1)
python3 -c "
import csv
for r in list(csv.DictReader(open('$f')))[:12]: print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['MinNs'], r['MaxNs'])
"
