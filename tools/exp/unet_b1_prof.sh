R=$PWD; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/up
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/up -o p -- python3 $R/tools/exp/unet_b1_prof.py > /dev/null 2>&1
f=$(find /tmp/up -name "*kernel_stats.csv" | head -1)
python3 -c "
import csv
rows=list(csv.DictReader(open('$f')))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print('kernel time per call us', tot/60/1e3, 'launches per call', sum(int(r['Calls']) for r in rows)/60)
for r in rows[:22]: print('%-64s %5.1f/call avg %6.1f us  %4.1f%%' % (r['Name'][:64], int(r['Calls'])/60, float(r['AverageNs'])/1e3, 100*int(r['TotalDurationNs'])/tot))
"
