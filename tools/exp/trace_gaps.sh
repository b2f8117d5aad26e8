# kernel-trace of a bench mode: busy time, span and the largest idle gaps between consecutive kernels of the timed steps
R=$PWD; MODE_ARGS="$@"; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/tg
rocprofv3 --kernel-trace --output-format csv -d /tmp/tg -o p -- python3 $R/bench.py $MODE_ARGS --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
f=$(find /tmp/tg -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows) // 2:]                      # the second half: timed steps
span = int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows)
gaps = [(int(rows[i + 1]['Start_Timestamp']) - int(rows[i]['End_Timestamp']), rows[i]['Kernel_Name'][:40], rows[i + 1]['Kernel_Name'][:40]) for i in range(len(rows) - 1)]
pos = [g for g in gaps if g[0] > 0]
print(f"kernels {len(rows)}  span {span / 1e6:.3f} ms  busy {busy / 1e6:.3f} ms  idle {100 * (span - busy) / span:.1f} %  gaps>0: {len(pos)}  mean gap {sum(g[0] for g in pos) / max(len(pos), 1) / 1e3:.2f} us")
from collections import Counter
c = Counter()
for g in pos: c[(g[1], g[2])] += g[0]
for (a, b), t in c.most_common(8): print(f"  {t / 1e3:8.1f} us total  after {a} -> before {b}")
PY
