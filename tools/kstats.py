"""Prints a rocprofv3 kernel_stats.csv compactly (newest file under the given directory)."""
import csv, glob, os, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
files = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(files[-1])))
print(files[-1])
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print(f"{r['Name'][:64]:64s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f} us {float(r['Percentage']):5.1f}%")
