"""Aggregates rocprofv3 --pmc CSVs (one row per dispatch and counter) into per-kernel averages."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?").split("(")[0]
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, counters in sorted(agg.items()):
    if not any(s in k for s in ("fused_block", "conv3x3", "base_conv", "head_kernel", "wgrad", "bwd3x3", "fwd_block", "bwd_block", "uh_", "uo_", "ug_", "uf_")):
        continue
    print(k)
    for c, vals in sorted(counters.items()):
        # max: the full-size dispatches when a run also holds small ones (parity crops)
        print(f"   {c:32s} avg/dispatch {sum(vals) / len(vals):16.1f}   max {max(vals):16.1f}   dispatches {len(vals)}")
