"""Two-stream view of ONE data-parallel training step from a rocprofv3 --kernel-trace CSV of
`bench.py --mode train --force-collective` (one GPU: the C ABI's all-reduce in a one-rank RCCL group, on the communicator's own stream):
the kernels between the last backward-block launch of a step and the first optimizer kernel, with stream, start and end relative to the
all-reduce's start.  Shows the gradient exchange running BESIDE the next batch's corruption (noise_augment_kernel) on the compute stream.
usage: dp_trace.py <kernel_trace.csv>"""
import csv, sys

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ar = [i for i, r in enumerate(rows) if any(k in r["Kernel_Name"].lower() for k in ("nccl", "rccl", "allreduce", "all_reduce"))]
if not ar:
    names = sorted({r["Kernel_Name"].split("(")[0][:60] for r in rows})
    sys.exit("no collective kernel in the trace; kernels: " + " | ".join(names))
i = ar[len(ar) // 2]                                   # a step from the middle of the run
t0 = int(rows[i]["Start_Timestamp"])
lo = max(0, i - 6)
print(f"# kernels around all-reduce #{len(ar) // 2} of {len(ar)}; times in us relative to its start; stream / queue as rocprofv3 reports them")
print(f"{'stream':>6s} {'queue':>5s} {'start':>9s} {'end':>9s} {'dur':>8s}  kernel")
for r in rows[lo:i + 8]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{r['Stream_Id']:>6s} {r['Queue_Id']:>5s} {s:9.1f} {e:9.1f} {e - s:8.1f}  {r['Kernel_Name'][:90]}")
a_s, a_e = 0.0, (int(rows[i]["End_Timestamp"]) - t0) / 1e3
ov = [r for r in rows[lo:i + 8] if r is not rows[i] and int(r["Start_Timestamp"]) < int(rows[i]["End_Timestamp"]) and int(r["End_Timestamp"]) > t0]
print(f"# all-reduce: {a_e:.1f} us; kernels overlapping it in time: " + (", ".join(sorted({r['Kernel_Name'].split('(')[0][:40] for r in ov})) or "none"))
