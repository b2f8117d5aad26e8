"""Duration of ONE kernel in steady state from a profiled bench run (rocprofv3 --kernel-trace CSV).  The summary's AverageNs also holds the
warm-up steps (cold caches, ramping clocks) and the small launches of the parity crop, which is why it read 3-5 % above the HIP-event
figure of the same run.  Here: the launches of the kernel's dominant grid size, without the first fifth of them (the warm-up steps).
usage: kstats_timed.py <kernel_trace.csv> <kernel name substring>  ->  one JSON line"""
import collections, csv, json, statistics, sys

path, name = sys.argv[1], sys.argv[2]
rows = [r for r in csv.DictReader(open(path)) if name in r["Kernel_Name"]]
by_grid = collections.defaultdict(list)
for r in rows:
    by_grid[(r["Grid_Size_X"], r["Workgroup_Size_X"])].append(r)
grid, full = max(by_grid.items(), key=lambda kv: sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in kv[1]))
full.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in full]
skip = max(3, len(dur) // 5)
steady = dur[skip:] if len(dur) > skip + 2 else dur
print(json.dumps({"kernel": name, "calls_in_trace": len(rows), "calls_of_the_dominant_grid": len(dur), "grid": grid[0], "skipped_first": skip,
                  "all_calls_average_ns": sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / max(len(rows), 1),
                  "steady_average_ns": sum(steady) / len(steady), "steady_median_ns": statistics.median(steady),
                  "steady_min_ns": min(steady), "steady_max_ns": max(steady)}))
