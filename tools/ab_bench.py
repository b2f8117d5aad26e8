"""Runs bench.py once per variant (separate processes, same box) and prints one compact line each.
usage: python tools/ab_bench.py "--fused-tile 0" "--fused-tile 1" ...   (extra common args via AB_ARGS)"""
import json
import os
import subprocess
import sys

common = os.environ.get("AB_ARGS", "--steps 10 --warmup 3 --no-cpu-baseline").split()
rounds = int(os.environ.get("AB_ROUNDS", "2"))
for r in range(rounds):
    for variant in sys.argv[1:]:
        out = subprocess.run([sys.executable, "bench.py"] + common + variant.split(), capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(f"[{variant}] FAILED rc={out.returncode}: {out.stderr[-400:]}")
            continue
        d = json.loads(line[-1])
        rf = d["roofline"]
        print(f"round {r} [{variant:24s}] {d['value']:9.1f} img/s  {d['ms_per_step']:7.3f} ms/step  block launch {rf['launch_us']:7.1f} us  "
              f"{rf['achieved']:6.1f} TF  frac {rf['frac']:.3f}  max_lsb {d['parity']['max_abs_lsb']}", flush=True)
