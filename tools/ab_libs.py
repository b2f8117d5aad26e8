"""A/B of library builds (BFCNN_HIP_LIB) x bench args.  usage: python tools/ab_libs.py lib1.so lib2.so ... [-- bench args]"""
import json
import os
import subprocess
import sys

args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--")
    args, extra = args[:i], args[i + 1:]
for r in range(int(os.environ.get("AB_ROUNDS", "2"))):
    for lib in args:
        env = dict(os.environ)
        if lib != "default":
            env["BFCNN_HIP_LIB"] = os.path.abspath(lib)
        out = subprocess.run([sys.executable, "bench.py", "--steps", "10", "--warmup", "3", "--no-cpu-baseline"] + extra,
                             capture_output=True, text=True, env=env)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(f"[{lib}] FAILED rc={out.returncode}: {out.stderr[-300:]}")
            continue
        d = json.loads(line[-1])
        rf = d["roofline"]
        print(f"round {r} [{os.path.basename(lib):28s}] {d['value']:9.1f} img/s  block launch {rf['launch_us']:7.1f} us  frac {rf['frac']:.3f}", flush=True)
