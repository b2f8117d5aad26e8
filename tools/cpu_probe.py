"""How many host cores does the GPU box really give us, and how does the C port scale?"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except Exception as e:
        print(f, "n/a")
print(subprocess.run("lscpu | grep -E 'Model name|Socket|Core|Thread|^CPU\\(s\\)'", shell=True, capture_output=True, text=True).stdout)
code = r'''
import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
from oracle import bfcnn_oracle as O, port
spec = O.ResnetSpec.from_config(O.canonical_config(no_layers=18)["model"])
params, state = O.init_params(spec, seed=42)
_, noisy = O.synthetic_batch(16, 256, 256, seed=1)
h = port.lib(rebuild=True)
port.forward_u8(spec, params, state, noisy[:2], h)
t = time.perf_counter(); port.forward_u8(spec, params, state, noisy, h); dt = time.perf_counter() - t
print("OMP_NUM_THREADS=%s: %.2f img/s" % (os.environ.get("OMP_NUM_THREADS"), 16 / dt))
'''
for n in (8, 16, 32, 64, 128, 256):
    env = dict(os.environ, OMP_NUM_THREADS=str(n), OMP_PROC_BIND="close")
    print(subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env).stdout.strip(), flush=True)
