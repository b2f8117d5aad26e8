"""ctypes loader of the C port (oracle/bfcnn_port.c) -- test infrastructure, see its header."""
import ctypes as C
import pathlib
import subprocess

import math
import os

import numpy as np

_HERE = pathlib.Path(__file__).resolve().parent
_LIB = _HERE / "_build" / "libbfcnn_port.so"


def build(march: str = "native"):
    subprocess.run(["bash", str(_HERE / "build.sh")], check=True, env={**__import__("os").environ, "BFCNN_PORT_MARCH": march},
                   stdout=subprocess.DEVNULL)


def effective_cores() -> int:
    """host cores this process may really use: scheduler affinity capped by the cgroup CPU quota
    (a GPU box gives one GPU's share, e.g. cpu.max = "1600000 100000" = 16 CPUs of 256 visible)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except Exception:
        pass
    return n


def lib(rebuild: bool = False):
    if rebuild or not _LIB.exists():
        build()
    h = C.CDLL(str(_LIB))
    h.bfcnn_port_set_threads.argtypes = [C.c_int]
    h.bfcnn_port_max_threads.restype = C.c_int
    h.bfcnn_port_set_threads(int(os.environ.get("OMP_NUM_THREADS", effective_cores())))
    h.bfcnn_port_forward_u8.restype = C.c_int
    h.bfcnn_port_forward_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                        C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    return h


def forward_u8(spec, params, state, image_u8, handle=None):
    """DenoiserModule.__call__ through the C port (canonical resnet only)."""
    h = handle or lib()
    params = np.ascontiguousarray(params, np.float32)
    state = np.ascontiguousarray(state, np.float32)
    x = np.ascontiguousarray(image_u8, np.uint8)
    B, H, W, cin = x.shape
    out = np.empty((B, H, W, spec.out_channels), np.uint8)
    rc = h.bfcnn_port_forward_u8(params.ctypes.data, state.ctypes.data, spec.no_layers, spec.kernel_size, cin,
                                 spec.head_filters, spec.out_channels, spec.bn_eps, x.ctypes.data, out.ctypes.data, B, H, W)
    if rc != 0:
        raise RuntimeError(f"bfcnn_port_forward_u8 failed: {rc}")
    return out
