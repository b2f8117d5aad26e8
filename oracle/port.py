"""ctypes loader of the C port (oracle/bfcnn_port.c) -- test infrastructure, see its header."""
import ctypes as C
import pathlib
import subprocess

import numpy as np

_HERE = pathlib.Path(__file__).resolve().parent
_LIB = _HERE / "_build" / "libbfcnn_port.so"


def build(march: str = "native"):
    subprocess.run(["bash", str(_HERE / "build.sh")], check=True, env={**__import__("os").environ, "BFCNN_PORT_MARCH": march},
                   stdout=subprocess.DEVNULL)


def lib(rebuild: bool = False):
    if rebuild or not _LIB.exists():
        build()
    h = C.CDLL(str(_LIB))
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
