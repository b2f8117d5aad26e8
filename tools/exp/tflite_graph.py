"""Lists the operators of a TFLite flatbuffer (builtin code, input / output tensor names and shapes) without any
TFLite runtime: a minimal vtable reader for the handful of schema tables needed (Model, SubGraph, Tensor, Operator,
OperatorCode).  Used once to read the dataflow of the reference's exported `denoiser_model.tflite` (the graph revision
its one trained archive was written with); not part of the product or the tests.
usage: python tools/exp/tflite_graph.py model.tflite [name-filter]"""
import struct
import sys

OPS = {0: "ADD", 1: "AVERAGE_POOL_2D", 2: "CONCATENATION", 3: "CONV_2D", 4: "DEPTHWISE_CONV_2D", 6: "DEQUANTIZE", 9: "FULLY_CONNECTED",
       14: "LOGISTIC", 17: "MAX_POOL_2D", 18: "MUL", 19: "RELU", 22: "RESHAPE", 23: "RESIZE_BILINEAR", 25: "SOFTMAX", 28: "TANH",
       34: "PAD", 36: "GATHER", 39: "TRANSPOSE", 40: "MEAN", 41: "SUB", 42: "DIV", 43: "SQUEEZE", 45: "STRIDED_SLICE", 47: "EXP",
       53: "CAST", 55: "MAXIMUM", 57: "MINIMUM", 59: "NEG", 61: "GREATER", 65: "SLICE", 70: "EXPAND_DIMS", 73: "LOG", 74: "SUM",
       75: "SQRT", 76: "RSQRT", 77: "SHAPE", 78: "POW", 80: "FAKE_QUANT", 82: "REDUCE_MAX", 83: "PACK", 88: "UNPACK", 92: "SQUARE",
       94: "FILL", 97: "RESIZE_NEAREST_NEIGHBOR", 98: "LEAKY_RELU", 99: "SQUARED_DIFFERENCE", 101: "ABS", 104: "CEIL", 114: "QUANTIZE",
       117: "HARD_SWISH", 126: "BATCH_MATMUL", 130: "BROADCAST_TO", 150: "GELU", 32: "CUSTOM", 8: "FLOOR", 44: "UNIDIRECTIONAL_SEQUENCE_LSTM",
       90: "FLOOR_MOD", 102: "SPLIT_V", 49: "SPLIT", 118: "IF", 119: "WHILE"}


class FB:
    def __init__(self, b):
        self.b = b

    def u32(self, o): return struct.unpack_from("<I", self.b, o)[0]
    def i32(self, o): return struct.unpack_from("<i", self.b, o)[0]
    def u16(self, o): return struct.unpack_from("<H", self.b, o)[0]

    def root(self): return self.u32(0)

    def field(self, table, idx):
        """absolute offset of field idx of the table, or None."""
        vt = table - self.i32(table)
        vsize = self.u16(vt)
        slot = 4 + 2 * idx
        if slot >= vsize:
            return None
        off = self.u16(vt + slot)
        return table + off if off else None

    def indirect(self, o): return o + self.u32(o)

    def vector(self, table, idx):
        f = self.field(table, idx)
        if f is None:
            return 0, 0
        v = self.indirect(f)
        return self.u32(v), v + 4

    def tables(self, table, idx):
        n, base = self.vector(table, idx)
        return [self.indirect(base + 4 * i) for i in range(n)]

    def ints(self, table, idx):
        n, base = self.vector(table, idx)
        return [self.i32(base + 4 * i) for i in range(n)]

    def string(self, table, idx):
        f = self.field(table, idx)
        if f is None:
            return ""
        s = self.indirect(f)
        n = self.u32(s)
        return self.b[s + 4:s + 4 + n].decode(errors="replace")

    def scalar(self, table, idx, fmt, default=0):
        f = self.field(table, idx)
        return struct.unpack_from("<" + fmt, self.b, f)[0] if f is not None else default


def main(path, flt=None):
    fb = FB(open(path, "rb").read())
    model = fb.root()
    codes = []
    for oc in fb.tables(model, 1):                       # operator_codes
        dep = fb.scalar(oc, 0, "b")
        builtin = fb.scalar(oc, 3, "i")
        code = max(dep, builtin)
        codes.append(OPS.get(code, f"op{code}") + (":" + fb.string(oc, 1) if code == 32 else ""))
    for sg in fb.tables(model, 2):
        tensors = fb.tables(sg, 0)
        tname = [fb.string(t, 3) for t in tensors]
        tshape = [fb.ints(t, 0) for t in tensors]
        tbuf = [fb.scalar(t, 2, "I") for t in tensors]
        ttype = [fb.scalar(t, 1, "b") for t in tensors]
        print("subgraph", fb.string(sg, 4), "inputs", fb.ints(sg, 1), "outputs", fb.ints(sg, 2))
        short = lambda s: s.replace("hydra/unet_laplacian_backbone/unet_laplacian/", "").replace("hydra/", "")[:70]
        for i, op in enumerate(fb.tables(sg, 3)):
            code = codes[fb.scalar(op, 0, "I")]
            ins, outs = fb.ints(op, 1), fb.ints(op, 2)
            line = f"{i:4d} {code:18s} " + ", ".join(f"t{j}{tshape[j]}" for j in outs) + " <- " + \
                   ", ".join((f"t{j}" + ("" if j < 0 else f"{tshape[j]}:{short(tname[j]).split(';')[0]}")) for j in ins)
            if flt is None or flt in line:
                print(line)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
