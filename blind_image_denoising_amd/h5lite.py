"""Minimal read-only HDF5 reader for Keras weight files (`model.weights.h5` inside a `.keras` archive, or a legacy
`.h5` weights file): superblock version 0, old-style groups (symbol-table B-trees + local heaps), version-1 object
headers, contiguous or compact little-endian float / integer datasets.  That is what h5py writes for the uncompressed,
unchunked arrays Keras 2.x saves (bfcnn/export_model.py:106-110 and `model.save`); anything else raises
`NotImplementedError` with the feature named.  No h5py in this image, hence this file."""
import struct
from typing import Dict, Iterator, Tuple, Union

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5File:
    """`H5File(bytes_or_path)`; `datasets()` yields (path, numpy array); `tree()` the nested dict of names."""

    def __init__(self, source: Union[bytes, str]):
        if isinstance(source, (bytes, bytearray, memoryview)):
            self.b = bytes(source)
        else:
            with open(source, "rb") as f:
                self.b = f.read()
        b = self.b
        if b[:8] != SIGNATURE:
            raise ValueError("not an HDF5 file")
        version = b[8]
        if version not in (0, 1):
            raise NotImplementedError(f"HDF5 superblock version {version}")
        if b[13] != 8 or b[14] != 8:
            raise NotImplementedError("HDF5 offsets / lengths that are not 8 bytes")
        pos = 24 + (4 if version == 1 else 0)            # v1 adds indexed-storage K + reserved
        self.base = struct.unpack_from("<Q", b, pos)[0]
        root_entry = pos + 32                              # base, free-space, EOF, driver info addresses
        self.root = self._symbol_entry(root_entry)

    # ---- low level --------------------------------------------------------------------------------
    def _u(self, fmt: str, off: int):
        return struct.unpack_from("<" + fmt, self.b, off)

    def _symbol_entry(self, off: int) -> Dict:
        name_off, header, cache = self._u("QQI", off)
        e = {"name_off": name_off, "header": header + self.base}
        if cache == 1:
            btree, heap = self._u("QQ", off + 24)
            e["btree"], e["heap"] = btree + self.base, heap + self.base
        return e

    def _messages(self, addr: int) -> Iterator[Tuple[int, int, int]]:
        """(type, data offset, size) of every message of the version-1 object header at addr."""
        b = self.b
        if b[addr] != 1:
            raise NotImplementedError(f"object header version {b[addr]} (new-style groups / h5py libver='latest')")
        nmsg, = self._u("H", addr + 2)
        size, = self._u("I", addr + 8)
        blocks = [(addr + 16, size)]
        seen = 0
        while blocks and seen < nmsg:
            off, left = blocks.pop(0)
            end = off + left
            while off + 8 <= end and seen < nmsg:
                mtype, msize = self._u("HH", off)
                data = off + 8
                seen += 1
                if mtype == 0x0010:                        # continuation
                    coff, clen = self._u("QQ", data)
                    blocks.append((coff + self.base, clen))
                else:
                    yield mtype, data, msize
                off = data + msize

    def _heap_name(self, heap: int, off: int) -> str:
        if self.b[heap:heap + 4] != b"HEAP":
            raise ValueError("bad local heap")
        seg, = self._u("Q", heap + 24)
        start = seg + self.base + off
        end = self.b.index(b"\x00", start)
        return self.b[start:end].decode()

    def _group_children(self, btree: int, heap: int) -> Iterator[Tuple[str, Dict]]:
        b = self.b
        if b[btree:btree + 4] != b"TREE":
            raise ValueError("bad group B-tree node")
        ntype, level, used = b[btree + 4], b[btree + 5], self._u("H", btree + 6)[0]
        if ntype != 0:
            raise NotImplementedError("chunk B-tree where a group B-tree was expected")
        pos = btree + 24
        for i in range(used):
            child, = self._u("Q", pos + 8 + 16 * i)
            child += self.base
            if level > 0:
                yield from self._group_children(child, heap)
                continue
            if b[child:child + 4] != b"SNOD":
                raise ValueError("bad symbol table node")
            nsym, = self._u("H", child + 6)
            for j in range(nsym):
                e = self._symbol_entry(child + 8 + 40 * j)
                yield self._heap_name(heap, e["name_off"]), e

    def _group_of(self, entry: Dict):
        """(btree, heap) if the object is a group, else None."""
        if "btree" in entry:
            return entry["btree"], entry["heap"]
        for mtype, data, _ in self._messages(entry["header"]):
            if mtype == 0x0011:
                btree, heap = self._u("QQ", data)
                return btree + self.base, heap + self.base
        return None

    def _dataset(self, header: int) -> np.ndarray:
        shape, dtype, layout = None, None, None
        for mtype, data, size in self._messages(header):
            b = self.b
            if mtype == 0x0001:                            # dataspace
                ver, rank, flags = b[data], b[data + 1], b[data + 2]
                off = data + (8 if ver == 1 else 4)
                shape = tuple(self._u("Q", off + 8 * i)[0] for i in range(rank))
            elif mtype == 0x0003:                          # datatype
                cls, bits0 = b[data] & 15, b[data + 1]
                nbytes, = self._u("I", data + 4)
                if bits0 & 1:
                    raise NotImplementedError("big-endian HDF5 datatype")
                if cls == 1:
                    dtype = {2: np.float16, 4: np.float32, 8: np.float64}[nbytes]
                elif cls == 0:
                    signed = bool(bits0 & 8)
                    dtype = np.dtype(f"<{'i' if signed else 'u'}{nbytes}")
                else:
                    raise NotImplementedError(f"HDF5 datatype class {cls}")
            elif mtype == 0x0008:                          # layout
                ver, cls = b[data], b[data + 1]
                if ver != 3:
                    raise NotImplementedError(f"HDF5 data layout version {ver}")
                if cls == 1:
                    addr, nbytes = self._u("QQ", data + 2)
                    layout = (None if addr == UNDEF else addr + self.base, nbytes)
                elif cls == 0:
                    nbytes, = self._u("H", data + 2)
                    layout = (data + 4, nbytes)
                else:
                    raise NotImplementedError("chunked HDF5 dataset (compression / resizable arrays)")
            elif mtype == 0x000B:
                raise NotImplementedError("HDF5 filter pipeline (compressed dataset)")
        if shape is None or dtype is None or layout is None:
            raise ValueError("object is not a dataset")
        n = int(np.prod(shape)) if shape else 1
        if layout[0] is None:
            return np.zeros(shape, dtype)
        return np.frombuffer(self.b, dtype=dtype, count=n, offset=layout[0]).reshape(shape).copy()

    # ---- public -------------------------------------------------------------------------------------
    def _walk(self, entry: Dict, prefix: str) -> Iterator[Tuple[str, np.ndarray]]:
        grp = self._group_of(entry)
        if grp is None:
            yield prefix, self._dataset(entry["header"])
            return
        for name, child in self._group_children(*grp):
            yield from self._walk(child, f"{prefix}/{name}" if prefix else name)

    def datasets(self) -> Iterator[Tuple[str, np.ndarray]]:
        """every dataset as (path, array), in the file's own (alphabetical B-tree) order."""
        yield from self._walk(self.root, "")


def read_keras_archive(path: str) -> Dict[str, np.ndarray]:
    """{dataset path: array} of the `model.weights.h5` member of a `.keras` zip archive."""
    import zipfile
    with zipfile.ZipFile(path) as z:
        return dict(H5File(z.read("model.weights.h5")).datasets())
