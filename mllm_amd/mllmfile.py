"""Reader / writer for the reference's ``.mllm`` weight container.

On-disk layout (reference: mllm/ParamLoader.cpp:14-31 diagram, index parse :157-286; writer
tools/quantizer/ParamWriter.cpp:25-84; python writer tools/convertor/converter.py:23-118)::

    int32  magic = 20012
    uint64 index_len                     # bytes of the index that follows
    repeat until index_len consumed:
        int32  name_len ; bytes name
        uint64 data_len ; uint64 file_offset ; int32 dtype
    raw tensor bytes (on-disk == in-memory block layout, e.g. block_q4_K = 144 B / 256 weights)

Only what the hot path needs lives here: this is the "weights enter here" boundary.
"""
from __future__ import annotations

import mmap
import struct
from dataclasses import dataclass
from typing import Dict, Iterable, List, Tuple

import numpy as np

MAGIC = 20012  # mllm/ParamLoader.hpp:48

# mllm/Types.hpp:63-97 (DataType enum values that can appear in the five configs' files)
F32, F16, Q4_0, Q8_0, Q4_K, Q6_K, Q8_K = 0, 1, 2, 8, 12, 14, 15
DTYPE_NAME = {F32: "F32", F16: "F16", Q4_0: "Q4_0", Q8_0: "Q8_0", Q4_K: "Q4_K", Q6_K: "Q6_K", Q8_K: "Q8_K"}
# (block elements, block bytes): mllm/DataType.hpp:75-78,93-98,137-140,159-163
BLOCK = {F32: (1, 4), F16: (1, 2), Q4_0: (32, 18), Q8_0: (32, 34), Q4_K: (256, 144), Q8_K: (256, 292)}


def nbytes(dtype: int, n_elem: int) -> int:
    be, bb = BLOCK[dtype]
    assert n_elem % be == 0
    return n_elem // be * bb


@dataclass
class Entry:
    name: str
    offset: int
    length: int
    dtype: int


class MllmFile:
    """mmap-backed reader (zero copy, like ParamLoader's alloc_mmap path, ParamLoader.cpp:124-128)."""

    def __init__(self, path: str):
        self.path = path
        self._f = open(path, "rb")
        self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        magic, index_len = struct.unpack_from("<iQ", self._mm, 0)
        if magic != MAGIC:
            raise ValueError(f"{path}: magic number error ({magic})")
        pos, end = 12, 12 + index_len
        self.entries: Dict[str, Entry] = {}
        while pos < end:
            (nl,) = struct.unpack_from("<i", self._mm, pos)
            pos += 4
            name = self._mm[pos:pos + nl].decode()
            pos += nl
            length, offset, dtype = struct.unpack_from("<QQi", self._mm, pos)
            pos += 20
            self.entries[name] = Entry(name, offset, length, dtype)

    def names(self) -> List[str]:
        return list(self.entries)

    def raw(self, name: str) -> np.ndarray:
        e = self.entries[name]
        return np.frombuffer(self._mm, dtype=np.uint8, count=e.length, offset=e.offset)

    def f32(self, name: str) -> np.ndarray:
        e = self.entries[name]
        assert e.dtype == F32, (name, e.dtype)
        return np.frombuffer(self._mm, dtype=np.float32, count=e.length // 4, offset=e.offset)

    def dtype(self, name: str) -> int:
        return self.entries[name].dtype

    def close(self):
        self._mm.close()
        self._f.close()


def write_mllm(path: str, tensors: Iterable[Tuple[str, int, bytes | np.ndarray]]) -> None:
    """Write ``(name, dtype, raw-bytes-or-array)`` triples. Streams tensor by tensor (index first, padded)."""
    items = []
    tensors = list(tensors) if not isinstance(tensors, list) else tensors
    index_len = sum(4 + len(n.encode()) + 8 + 8 + 4 for n, _, _ in tensors)
    with open(path, "wb") as f:
        f.write(struct.pack("<iQ", MAGIC, index_len))
        f.write(b"\0" * index_len)
        for name, dtype, data in tensors:
            buf = data.tobytes() if isinstance(data, np.ndarray) else bytes(data)
            items.append((name, f.tell(), len(buf), dtype))
            f.write(buf)
        f.seek(12)
        for name, off, ln, dtype in items:
            nb = name.encode()
            f.write(struct.pack("<i", len(nb)) + nb + struct.pack("<QQi", ln, off, dtype))
