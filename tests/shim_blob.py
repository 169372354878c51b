"""Named-array files exchanged with tests/shim_run/shim_run.cpp: "VOBN", int32 n, then per array int32 name length, name,
int32 dtype (0 u8, 1 i32, 2 f32, 3 f64), int32 ndim, int64 dims, data."""
import struct

import numpy as np

_DT = [np.uint8, np.int32, np.float32, np.float64]


def write(path, arrays):
    with open(path, "wb") as f:
        f.write(b"VOBN" + struct.pack("<i", len(arrays)))
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            code = [i for i, d in enumerate(_DT) if a.dtype == d]
            assert code, (name, a.dtype)
            nb = name.encode()
            f.write(struct.pack("<i", len(nb)) + nb + struct.pack("<ii", code[0], a.ndim))
            f.write(struct.pack(f"<{a.ndim}q", *a.shape))
            f.write(a.tobytes())


def read(path):
    out = {}
    with open(path, "rb") as f:
        assert f.read(4) == b"VOBN"
        (n,) = struct.unpack("<i", f.read(4))
        for _ in range(n):
            (ln,) = struct.unpack("<i", f.read(4))
            name = f.read(ln).decode()
            code, nd = struct.unpack("<ii", f.read(8))
            dims = struct.unpack(f"<{nd}q", f.read(8 * nd)) if nd else ()
            cnt = int(np.prod(dims)) if nd else 1
            out[name] = np.frombuffer(f.read(cnt * np.dtype(_DT[code]).itemsize), dtype=_DT[code]).reshape(dims).copy()
    return out
