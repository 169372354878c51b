"""The restated QuickLZ level-1 coder / decoder pair the compressed-vocabulary tests rest on (tests/qlz_ref.py): round
trips over the cases the format distinguishes -- short and long headers, stored chunks, 3-byte and long matches, runs
(the distance-1 match of a run is decoded from three bytes earlier), chunks that end inside a run."""
import numpy as np

import qlz_ref


def _cases():
    rng = np.random.default_rng(5)
    yield "short text", b"abcabcabcabcabc hello hello hello hello world world"
    yield "tiny", b"x"
    yield "nine", b"123456789"
    yield "random (stored)", rng.integers(0, 256, 5000, dtype=np.uint8).tobytes()
    yield "zeros", bytes(10000)
    yield "run in the middle", rng.integers(0, 256, 300, dtype=np.uint8).tobytes() + b"\x07" * 700 + rng.integers(0, 4, 500, dtype=np.uint8).tobytes()
    yield "low entropy", rng.integers(0, 3, 10000, dtype=np.uint8).tobytes()
    rec = b"".join(np.uint32(i).tobytes() + np.uint32(i // 6).tobytes() + np.float64(0.25).tobytes() + b"\x20\0\0\0\x01\0\0\0\0\0\0\0" +
                   rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for i in range(1, 190))
    yield "vocabulary-like records", rec
    yield "215 bytes (short header)", (b"ab" * 120)[:215]
    yield "216 bytes (long header)", (b"ab" * 120)[:216]
    yield "long matches", (rng.integers(0, 256, 400, dtype=np.uint8).tobytes()) * 12


def test_round_trips():
    for name, data in _cases():
        c = qlz_ref.compress(data)
        assert qlz_ref.decompress(c) == data, name
        if name in ("zeros", "low entropy", "long matches", "vocabulary-like records"):
            assert c[0] & 1 and len(c) < len(data), name  # really compressed
        if name.startswith("random"):
            assert not c[0] & 1 and len(c) == len(data) + 9, name  # stored behind a long header
        assert (c[0] & 2 != 0) == (len(data) >= 216), name


def test_chunked_body():
    rng = np.random.default_rng(6)
    body = rng.integers(0, 5, 25000, dtype=np.uint8).tobytes()
    framed = qlz_ref.dbow3_compressed_body(body)
    n = int.from_bytes(framed[:4], "little")
    assert n == 3
    at, out = 4, b""
    for _ in range(n):
        size = int.from_bytes(framed[at + 1:at + 5], "little") if framed[at] & 2 else framed[at + 1]
        out += qlz_ref.decompress(framed[at:at + size])
        at += size
    assert at == len(framed) and out == body
