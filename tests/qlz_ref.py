"""QuickLZ 1.5.0, compression level 1, streaming buffer 0 -- coder and decoder restated from the published algorithm
(quicklz.c: qlz_compress_core / qlz_decompress_core), and DBoW3's chunk framing of a compressed vocabulary stream
(Vocabulary::toStream: uint32 chunk count, then the qlz_compress output of every 10000 bytes of the stream).
TEST INFRASTRUCTURE: this writes the files the loader's QuickLZ path is tested with (csrc/dataset_io.cpp); neither QuickLZ
nor DBoW3 is available here, so the format is UNPINNED -- self-consistent with the restated decoder, not checked against a
file written by the library itself."""
import struct

_HASH = 4096
_UNCOND, _UNCOMP_END, _MINOFF = 6, 4, 2


def _rd3(b, p):
    n = len(b)
    return (b[p] if p < n else 0) | ((b[p + 1] if p + 1 < n else 0) << 8) | ((b[p + 2] if p + 2 < n else 0) << 16)


def _hash(i):
    return ((i >> 12) ^ i) & (_HASH - 1)


def _compress_core(source):
    size = len(source)
    last_byte = size - 1
    src = 0
    out = bytearray(4)
    cword_ptr, cword_val = 0, 1 << 31
    last_matchstart = last_byte - _UNCOND - _UNCOMP_END
    off, cache = [0] * _HASH, [0] * _HASH
    lits = 0
    fetch = _rd3(source, src) if src <= last_matchstart else 0

    def flush():
        nonlocal cword_ptr, cword_val
        out[cword_ptr:cword_ptr + 4] = struct.pack("<I", (cword_val >> 1) | (1 << 31))
        cword_ptr = len(out)
        out.extend(b"\0\0\0\0")
        cword_val = 1 << 31

    while src <= last_matchstart:
        if cword_val & 1:
            if src > (size >> 1) and len(out) > src - (src >> 5):
                return None  # not compressible enough: the chunk is stored
            flush()
            fetch = _rd3(source, src)
        h = _hash(fetch)
        cached = fetch ^ cache[h]
        cache[h] = fetch
        o = off[h]
        off[h] = src
        same6 = src >= 3 and len(set(source[src - 3:src + 3])) == 1
        if cached == 0 and o != 0 and (src - o > _MINOFF or (src == o + 1 and lits >= 3 and src > 3 and same6)):
            cword_val = (cword_val >> 1) | (1 << 31)
            if source[o + 3] != source[src + 3]:
                out.extend(struct.pack("<H", (3 - 2) | (h << 4)))
                src += 3
            else:
                old = src
                src += 4
                if source[o + (src - old)] == source[src]:
                    src += 1
                    if source[o + (src - old)] == source[src]:
                        q = last_byte - _UNCOMP_END - (src - 5) + 1
                        remaining = min(q, 255)
                        src += 1
                        while source[o + (src - old)] == source[src] and (src - old) < remaining:
                            src += 1
                mlen = src - old
                if mlen < 18:
                    out.extend(struct.pack("<H", (mlen - 2) | (h << 4)))
                else:
                    out.extend(struct.pack("<I", (mlen << 16) | (h << 4))[:3])
            fetch = _rd3(source, src)
            lits = 0
        else:
            lits += 1
            out.append(source[src])
            src += 1
            cword_val >>= 1
            fetch = _rd3(source, src)
    while src <= last_byte:
        if cword_val & 1:
            flush()
        out.append(source[src])
        src += 1
        cword_val >>= 1
    while (cword_val & 1) != 1:
        cword_val >>= 1
    out[cword_ptr:cword_ptr + 4] = struct.pack("<I", (cword_val >> 1) | (1 << 31))
    while len(out) < 9 - 0:  # qlz_compress_core reports at least 9 bytes
        out.append(0)
    return bytes(out)


def compress(data):
    """qlz_compress of one buffer -> header + payload"""
    data = bytes(data)
    size = len(data)
    assert 0 < size < (1 << 31)
    base = 3 if size < 216 else 9
    core = _compress_core(data)
    payload, compressed = (core, 1) if core is not None else (data, 0)
    total = base + len(payload)
    flags = compressed | (1 << 2) | (1 << 6)  # level 1, streaming buffer 0
    if base == 9:
        return bytes([flags | 2]) + struct.pack("<II", total, size) + payload
    assert total < 256
    return bytes([flags, total, size]) + payload


def decompress(chunk):
    """qlz_decompress of one chunk (the decoder csrc/dataset_io.cpp mirrors)"""
    flags = chunk[0]
    hdr = 9 if flags & 2 else 3
    csize, dsize = (struct.unpack("<II", chunk[1:9]) if flags & 2 else (chunk[1], chunk[2]))
    assert csize == len(chunk)
    if not flags & 1:
        return bytes(chunk[hdr:hdr + dsize])
    assert (flags >> 2) & 3 == 1
    out = bytearray(dsize)
    table = [-1] * _HASH
    last = dsize - 1
    last_matchstart = last - _UNCOND - _UNCOMP_END
    dst, sp, last_hashed, cword = 0, hdr, -1, 1
    bitlut = [4, 0, 1, 0, 2, 0, 1, 0, 3, 0, 1, 0, 2, 0, 1, 0]

    def rd(p, n):
        return int.from_bytes(chunk[p:p + n].ljust(n, b"\0"), "little")

    def hash_upto(mx):
        nonlocal last_hashed
        while last_hashed < mx:
            last_hashed += 1
            if last_hashed + 2 <= last:
                table[_hash(_rd3(out, last_hashed))] = last_hashed

    while True:
        if cword == 1:
            cword = rd(sp, 4)
            sp += 4
        fetch = rd(sp, 4)
        if cword & 1:
            cword >>= 1
            h = (fetch >> 4) & 0xfff
            frm = table[h]
            if fetch & 0xf:
                mlen, sp = (fetch & 0xf) + 2, sp + 2
            else:
                mlen, sp = (fetch >> 16) & 0xff, sp + 3
            assert 0 <= frm < dst and dst + mlen <= dsize
            for i in range(mlen):
                out[dst + i] = out[frm + i]
            dst += mlen
            hash_upto(dst - mlen)
            last_hashed = dst - 1
        elif dst < last_matchstart:
            n = bitlut[cword & 0xf]
            out[dst:dst + n] = chunk[sp:sp + n]
            cword >>= n
            dst += n
            sp += n
            hash_upto(dst - 3)
        else:
            while dst <= last:
                if cword == 1:
                    sp += 4
                    cword = 1 << 31
                out[dst] = chunk[sp]
                dst += 1
                sp += 1
                cword >>= 1
            return bytes(out)


def dbow3_compressed_body(body, chunk_size=10000):
    """what Vocabulary::toStream writes behind (magic, compressed = true, node count): chunk count + chunks"""
    chunks = [compress(body[i:i + chunk_size]) for i in range(0, len(body), chunk_size)]
    return struct.pack("<I", len(chunks)) + b"".join(chunks)
