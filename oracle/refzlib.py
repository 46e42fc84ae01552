"""ctypes binding of the *compiled reference* (oracle/_ref/libzref.so).

TEST INFRASTRUCTURE ONLY.  Nothing under zlib_amd/ may import this module.

The shared object is built by oracle/Makefile straight from the read-only mount
(/root/reference/qcsrc/*.c, zlib 1.2.3); it is git-ignored but travels to the GPU box with
the gpurun snapshot, so `available()` can be true there even though /root/reference is not.

The z_stream layout mirrors /root/reference/h/zlib.h:82-101 on LP64 (112 bytes).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libzref.so")

Z_NO_FLUSH, Z_PARTIAL_FLUSH, Z_SYNC_FLUSH, Z_FULL_FLUSH, Z_FINISH, Z_BLOCK = 0, 1, 2, 3, 4, 5
Z_OK, Z_STREAM_END, Z_NEED_DICT = 0, 1, 2
Z_ERRNO, Z_STREAM_ERROR, Z_DATA_ERROR, Z_MEM_ERROR, Z_BUF_ERROR, Z_VERSION_ERROR = -1, -2, -3, -4, -5, -6
Z_DEFLATED = 8
CHUNK = 65536


class ZStream(C.Structure):
    _fields_ = [
        ("next_in", C.c_void_p), ("avail_in", C.c_uint), ("total_in", C.c_ulong),
        ("next_out", C.c_void_p), ("avail_out", C.c_uint), ("total_out", C.c_ulong),
        ("msg", C.c_char_p), ("state", C.c_void_p),
        ("zalloc", C.c_void_p), ("zfree", C.c_void_p), ("opaque", C.c_void_p),
        ("data_type", C.c_int), ("adler", C.c_ulong), ("reserved", C.c_ulong),
    ]


assert C.sizeof(ZStream) == 112

_lib = None


def available():
    return os.path.exists(_SO)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_SO)
        L.deflateInit2_.argtypes = [C.POINTER(ZStream), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_char_p, C.c_int]
        L.deflateInit_.argtypes = [C.POINTER(ZStream), C.c_int, C.c_char_p, C.c_int]
        L.deflate.argtypes = [C.POINTER(ZStream), C.c_int]
        L.deflateEnd.argtypes = [C.POINTER(ZStream)]
        L.deflateSetDictionary.argtypes = [C.POINTER(ZStream), C.c_char_p, C.c_uint]
        L.inflateInit2_.argtypes = [C.POINTER(ZStream), C.c_int, C.c_char_p, C.c_int]
        L.inflate.argtypes = [C.POINTER(ZStream), C.c_int]
        L.inflateEnd.argtypes = [C.POINTER(ZStream)]
        L.compress2.argtypes = [C.c_char_p, C.POINTER(C.c_ulong), C.c_char_p, C.c_ulong, C.c_int]
        L.uncompress.argtypes = [C.c_char_p, C.POINTER(C.c_ulong), C.c_char_p, C.c_ulong]
        L.compressBound.argtypes = [C.c_ulong]
        L.compressBound.restype = C.c_ulong
        L.adler32.argtypes = [C.c_ulong, C.c_char_p, C.c_uint]
        L.adler32.restype = C.c_ulong
        L.adler32_combine.argtypes = [C.c_ulong, C.c_ulong, C.c_long]
        L.adler32_combine.restype = C.c_ulong
        L.zlibVersion.restype = C.c_char_p
        _lib = L
    return _lib


def version():
    return lib().zlibVersion().decode()


def compress2(data: bytes, level: int) -> bytes:
    L = lib()
    n = C.c_ulong(L.compressBound(len(data)) + 64)
    out = C.create_string_buffer(n.value)
    rc = L.compress2(out, C.byref(n), data, len(data), level)
    if rc != Z_OK:
        raise RuntimeError("reference compress2 rc=%d" % rc)
    return out.raw[: n.value]


def uncompress(data: bytes, outlen: int):
    L = lib()
    n = C.c_ulong(outlen)
    out = C.create_string_buffer(max(outlen, 1))
    rc = L.uncompress(out, C.byref(n), data, len(data))
    return rc, out.raw[: n.value]


def adler32(data: bytes, start: int = 1) -> int:
    L = lib()
    a = start
    for off in range(0, len(data), 1 << 30):
        part = data[off: off + (1 << 30)]
        a = L.adler32(a, part, len(part))
    return a


def crc32(data: bytes, start: int = 0) -> int:
    L = lib()
    L.crc32.argtypes = [C.c_ulong, C.c_char_p, C.c_uint]
    L.crc32.restype = C.c_ulong
    return L.crc32(start, data, len(data)) & 0xFFFFFFFF


def crc32_combine(c1: int, c2: int, len2: int) -> int:
    L = lib()
    L.crc32_combine.argtypes = [C.c_ulong, C.c_ulong, C.c_long]
    L.crc32_combine.restype = C.c_ulong
    return L.crc32_combine(c1, c2, len2) & 0xFFFFFFFF


def deflate_wbits(data: bytes, level: int, wbits: int, strategy: int = 0) -> bytes:
    """One-shot deflate with the reference at the given windowBits (31: gzip wrapper)."""
    L = lib()
    s = ZStream()
    rc = L.deflateInit2_(C.byref(s), level, Z_DEFLATED, wbits, 8, strategy, b"1.2.3", C.sizeof(ZStream))
    assert rc == Z_OK, rc
    cap = len(data) + (len(data) >> 8) + 1024
    out = C.create_string_buffer(cap)
    inb = C.create_string_buffer(data, max(len(data), 1))
    s.next_in = C.addressof(inb); s.avail_in = len(data)
    s.next_out = C.addressof(out); s.avail_out = cap
    rc = L.deflate(C.byref(s), Z_FINISH)
    assert rc == Z_STREAM_END, rc
    n = s.total_out
    L.deflateEnd(C.byref(s))
    return out.raw[:n]


def inflate_wbits(data: bytes, wbits: int, outcap: int):
    """One-shot inflate with the reference at the given windowBits; returns (rc, bytes, consumed, msg, strm.adler)."""
    L = lib()
    s = ZStream()
    rc = L.inflateInit2_(C.byref(s), wbits, b"1.2.3", C.sizeof(ZStream))
    assert rc == Z_OK, rc
    out = C.create_string_buffer(max(outcap, 1))
    inb = C.create_string_buffer(data, max(len(data), 1))
    s.next_in = C.addressof(inb); s.avail_in = len(data)
    s.next_out = C.addressof(out); s.avail_out = outcap
    rc = L.inflate(C.byref(s), Z_FINISH)
    res = (rc, out.raw[:s.total_out], s.total_in, s.msg.decode() if s.msg else None, s.adler & 0xFFFFFFFF)
    L.inflateEnd(C.byref(s))
    return res


def deflate_chunk_raw(chunk: bytes, level: int, is_last: bool, pos0_matchable: bool = False, strategy: int = 0) -> bytes:
    """The per-chunk function F(bytes, level, pos0_matchable, is_last) of SURVEY.md section 8c, computed
    by the real reference: a fresh raw stream (windowBits=-15, memLevel=8, default strategy) fed the
    whole chunk, finished with Z_FINISH (last chunk) or Z_FULL_FLUSH (any other chunk).

    pos0_matchable=True reproduces "mode A" chunks k>=1 (the chunk does not start at window index 0, so
    its first position is not the NIL sentinel): a 3-byte preset dictionary -- legal on a raw stream in
    1.2.3 (reference deflate.c:325-328) -- shifts the chunk to window index 3.  Only dictionary position
    0 is ever inserted (deflate.c:349-351) and position 0 is NIL, so the junk bytes are unmatchable.
    """
    L = lib()
    s = ZStream()
    rc = L.deflateInit2_(C.byref(s), level, Z_DEFLATED, -15, 8, strategy, b"1.2.3", C.sizeof(ZStream))
    if rc != Z_OK:
        raise RuntimeError("deflateInit2_ rc=%d" % rc)
    if pos0_matchable:
        rc = L.deflateSetDictionary(C.byref(s), b"\x00\x01\x02", 3)
        if rc != Z_OK:
            raise RuntimeError("deflateSetDictionary rc=%d" % rc)
    cap = len(chunk) + (len(chunk) >> 8) + 256
    out = C.create_string_buffer(cap)
    inb = C.create_string_buffer(chunk, max(len(chunk), 1))
    s.next_in = C.addressof(inb)
    s.avail_in = len(chunk)
    s.next_out = C.addressof(out)
    s.avail_out = cap
    rc = L.deflate(C.byref(s), Z_FINISH if is_last else Z_FULL_FLUSH)
    want = Z_STREAM_END if is_last else Z_OK
    if rc != want or s.avail_in != 0:
        L.deflateEnd(C.byref(s))
        raise RuntimeError("reference deflate rc=%d avail_in=%d" % (rc, s.avail_in))
    n = s.total_out
    L.deflateEnd(C.byref(s))
    return out.raw[:n]


def deflate_chunk_dict_raw(dictionary: bytes, chunk: bytes, level: int, is_last: bool, strategy: int = 0) -> bytes:
    """F with a preset dictionary, by the reference: fresh raw stream, deflateSetDictionary, the chunk, Z_FINISH / Z_FULL_FLUSH."""
    L = lib()
    s = ZStream()
    rc = L.deflateInit2_(C.byref(s), level, Z_DEFLATED, -15, 8, strategy, b"1.2.3", C.sizeof(ZStream))
    assert rc == Z_OK, rc
    rc = L.deflateSetDictionary(C.byref(s), dictionary, len(dictionary))
    assert rc == Z_OK, rc
    cap = len(chunk) + (len(chunk) >> 8) + 256
    out = C.create_string_buffer(cap)
    inb = C.create_string_buffer(chunk, max(len(chunk), 1))
    s.next_in = C.addressof(inb); s.avail_in = len(chunk)
    s.next_out = C.addressof(out); s.avail_out = cap
    rc = L.deflate(C.byref(s), Z_FINISH if is_last else Z_FULL_FLUSH)
    assert rc == (Z_STREAM_END if is_last else Z_OK) and s.avail_in == 0, rc
    n = s.total_out
    L.deflateEnd(C.byref(s))
    return out.raw[:n]


def zlib_header(level: int) -> bytes:
    """2-byte zlib header for windowBits=15, no dictionary (reference deflate.c:625-641)."""
    hdr = (Z_DEFLATED + ((15 - 8) << 4)) << 8
    if level < 2:
        lf = 0
    elif level < 6:
        lf = 1
    elif level == 6:
        lf = 2
    else:
        lf = 3
    hdr |= lf << 6
    hdr += 31 - (hdr % 31)
    return bytes([hdr >> 8, hdr & 0xFF])


def deflate_mode_b(data: bytes, level: int, chunk: int = CHUNK) -> bytes:
    """Mode B stream: header + F(chunk_k) concatenated + big-endian adler32."""
    parts = [zlib_header(level)]
    n = len(data)
    nchunks = max(1, (n + chunk - 1) // chunk)
    for k in range(nchunks):
        parts.append(deflate_chunk_raw(data[k * chunk:(k + 1) * chunk], level, k == nchunks - 1))
    parts.append(adler32(data).to_bytes(4, "big"))
    return b"".join(parts)


def deflate_mode_a(data: bytes, level: int, chunk: int = CHUNK) -> bytes:
    """Mode A: one zlib stream, deflate(chunk, Z_FULL_FLUSH) per chunk, Z_FINISH on the last."""
    L = lib()
    s = ZStream()
    rc = L.deflateInit_(C.byref(s), level, b"1.2.3", C.sizeof(ZStream))
    assert rc == Z_OK
    cap = len(data) + (len(data) >> 8) + 1024
    out = C.create_string_buffer(cap)
    inb = C.create_string_buffer(data, max(len(data), 1))
    s.next_out = C.addressof(out)
    s.avail_out = cap
    n = len(data)
    nchunks = max(1, (n + chunk - 1) // chunk)
    for k in range(nchunks):
        lo = k * chunk
        hi = min(n, lo + chunk)
        s.next_in = C.addressof(inb) + lo
        s.avail_in = hi - lo
        last = k == nchunks - 1
        rc = L.deflate(C.byref(s), Z_FINISH if last else Z_FULL_FLUSH)
        assert rc == (Z_STREAM_END if last else Z_OK), rc
    total = s.total_out
    L.deflateEnd(C.byref(s))
    return out.raw[:total]


def inflate_raw(data: bytes, outcap: int):
    """Raw inflate (windowBits=-15) of a whole buffer; returns (rc, bytes, consumed, msg)."""
    L = lib()
    s = ZStream()
    rc = L.inflateInit2_(C.byref(s), -15, b"1.2.3", C.sizeof(ZStream))
    assert rc == Z_OK
    out = C.create_string_buffer(max(outcap, 1))
    inb = C.create_string_buffer(data, max(len(data), 1))
    s.next_in = C.addressof(inb)
    s.avail_in = len(data)
    s.next_out = C.addressof(out)
    s.avail_out = outcap
    rc = L.inflate(C.byref(s), Z_FINISH)
    n = s.total_out
    used = s.total_in
    msg = s.msg.decode() if s.msg else None
    L.inflateEnd(C.byref(s))
    return rc, out.raw[:n], used, msg


def inflate_zlib(data: bytes, outcap: int):
    rc, out = uncompress(data, outcap)
    return rc, out


def deflate_calls(data: bytes, level: int, calls=(), wbits: int = -15, strategy: int = 0, dictionary: bytes = None, params=None) -> bytes:
    """ONE stream of the reference driven call by call: for (upto, flush) in calls, deflate() is handed data[fed:upto] with that flush
    value (Z_NO_FLUSH slices the input, Z_SYNC_FLUSH / Z_FULL_FLUSH / Z_PARTIAL_FLUSH flush); a last call hands over the rest with
    Z_FINISH.  calls == () is what compress2() does (qcsrc/compress.c:22-58).  Output space is never short."""
    L = lib()
    s = ZStream()
    rc = L.deflateInit2_(C.byref(s), level, Z_DEFLATED, wbits, 8, strategy, b"1.2.3", C.sizeof(ZStream))
    assert rc == Z_OK, rc
    if dictionary is not None:
        rc = L.deflateSetDictionary(C.byref(s), dictionary, len(dictionary))
        assert rc == Z_OK, rc
    cap = len(data) + (len(data) >> 3) + 64 * (len(calls) + 2) + 1024
    out = C.create_string_buffer(cap)
    inb = C.create_string_buffer(data, max(len(data), 1))
    s.next_out = C.addressof(out); s.avail_out = cap
    fed = 0
    L.deflateParams.argtypes = [C.POINTER(ZStream), C.c_int, C.c_int]
    for k, (upto, flush) in enumerate(list(calls) + [(len(data), Z_FINISH)]):
        if params and k in params:  # deflateParams() in front of call k (it may flush through next_out: nothing is pending on next_in)
            s.next_in = C.addressof(inb) + fed; s.avail_in = 0
            rc = L.deflateParams(C.byref(s), params[k][0], params[k][1])
            assert rc in (Z_OK, Z_BUF_ERROR), rc  # (Z_BUF_ERROR: its own Z_PARTIAL_FLUSH right behind another flush; the parameters are set all the same)
        s.next_in = C.addressof(inb) + fed; s.avail_in = upto - fed
        rc = L.deflate(C.byref(s), flush)
        want = Z_STREAM_END if flush == Z_FINISH else Z_OK
        if rc != want or s.avail_in != 0:
            L.deflateEnd(C.byref(s))
            raise RuntimeError("reference deflate rc=%d avail_in=%d (flush %d at %d)" % (rc, s.avail_in, flush, upto))
        fed = upto
    n = s.total_out
    L.deflateEnd(C.byref(s))
    return out.raw[:n]
