"""ctypes binding of this repo's CPU restatement (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY:
importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from zlib_amd/."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("ZAMD_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")  # (ZAMD_ORACLE_LIB: the sanitizer build, tests/test_sanitizers_cpu.py)
_lib = None


class Token(C.Structure):
    _fields_ = [("dist", C.c_uint16), ("lc", C.c_uint8), ("pad", C.c_uint8)]


class ChunkInfo(C.Structure):
    _fields_ = [("ntokens", C.c_uint32), ("nblocks", C.c_uint32), ("btype", C.c_uint32 * 8),
                ("data_type", C.c_uint32)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.ora_deflate_chunk.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                        C.c_void_p, C.c_void_p]
        L.ora_deflate_chunk.restype = C.c_size_t
        L.ora_deflate_chunk_s.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                          C.c_void_p, C.c_void_p]
        L.ora_deflate_chunk_s.restype = C.c_size_t
        L.ora_deflate_chunk_d.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                          C.c_void_p, C.c_void_p]
        L.ora_deflate_chunk_d.restype = C.c_size_t
        L.ora_deflate_stream_s.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t]
        L.ora_deflate_stream_s.restype = C.c_size_t
        L.ora_deflate_stream.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t]
        L.ora_deflate_stream.restype = C.c_size_t
        L.ora_deflate_cont.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.ora_deflate_cont.restype = C.c_size_t
        L.ora_deflate_cont_p.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.ora_deflate_cont_p.restype = C.c_size_t
        L.ora_deflate_bound.argtypes = [C.c_size_t, C.c_size_t]
        L.ora_deflate_bound.restype = C.c_size_t
        L.ora_adler32.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t]
        L.ora_adler32.restype = C.c_uint32
        L.ora_adler32_combine.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64]
        L.ora_adler32_combine.restype = C.c_uint32
        L.ora_crc32.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t]
        L.ora_crc32.restype = C.c_uint32
        L.ora_crc32_combine.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64]
        L.ora_crc32_combine.restype = C.c_uint32
        for f in (L.ora_inflate_raw, L.ora_inflate_zlib):
            f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                          C.POINTER(C.c_size_t), C.POINTER(C.c_char_p)]
            f.restype = C.c_int
        _lib = L
    return _lib


def deflate_chunk(chunk: bytes, level: int, is_last: bool, pos0_matchable: bool = False, want_tokens=False, strategy: int = 0):
    L = lib()
    cap = len(chunk) + 512
    out = C.create_string_buffer(cap)
    info = ChunkInfo()
    toks = (Token * max(len(chunk), 1))() if want_tokens else None
    n = L.ora_deflate_chunk_s(chunk, len(chunk), level, strategy, int(pos0_matchable), int(is_last), out, cap,
                            C.cast(toks, C.c_void_p) if want_tokens else None, C.byref(info))
    if n == 0:
        raise RuntimeError("oracle deflate_chunk failed")
    if want_tokens:
        return out.raw[:n], info, [(toks[i].dist, toks[i].lc) for i in range(info.ntokens)]
    return out.raw[:n]


def deflate_chunk_dict(dictionary: bytes, chunk: bytes, level: int, is_last: bool, strategy: int = 0) -> bytes:
    """The chunk function with a preset dictionary (its last MAX_DIST bytes count, deflate.c:336-339); len(chunk) + the
    dictionary bytes used must not exceed 65536."""
    d = dictionary[-32506:]
    if len(d) < 3:
        return deflate_chunk(chunk, level, is_last, strategy=strategy)
    buf = d + chunk
    cap = len(buf) + 512
    out = C.create_string_buffer(cap)
    n = lib().ora_deflate_chunk_d(buf, len(buf), len(d), level, strategy, 0, int(is_last), out, cap, None, None)
    if n == 0:
        raise RuntimeError("oracle deflate_chunk_d failed")
    return out.raw[:n]


def deflate_stream(data, level: int, chunk: int = 65536, strategy: int = 0) -> bytes:
    """data: bytes or a numpy uint8 array (no copy)."""
    L = lib()
    n = len(data)
    cap = L.ora_deflate_bound(n, chunk)
    out = C.create_string_buffer(cap)
    if isinstance(data, (bytes, bytearray)):
        src = C.cast(C.c_char_p(bytes(data)), C.c_void_p)
    else:
        src = C.c_void_p(data.ctypes.data)
    got = L.ora_deflate_stream_s(src, n, level, strategy, chunk, out, cap)
    if got == 0:
        raise RuntimeError("oracle deflate_stream failed")
    return out.raw[:got]


def adler32(data: bytes, start: int = 1) -> int:
    return lib().ora_adler32(start, data, len(data))


def adler32_combine(a1, a2, len2):
    return lib().ora_adler32_combine(a1, a2, len2)


def crc32(data: bytes, start: int = 0) -> int:
    return lib().ora_crc32(start, data, len(data))


def crc32_combine(c1, c2, len2):
    return lib().ora_crc32_combine(c1, c2, len2)


def gzip_header(level: int) -> bytes:
    """The 10 bytes deflate() writes for windowBits 31 when no gz_header was set (reference deflate.c:578-596); OS_CODE 3."""
    return bytes([31, 139, 8, 0, 0, 0, 0, 0, 2 if level == 9 else 4 if level < 2 else 0, 3])


def deflate_stream_gzip(data, level: int, chunk: int = 65536) -> bytes:
    """Mode B body in a gzip member: header, the raw chunk streams of deflate_stream, CRC-32 and length (deflate.c:833-843)."""
    z = deflate_stream(data, level, chunk)  # zlib-wrapped: 2-byte header, body, 4-byte Adler
    raw = bytes(data) if isinstance(data, (bytes, bytearray)) else data.tobytes()
    return gzip_header(level) + z[2:-4] + crc32(raw).to_bytes(4, "little") + (len(raw) & 0xFFFFFFFF).to_bytes(4, "little")


def _inflate(fn, data: bytes, outcap: int):
    out = C.create_string_buffer(max(outcap, 1))
    used, prod, msg = C.c_size_t(0), C.c_size_t(0), C.c_char_p()
    rc = fn(data, len(data), out, outcap, C.byref(used), C.byref(prod), C.byref(msg))
    return rc, out.raw[:prod.value], used.value, (msg.value.decode() if msg.value else None)


def inflate_raw(data: bytes, outcap: int):
    return _inflate(lib().ora_inflate_raw, data, outcap)


def inflate_zlib(data: bytes, outcap: int):
    return _inflate(lib().ora_inflate_zlib, data, outcap)


def deflate_cont(data: bytes, level: int, calls=(), strategy: int = 0, dictionary: bytes = b"", params=None) -> bytes:
    """The continuous raw stream (ora_deflate_cont): calls = [(upto, flush), ...] as in refzlib.deflate_calls; the Z_FINISH call is implied.
    params: {call index: (level, strategy)} -- deflateParams() in front of that call (len(calls): in front of the Z_FINISH call)."""
    L = lib()
    if params:
        d = dictionary[-32506:] if dictionary and len(dictionary) >= 3 else b""
        buf = d + data
        n1 = len(calls) + 1
        cuts = (C.c_uint32 * n1)(*([u for u, _ in calls] + [0]))
        kinds = (C.c_int32 * n1)(*([f for _, f in calls] + [0]))
        pl = (C.c_int32 * n1)(*[params.get(k, (-1, -1))[0] for k in range(n1)])
        ps = (C.c_int32 * n1)(*[params.get(k, (-1, -1))[1] for k in range(n1)])
        cap = len(data) + (len(data) >> 3) + 64 * (len(calls) + 2) + 1024
        out = C.create_string_buffer(cap)
        n = L.ora_deflate_cont_p(buf, len(buf), len(d), level, strategy, cuts, kinds, pl, ps, len(calls), out, cap)
        if n == 0:
            raise RuntimeError("oracle deflate_cont_p failed")
        return out.raw[:n]
    d = dictionary[-32506:] if dictionary and len(dictionary) >= 3 else b""
    buf = d + data
    cuts = (C.c_uint32 * max(len(calls), 1))(*[u for u, _ in calls])
    kinds = (C.c_int32 * max(len(calls), 1))(*[f for _, f in calls])
    cap = len(data) + (len(data) >> 3) + 64 * (len(calls) + 2) + 1024
    out = C.create_string_buffer(cap)
    n = L.ora_deflate_cont(buf, len(buf), len(d), level, strategy, cuts, kinds, len(calls), out, cap)
    if n == 0:
        raise RuntimeError("oracle deflate_cont failed")
    return out.raw[:n]


def cont_stream(data: bytes, level: int, calls=(), wbits: int = 15, strategy: int = 0, dictionary: bytes = None) -> bytes:
    """What the reference's deflate() writes for ONE stream driven by `calls` = [(upto, flush), ...] (the Z_FINISH call is implied; () is compress2):
    the header deflate() writes for these windowBits (15 zlib, -15 raw, 31 gzip without a gz_header; deflate.c:578-649), the continuous raw stream
    (ora_deflate_cont) and the trailer (deflate.c:833-855).  The CPU restatement's counterpart of refzlib.deflate_calls."""
    if level == -1:
        level = 6
    raw = deflate_cont(data, level, calls, strategy, dictionary or b"")
    if wbits < 0:
        return raw
    if wbits > 15:
        xfl = 2 if level == 9 else 4 if (strategy >= 2 or level < 2) else 0
        return bytes([31, 139, 8, 0, 0, 0, 0, 0, xfl, 3]) + raw + crc32(data).to_bytes(4, "little") + (len(data) & 0xFFFFFFFF).to_bytes(4, "little")
    hdr = (8 + ((wbits - 8) << 4)) << 8
    lf = 0 if (strategy >= 2 or level < 2) else 1 if level < 6 else 2 if level == 6 else 3
    hdr |= lf << 6
    use_dict = dictionary is not None and len(dictionary) >= 0 and dictionary is not None
    if dictionary is not None:
        hdr |= 0x20
    hdr += 31 - hdr % 31
    head = bytes([hdr >> 8, hdr & 255])
    if dictionary is not None:
        head += adler32(dictionary).to_bytes(4, "big")
    return head + raw + adler32(data).to_bytes(4, "big")
