"""ctypes view of the zlib-compatible host library (zlib_amd/libzamd_z.so) -- same z_stream layout as the reference
(/root/reference/h/zlib.h:82-101).  Test helper."""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "zlib_amd", "libzamd_z.so")
Z_NO_FLUSH, Z_PARTIAL_FLUSH, Z_SYNC_FLUSH, Z_FULL_FLUSH, Z_FINISH = 0, 1, 2, 3, 4
Z_OK, Z_STREAM_END, Z_NEED_DICT, Z_STREAM_ERROR, Z_DATA_ERROR, Z_MEM_ERROR, Z_BUF_ERROR, Z_VERSION_ERROR = 0, 1, 2, -2, -3, -4, -5, -6


class ZStream(C.Structure):
    _fields_ = [("next_in", C.c_void_p), ("avail_in", C.c_uint), ("total_in", C.c_ulong), ("next_out", C.c_void_p),
                ("avail_out", C.c_uint), ("total_out", C.c_ulong), ("msg", C.c_char_p), ("state", C.c_void_p),
                ("zalloc", C.c_void_p), ("zfree", C.c_void_p), ("opaque", C.c_void_p), ("data_type", C.c_int),
                ("adler", C.c_ulong), ("reserved", C.c_ulong)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        try:
            import torch  # noqa: F401  (one HIP runtime per process: see zlib_amd/gpu.py)
        except ImportError:
            pass
        L = C.CDLL(SO)
        P = C.POINTER(ZStream)
        L.deflateInit_.argtypes = [P, C.c_int, C.c_char_p, C.c_int]
        L.deflateInit2_.argtypes = [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.deflate.argtypes = [P, C.c_int]
        L.deflateEnd.argtypes = [P]
        L.deflateReset.argtypes = [P]
        L.inflateInit_.argtypes = [P, C.c_char_p, C.c_int]
        L.inflateInit2_.argtypes = [P, C.c_int, C.c_char_p, C.c_int]
        L.inflate.argtypes = [P, C.c_int]
        L.inflateEnd.argtypes = [P]
        L.compress2.argtypes = [C.c_void_p, C.POINTER(C.c_ulong), C.c_void_p, C.c_ulong, C.c_int]
        L.compress.argtypes = [C.c_void_p, C.POINTER(C.c_ulong), C.c_void_p, C.c_ulong]
        L.uncompress.argtypes = [C.c_void_p, C.POINTER(C.c_ulong), C.c_void_p, C.c_ulong]
        L.compressBound.argtypes = [C.c_ulong]
        L.compressBound.restype = C.c_ulong
        L.adler32.argtypes = [C.c_ulong, C.c_char_p, C.c_uint]
        L.adler32.restype = C.c_ulong
        L.adler32_combine.argtypes = [C.c_ulong, C.c_ulong, C.c_long]
        L.adler32_combine.restype = C.c_ulong
        L.zlibVersion.restype = C.c_char_p
        L.zError.argtypes = [C.c_int]
        L.zError.restype = C.c_char_p
        _lib = L
    return _lib


def compress2(data: bytes, level: int, cap=None):
    L = lib()
    n = C.c_ulong(L.compressBound(len(data)) if cap is None else cap)
    out = C.create_string_buffer(max(n.value, 1))
    src = C.create_string_buffer(data, max(len(data), 1))
    rc = L.compress2(out, C.byref(n), src, len(data), level)
    return rc, out.raw[: n.value] if rc == 0 else b""


def uncompress(z: bytes, cap: int):
    L = lib()
    n = C.c_ulong(cap)
    out = C.create_string_buffer(max(cap, 1))
    src = C.create_string_buffer(z, max(len(z), 1))
    rc = L.uncompress(out, C.byref(n), src, len(z))
    return rc, out.raw[: n.value] if rc == 0 else b""


def deflate_stream(data: bytes, level: int, plan, in_step=None, out_step=None, window_bits=15, strategy=0):
    """Drive deflate() like a streaming caller.  plan: list of (nbytes, flush) pieces; in_step/out_step: feed/drain
    granularity inside a piece (None = everything at once).  Returns (bytes, return codes seen, final z_stream)."""
    L = lib()
    s = ZStream()
    rc = L.deflateInit2_(C.byref(s), level, 8, window_bits, 8, strategy, b"1.2.3", C.sizeof(ZStream))
    assert rc == Z_OK, rc
    src = C.create_string_buffer(data, max(len(data), 1))
    cap = L.compressBound(len(data)) + 64 * len(plan) + 1024
    out = C.create_string_buffer(cap)
    opos, ipos, codes = 0, 0, []
    for nbytes, flush in plan:
        end = ipos + nbytes
        while True:
            last_piece = True
            step = end - ipos if in_step is None else min(in_step, end - ipos)
            if in_step is not None and ipos + step < end:
                last_piece = False
            s.next_in = C.addressof(src) + ipos
            s.avail_in = step
            ipos += step
            fl = flush if last_piece else Z_NO_FLUSH
            while True:
                room = cap - opos if out_step is None else min(out_step, cap - opos)
                s.next_out = C.addressof(out) + opos
                s.avail_out = room
                rc = L.deflate(C.byref(s), fl)
                opos += room - s.avail_out
                codes.append(rc)
                assert rc in (Z_OK, Z_STREAM_END, Z_BUF_ERROR), rc
                if rc == Z_STREAM_END or (rc == Z_BUF_ERROR):
                    break
                if s.avail_in == 0 and s.avail_out != 0:
                    break
            if last_piece:
                break
    total_in, total_out, adler, dtype = s.total_in, s.total_out, s.adler, s.data_type
    end_rc = L.deflateEnd(C.byref(s))
    return out.raw[:opos], codes, dict(total_in=total_in, total_out=total_out, adler=adler, data_type=dtype, end_rc=end_rc)


def inflate_stream(z: bytes, cap: int, in_step=None, out_step=None, flush=Z_NO_FLUSH, window_bits=15):
    L = lib()
    s = ZStream()
    rc = L.inflateInit2_(C.byref(s), window_bits, b"1.2.3", C.sizeof(ZStream))
    assert rc == Z_OK, rc
    src = C.create_string_buffer(z, max(len(z), 1))
    out = C.create_string_buffer(max(cap, 1))
    ipos, opos, rc, calls = 0, 0, Z_OK, 0
    while rc == Z_OK and calls < 10_000_000:
        step = len(z) - ipos if in_step is None else min(in_step, len(z) - ipos)
        room = cap - opos if out_step is None else min(out_step, cap - opos)
        s.next_in = C.addressof(src) + ipos
        s.avail_in = step
        s.next_out = C.addressof(out) + opos
        s.avail_out = room
        rc = L.inflate(C.byref(s), flush)
        ipos += step - s.avail_in
        opos += room - s.avail_out
        calls += 1
        if rc == Z_OK and step == 0 and room - s.avail_out == 0 and opos == cap:
            break
    msg = s.msg.decode() if s.msg else None
    adler = s.adler
    L.inflateEnd(C.byref(s))
    return rc, out.raw[:opos], msg, adler
