#!/usr/bin/env python3
"""Golden vectors of the API rows added in round 2, produced by the compiled reference:  python oracle/gen_golden_api.py
->  tests/golden/api_kat.json
  tune      deflateTune (qcsrc/deflate.c:453-470): sha256[:16] of the reference's raw chunk stream, fresh stream + tune + Z_FINISH / Z_FULL_FLUSH
  gzhead    deflateSetHeader (deflate.c:393-401, 578-754): the reference's whole gzip member (hex) for one-chunk inputs
  trailing  what the reference's inflate() does with bytes behind the end of a stream (inflate.c:1114): rc, total_in, total_out
  prefix    what it delivers for a prefix of a flushed stream that ends right behind a flush marker
TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cases, refzlib as R  # noqa: E402


class GzHeader(C.Structure):
    _fields_ = [("text", C.c_int), ("time", C.c_ulong), ("xflags", C.c_int), ("os", C.c_int), ("extra", C.c_char_p), ("extra_len", C.c_uint),
                ("extra_max", C.c_uint), ("name", C.c_char_p), ("name_max", C.c_uint), ("comment", C.c_char_p), ("comm_max", C.c_uint),
                ("hcrc", C.c_int), ("done", C.c_int)]


TUNES = [(4, 4, 8, 4), (8, 16, 128, 128), (32, 258, 258, 4096), (1, 1, 3, 1), (3, 7, 20, 3), (258, 258, 258, 65535), (6, 30, 64, 2), (4, 5, 258, 40)]
TUNE_LEVELS = [1, 3, 4, 6, 9]
GZHEADS = [dict(text=1, time=0x12345678, os=3, extra=None, name=b"file.txt", comment=None, hcrc=0),
           dict(text=0, time=1, os=255, extra=b"\x01\x02EXTRA-FIELD", name=b"n", comment=b"a comment", hcrc=1),
           dict(text=0, time=0, os=0, extra=b"", name=None, comment=b"", hcrc=1),
           dict(text=1, time=0xFFFFFFFF, os=11, extra=None, name=None, comment=None, hcrc=0)]


def tune_cases():
    for kind, n, seed in (("text", 40000, 1), ("mix", 65536, 2), ("ab", 9000, 3), ("period", 30000, 4), ("runs", 65000, 5), ("rand", 5000, 6)):
        yield kind, n, seed


def ref_tuned_chunk(L, data, level, tune, last):
    s = R.ZStream()
    assert L.deflateInit2_(C.byref(s), level, 8, -15, 8, 0, b"1.2.3", C.sizeof(R.ZStream)) == 0
    assert L.deflateTune(C.byref(s), *tune) == 0
    cap = len(data) + (len(data) >> 8) + 256
    out = C.create_string_buffer(cap); inb = C.create_string_buffer(data, max(len(data), 1))
    s.next_in = C.addressof(inb); s.avail_in = len(data); s.next_out = C.addressof(out); s.avail_out = cap
    rc = L.deflate(C.byref(s), R.Z_FINISH if last else R.Z_FULL_FLUSH)
    assert rc == (1 if last else 0)
    z = out.raw[: s.total_out]
    L.deflateEnd(C.byref(s))
    return z


def ref_gzip_member(L, data, level, h):
    s = R.ZStream()
    assert L.deflateInit2_(C.byref(s), level, 8, 31, 8, 0, b"1.2.3", C.sizeof(R.ZStream)) == 0
    g = GzHeader(text=h["text"], time=h["time"], os=h["os"], extra=h["extra"], extra_len=len(h["extra"] or b""), name=h["name"], comment=h["comment"], hcrc=h["hcrc"])
    assert L.deflateSetHeader(C.byref(s), C.byref(g)) == 0
    cap = len(data) + 1024
    out = C.create_string_buffer(cap); inb = C.create_string_buffer(data, max(len(data), 1))
    s.next_in = C.addressof(inb); s.avail_in = len(data); s.next_out = C.addressof(out); s.avail_out = cap
    assert L.deflate(C.byref(s), R.Z_FINISH) == 1
    z = out.raw[: s.total_out]
    L.deflateEnd(C.byref(s))
    return z


def ref_inflate_verdict(L, z, wbits, cap):
    s = R.ZStream()
    assert L.inflateInit2_(C.byref(s), wbits, b"1.2.3", C.sizeof(R.ZStream)) == 0
    out = C.create_string_buffer(cap); inb = C.create_string_buffer(z, max(len(z), 1))
    s.next_in = C.addressof(inb); s.avail_in = len(z); s.next_out = C.addressof(out); s.avail_out = cap
    rc = L.inflate(C.byref(s), R.Z_NO_FLUSH)
    res = [rc, int(s.total_in), int(s.total_out), int(s.avail_in), hashlib.sha256(out.raw[: s.total_out]).hexdigest()[:16]]
    L.inflateEnd(C.byref(s))
    return res


def main():
    L = R.lib()
    L.deflateTune.argtypes = [C.POINTER(R.ZStream), C.c_int, C.c_int, C.c_int, C.c_int]
    L.deflateSetHeader.argtypes = [C.POINTER(R.ZStream), C.POINTER(GzHeader)]
    out = {"tune": {}, "gzhead": [], "trailing": [], "prefix": []}
    for kind, n, seed in tune_cases():
        d = cases.make(kind, n, seed)
        for level in TUNE_LEVELS:
            for t in TUNES:
                out["tune"]["%s/%d/%d/L%d/%s" % (kind, n, seed, level, "-".join(map(str, t)))] = [
                    hashlib.sha256(ref_tuned_chunk(L, d, level, t, last)).hexdigest()[:16] for last in (False, True)]
    for i, h in enumerate(GZHEADS):
        for level, (kind, n, seed) in ((6, ("text", 3000, 7)), (1, ("mix", 20000, 8)), (9, ("ab", 500, 9)), (0, ("text", 100, 10))):
            z = ref_gzip_member(L, cases.make(kind, n, seed), level, h)
            out["gzhead"].append(dict(head=i, level=level, kind=kind, n=n, seed=seed, member=z.hex()))
    # bytes behind the end of a stream
    d = cases.make("text", 100000, 11)
    zl = R.deflate_mode_b(d, 6)
    gz = R.deflate_wbits(d[:30000], 6, 31)
    raw = R.deflate_chunk_raw(d[:50000], 6, True)
    for name, z, wbits, cap in (("zlib+garbage", zl + b"trailing garbage \x00\x00\xff\xff more", 15, 200000), ("gzip+gzip", gz + gz, 31, 200000),
                                ("gzip+gzip auto", gz + gz, 47, 200000), ("raw+garbage", raw + bytes(range(256)), -15, 200000), ("zlib exact", zl, 15, 200000)):
        out["trailing"].append(dict(name=name, wbits=wbits, stream=z.hex() if len(z) < 70000 else None, gen=name, verdict=ref_inflate_verdict(L, z, wbits, cap)))
    # a prefix of a mode-B stream that ends right behind the k-th flush marker: the reference has delivered k chunks
    parts = [R.deflate_chunk_raw(d[i * 30000:(i + 1) * 30000], 6, i == 3) for i in range(4)]
    zb = R.zlib_header(6) + b"".join(parts)
    for k in (1, 2, 3):
        cut = 2 + sum(len(p) for p in parts[:k])
        out["prefix"].append(dict(chunks=k, cut=cut, verdict=ref_inflate_verdict(L, zb[:cut], 15, 200000)))
    with open(os.path.join(ROOT, "tests", "golden", "api_kat.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote api_kat.json: %d tune cases, %d gzip members, %d trailing, %d prefix" % (len(out["tune"]), len(out["gzhead"]), len(out["trailing"]), len(out["prefix"])))


if __name__ == "__main__":
    main()
