"""deflateInit2's geometry (qcsrc/deflate.c:222-297): every windowBits 9..15 x memLevel 1..9 against the compiled reference's chunk streams
(tests/golden/geometry_kat.json, written by oracle/gen_golden_geometry.py) -- through the engine's C ABI (zgpu_deflate_set_geometry + the segment entry
point: the lane-per-chunk loop with run-time window, hash and block sizes, many window slides per chunk, blocks that may not be stored) and through the
z_stream API (deflateInit2 + deflate / deflateSetDictionary / deflateBound; level 0 on the host)."""
import ctypes as C
import hashlib
import json
import os
import zlib

import pytest

pytestmark = pytest.mark.gpu

from oracle import cases  # noqa: E402
import zhost as Z  # noqa: E402

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "geometry_kat.json")))


def h16(b):
    return hashlib.sha256(b).hexdigest()[:16]


def test_every_geometry_through_the_engine():
    import zlib_amd
    from zlib_amd import gpu
    groups = {}
    for c in KAT["chunk"]:
        if c["level"] > 0:
            groups.setdefault((c["w"], c["m"], c["level"]), []).append(c)
    e = zlib_amd.Engine(0)
    bad = []
    try:
        for (w, m, level), cs in sorted(groups.items()):
            e.set_geometry(w, m)
            bufs = [cases.make(c["kind"], c["n"], c["seed"]) for c in cs]
            for last in (0, 1):
                if not last and (w + m + level) % 3:  # (the flush ending for a third of the groups: it differs from the final one in the last block's header only)
                    continue
                segs = e.deflate_segments_host(bufs, level, flags=gpu.F_FINAL if last else 0)
                for c, z, b in zip(cs, segs, bufs):
                    if len(z) != c["len"][last] or h16(z) != c["sha"][last]:
                        bad.append((w, m, level, c["kind"], c["n"], last, len(z), c["len"][last]))
                    elif last:
                        assert zlib.decompressobj(-w).decompress(z) == b
        e.set_geometry(15, 8)
        with pytest.raises(gpu.EngineError):
            e.set_geometry(8, 8)
        with pytest.raises(gpu.EngineError):
            e.set_geometry(15, 10)
        e.set_geometry(12, 8)
        with pytest.raises(gpu.EngineError):  # the kernels built for the default geometry refuse any other
            e.deflate_host(b"abc" * 100, 6, flags=gpu.F_FINAL, lz_impl=gpu.LZ_WALK)
    finally:
        e.close()
    assert not bad, (len(bad), bad[:12])


def _deflate_all(L, data, level, wbits_arg, mem, dictionary=None, flush_each_chunk=False, last=True):
    s = Z.ZStream()
    assert L.deflateInit2_(C.byref(s), level, 8, wbits_arg, mem, 0, b"1.2.3", C.sizeof(Z.ZStream)) == Z.Z_OK
    if dictionary is not None:
        assert L.deflateSetDictionary(C.byref(s), dictionary, len(dictionary)) == Z.Z_OK
    cap = L.deflateBound(C.byref(s), len(data))
    out = C.create_string_buffer(cap)
    inb = C.create_string_buffer(data, max(len(data), 1))
    s.next_in = C.addressof(inb); s.avail_in = len(data); s.next_out = C.addressof(out); s.avail_out = cap
    rc = L.deflate(C.byref(s), Z.Z_FINISH if last else Z.Z_FULL_FLUSH)
    assert rc == (Z.Z_STREAM_END if last else Z.Z_OK) and s.avail_in == 0
    z = out.raw[: s.total_out]
    L.deflateEnd(C.byref(s))
    return z


def test_streams_through_deflateinit2():
    L = Z.lib()
    L.deflateBound.argtypes = [C.POINTER(Z.ZStream), C.c_ulong]
    L.deflateBound.restype = C.c_ulong
    data = cases.make("mix", 150000, 31)
    for c in KAT["stream"]:
        arg = {"zlib": c["w"], "raw": -c["w"], "gzip": c["w"] + 16}[c["wrap"]]
        z = _deflate_all(L, data, c["level"], arg, c["m"])
        assert (len(z), h16(z), z[:10].hex()) == (c["len"], c["sha"], c["head"]), c
        w = 9 if c["w"] == 8 else c["w"]
        back = zlib.decompressobj({"zlib": w, "raw": -w, "gzip": w + 16}[c["wrap"]]).decompress(z)
        assert back == data


def test_level_0_and_dictionaries_in_small_windows():
    L = Z.lib()
    L.deflateBound.argtypes = [C.POINTER(Z.ZStream), C.c_ulong]
    L.deflateBound.restype = C.c_ulong
    L.deflateSetDictionary.argtypes = [C.POINTER(Z.ZStream), C.c_char_p, C.c_uint]
    for c in KAT["chunk"]:
        if c["level"] == 0:  # stored blocks are cut on the host (deflate_stored's rules under the geometry)
            d = cases.make(c["kind"], c["n"], c["seed"])
            for last in (0, 1):
                z = _deflate_all(L, d, 0, -c["w"], c["m"], last=bool(last))
                assert (len(z), h16(z)) == (c["len"][last], c["sha"][last]), (c, last)
    dic = cases.make("text", 40000, 33)
    for c in KAT["dict"]:
        d = cases.make("text", c["n"], 34)
        for last in (0, 1):
            z = _deflate_all(L, d, c["level"], -c["w"], c["m"], dictionary=dic[: c["dict"]], last=bool(last))
            assert (len(z), h16(z)) == (c["len"][last], c["sha"][last]), (c, last)
