"""GPU: a chunk behind a preset dictionary (deflateSetDictionary, qcsrc/deflate.c:315-354) through the engine's C ABI, against the
oracle (pinned to the reference in tests/test_dictionary_cpu.py) and the reference's golden hashes."""
import hashlib

import pytest

pytestmark = pytest.mark.gpu

from oracle import cases, oracle_py as O  # noqa: E402
from test_dictionary_cpu import dict_cases  # noqa: E402


@pytest.fixture(scope="module")
def eng():
    import zlib_amd
    e = zlib_amd.Engine(0)
    yield e
    e.close()


def test_dictionary_chunk_matches_reference_vectors(eng, golden):
    kat = golden("dict_kat.json")
    n = 0
    for key, lvl, strat, d, x in dict_cases():
        if lvl == 0 or len(d) < 3:
            continue  # level 0 is host-side framing; a dictionary shorter than MIN_MATCH is ignored (deflate.c:335)
        if lvl == 9 and len(x) > 30000 and n % 3:
            n += 1
            continue  # (one lane walks 4096-deep chains here: a third of the big level-9 cases)
        n += 1
        got = [hashlib.sha256(eng.deflate_dict_chunk_host(d, x, lvl, last, strategy=strat)).hexdigest()[:16] for last in (False, True)]
        assert got == kat[key], key
    data = cases.make("text", 5000, 3)
    z = eng.deflate_dict_chunk_host(data[:2000], data, 6, True)
    assert z == O.deflate_chunk_dict(data[:2000], data, 6, True) and len(z) < len(O.deflate_chunk(data, 6, True))
    assert eng.last.adler32 == O.adler32(data)


def test_dictionary_chunk_errors(eng):
    from zlib_amd import gpu
    with pytest.raises(gpu.EngineError):
        eng.deflate_dict_chunk_host(b"ab", b"data", 6, True)            # fewer than MIN_MATCH dictionary bytes
    with pytest.raises(gpu.EngineError):
        eng.deflate_dict_chunk_host(bytes(30000), bytes(40000), 6, True)  # window content over 64 KiB


def test_host_api_dictionary(golden):
    """deflateSetDictionary / inflateSetDictionary through libzamd_z.so: our stream = header with DICTID, the first chunk behind the
    dictionary, ordinary chunks, Adler; the reference's own dictionary streams (tests/golden/dict_streams.json) inflate."""
    import ctypes as C
    import zhost as Z
    from oracle import corpus_py as CP
    L = Z.lib()

    def deflate_with_dict(dictionary, data, level, pieces):
        s = Z.ZStream()
        assert L.deflateInit_(C.byref(s), level, b"1.2.3", C.sizeof(Z.ZStream)) == Z.Z_OK
        assert L.deflateSetDictionary(C.byref(s), dictionary, len(dictionary)) == Z.Z_OK
        dictid = s.adler
        src = C.create_string_buffer(data, max(len(data), 1))
        cap = len(data) + 8192
        out = C.create_string_buffer(cap)
        s.next_out = C.addressof(out); s.avail_out = cap
        pos = 0
        for i, n in enumerate(pieces):
            s.next_in = C.addressof(src) + pos; s.avail_in = n
            pos += n
            rc = L.deflate(C.byref(s), Z.Z_FINISH if i == len(pieces) - 1 else Z.Z_NO_FLUSH)
            assert rc == (Z.Z_STREAM_END if i == len(pieces) - 1 else Z.Z_OK), rc
        z = out.raw[: s.total_out]
        assert L.deflateSetDictionary(C.byref(s), dictionary, len(dictionary)) == Z.Z_STREAM_ERROR  # deflate.c:326-328
        assert L.deflateEnd(C.byref(s)) == Z.Z_OK
        return z, dictid

    def inflate_with_dict(z, dictionary, cap, wrong=None):
        s = Z.ZStream()
        assert L.inflateInit_(C.byref(s), b"1.2.3", C.sizeof(Z.ZStream)) == Z.Z_OK
        src = C.create_string_buffer(z, len(z))
        out = C.create_string_buffer(cap)
        s.next_in = C.addressof(src); s.avail_in = len(z)
        s.next_out = C.addressof(out); s.avail_out = cap
        assert L.inflateSetDictionary(C.byref(s), dictionary, len(dictionary)) == Z.Z_STREAM_ERROR  # not asked for yet (inflate.c:1212)
        assert L.inflate(C.byref(s), Z.Z_NO_FLUSH) == Z.Z_NEED_DICT
        assert s.adler == O.adler32(dictionary)
        if wrong is not None:
            assert L.inflateSetDictionary(C.byref(s), wrong, len(wrong)) == Z.Z_DATA_ERROR
        assert L.inflateSetDictionary(C.byref(s), dictionary, len(dictionary)) == Z.Z_OK
        rc = L.inflate(C.byref(s), Z.Z_FINISH)
        got = out.raw[: s.total_out]
        assert L.inflateEnd(C.byref(s)) == Z.Z_OK
        return rc, got

    text = cases.make("text", 5000, 4)
    data = CP.chunks(0, 50, 3).tobytes()[:-321]
    for dictionary, level in ((text[:2000], 6), (CP.chunks(0, 49, 1).tobytes(), 9), (b"hello\0", 1), (text[:300], 0)):
        for pieces in ([len(data)], [1000, 70000, len(data) - 71000], [30000, 30000, 30000, len(data) - 90000]):
            # what the reference writes for these calls: ONE stream behind the dictionary (level 0 cuts its blocks by what each call brings)
            calls, pos = [], 0
            for n in pieces[:-1]:
                pos += n
                calls.append((pos, Z.Z_NO_FLUSH))
            want = O.cont_stream(data, level, calls, dictionary=dictionary)
            z, dictid = deflate_with_dict(dictionary, data, level, pieces)
            assert dictid == O.adler32(dictionary)
            assert z == want, (level, len(dictionary), pieces)
        rc, got = inflate_with_dict(want, dictionary, len(data) + 16, wrong=b"not the dictionary")
        assert (rc, got) == (Z.Z_STREAM_END, data)
    kat = golden("dict_streams.json")
    h = kat["hello"]
    assert inflate_with_dict(bytes.fromhex(h["stream"]), bytes.fromhex(h["dict"]), 100) == (Z.Z_STREAM_END, bytes.fromhex(h["data"]))
    z, _ = deflate_with_dict(bytes.fromhex(h["dict"]), bytes.fromhex(h["data"]), h["level"], [len(h["data"]) // 2])
    assert z.hex() == h["stream"]  # a single small chunk: byte-identical to the reference's stream (example.c's test_dict_deflate)
    t = kat["text"]
    full = cases.make(*t["dict_case"][:1], 5000, t["dict_case"][2])
    assert inflate_with_dict(bytes.fromhex(t["stream"]), full[: t["dict_case"][1]], 4000) == (Z.Z_STREAM_END, full[t["data_slice"][0]: t["data_slice"][1]])
