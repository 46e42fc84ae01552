"""GPU: the deflate strategies (Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED; qcsrc/deflate.c:1485-1497, 1594-1611, trees.c:986)
through the engine's C ABI, against the oracle (pinned to the reference for all four in tests/test_strategies_cpu.py)."""
import pytest

pytestmark = pytest.mark.gpu

from oracle import cases, corpus_py as CP, oracle_py as O  # noqa: E402


@pytest.fixture(scope="module")
def eng():
    import zlib_amd
    e = zlib_amd.Engine(0)
    yield e
    e.close()


def inputs():
    g = cases.Lcg(31)
    out = {"corpus0x6": CP.chunks(0, 20, 6).tobytes()[:-999], "corpus1x3": CP.chunks(1, 5, 3).tobytes(), "hello": cases.hello_1mib()[:200000],
           "empty": b"", "one": b"x", "zeros": bytes(70000)}
    for kind in cases.KINDS:
        out[kind] = cases.make(kind, 65536 - g.below(400), g.below(1000))
    return out


@pytest.mark.parametrize("strategy", [1, 2, 3, 4])
def test_strategies_match_oracle(eng, strategy):
    from zlib_amd import gpu
    data = inputs()
    for level in (1, 3, 4, 6, 9):
        impls = [gpu.LZ_AUTO] + ([gpu.LZ_SERIAL] if level >= 4 else [])
        for name, d in data.items():
            if level == 9 and name == "ab":
                impls = [gpu.LZ_AUTO]  # (the serial kernel walks 4096-deep chains of this input with one lane)
            want = O.deflate_stream(d, level, strategy=strategy)
            for impl in impls:
                got = eng.deflate_host(d, level, lz_impl=impl, strategy=strategy)
                assert got == want, (strategy, level, name, impl, len(got), len(want))


def test_strategy_errors(eng):
    from zlib_amd import gpu
    with pytest.raises(gpu.EngineError):
        eng.deflate_host(b"abc", 6, strategy=5)
    with pytest.raises(gpu.EngineError):
        eng.deflate_host(b"abc" * 100, 6, lz_impl=gpu.LZ_PARALLEL, strategy=1)


def test_host_api_strategies_and_params():
    import ctypes as C
    import zhost as Z
    data = CP.chunks(0, 33, 3).tobytes()[:-500]
    for strategy in (1, 2, 3, 4):
        for level in (1, 6):
            z, codes, info = Z.deflate_stream(data, level, [(len(data), Z.Z_FINISH)], strategy=strategy)
            # one continuous stream, the reference's bytes -- except Z_RLE at levels 1-3, which is served in independent 64 KiB chunks
            want = O.deflate_stream(data, level, strategy=strategy) if (strategy == 3 and level <= 3) else O.cont_stream(data, level, strategy=strategy)
            assert z == want, (strategy, level)
            assert Z.inflate_stream(z, len(data) + 8)[:2] == (Z.Z_STREAM_END, data)
    # deflateParams (deflate.c:416-451): 100000 bytes at level 1, then level 9 / Z_FILTERED for the rest: the compress function changes, so what has
    # been read is flushed with Z_PARTIAL_FLUSH first; the window stays, with the chains as deflate_fast left them
    L = Z.lib()
    s = Z.ZStream()
    assert L.deflateInit_(C.byref(s), 1, b"1.2.3", C.sizeof(Z.ZStream)) == Z.Z_OK
    src = C.create_string_buffer(data, len(data))
    cap = len(data) + 4096
    out = C.create_string_buffer(cap)
    s.next_in = C.addressof(src); s.avail_in = 100000
    s.next_out = C.addressof(out); s.avail_out = cap
    assert L.deflate(C.byref(s), Z.Z_NO_FLUSH) == Z.Z_OK and s.avail_in == 0
    assert L.deflateParams(C.byref(s), 10, 0) == Z.Z_STREAM_ERROR and L.deflateParams(C.byref(s), 9, 5) == Z.Z_STREAM_ERROR
    assert L.deflateParams(C.byref(s), 9, 1) == Z.Z_OK
    s.next_in = C.addressof(src) + 100000; s.avail_in = len(data) - 100000
    assert L.deflate(C.byref(s), Z.Z_FINISH) == Z.Z_STREAM_END
    z = out.raw[: s.total_out]
    assert L.deflateEnd(C.byref(s)) == Z.Z_OK
    raw = O.deflate_cont(data, 1, [(100000, Z.Z_NO_FLUSH)], params={1: (9, 1)})
    assert z == O.deflate_stream(b"", 1)[:2] + raw + O.adler32(data).to_bytes(4, "big")
    assert Z.inflate_stream(z, len(data) + 8)[:2] == (Z.Z_STREAM_END, data)
