"""SURVEY.md 8f N4 -- streams that were not produced in chunks (the system zlib's output) decoded in pieces: block starts found by
search, every piece decoded with the window in front of it unknown, the chain of pieces checked, the unknowns filled in
(zgpu_inflate.hip, spec_*).  Bytes must equal the input; verdicts on damaged and cut streams must be the one-workgroup decoder's
(which the round-1 tests pin against the reference and the system zlib); the counter says which way a stream went."""
import zlib

import numpy as np
import pytest

from oracle import cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import zlib_amd
    e = zlib_amd.Engine(0)
    yield e
    e.close()


def corpus(eng, kind, first, nchunks):
    import torch
    src = torch.empty(nchunks * 65536, dtype=torch.uint8, device="cuda")
    eng.corpus_fill_device(kind, 0x5EED5117, first, nchunks, src.data_ptr())
    return src.cpu().numpy().tobytes()


def raw_deflate(data, level, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=-15, mem=8):
    co = zlib.compressobj(level, zlib.DEFLATED, wbits, mem, strategy)
    return co.compress(data) + co.flush()


@pytest.mark.parametrize("level", [1, 6, 9])
@pytest.mark.parametrize("kind", [0, 1])
def test_system_zlib_streams_are_decoded_in_pieces(eng, kind, level):
    data = corpus(eng, kind, 100 * level, 256)  # 16 MiB
    raw = raw_deflate(data, level)
    before = eng.spec_counts()
    out = eng.inflate_stream_host(raw, len(data))
    after = eng.spec_counts()
    assert out == data
    assert eng.last_inflate.adler32 == zlib.adler32(data) and eng.last_inflate.crc32 == zlib.crc32(data)
    assert after[0] == before[0] + 1 and after[1] == before[1], (before, after)


def test_streams_of_other_shapes(eng):
    """Stored blocks (incompressible data), fixed blocks (Z_FIXED), Huffman-only, RLE, small windows, runs of zeros (matches of distance 1
    across piece borders), a window kept across sync flushes: whatever the finder finds or does not find, the bytes are the input's."""
    rnd = np.random.default_rng(7).integers(0, 256, 6 << 20, dtype=np.uint8).tobytes()
    text = corpus(eng, 1, 5, 96)
    mix = corpus(eng, 0, 9, 96)
    streams = [
        ("stored", raw_deflate(rnd, 6), rnd),
        ("level0", raw_deflate(text, 0), text),
        ("fixed", raw_deflate(text, 6, zlib.Z_FIXED), text),
        ("huffman", raw_deflate(mix, 6, zlib.Z_HUFFMAN_ONLY), mix),
        ("rle", raw_deflate(mix, 6, zlib.Z_RLE), mix),
        ("window512", raw_deflate(text, 6, wbits=-9), text),
        ("mem1", raw_deflate(mix, 9, mem=1), mix),
        ("zeros", raw_deflate(bytes(32 << 20), 6), bytes(32 << 20)),
        ("mixed", raw_deflate(text[: 2 << 20] + rnd[: 1 << 20] + bytes(1 << 20) + mix[: 2 << 20], 6), text[: 2 << 20] + rnd[: 1 << 20] + bytes(1 << 20) + mix[: 2 << 20]),
    ]
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    z = b"".join(co.compress(text[i:i + 100000]) + co.flush(zlib.Z_SYNC_FLUSH) for i in range(0, len(text), 100000)) + co.flush()
    streams.append(("syncflush", z, text))
    for name, raw, want in streams:
        out = eng.inflate_stream_host(raw, len(want))
        assert out == want, name
        assert eng.last_inflate.crc32 == zlib.crc32(want), name


def test_deflate_streams_inside_stored_blocks_do_not_derail_the_pieces(eng):
    """An archive of compressed files compresses to stored blocks that are full of block headers -- none of them a block start of THIS stream.
    A piece that runs past such a start shows it up; the list of starts is repaired and the pieces decoded again (no one-workgroup fallback)."""
    text = corpus(eng, 1, 40, 160)
    inner = b"".join(zlib.compress(text[i:i + (1 << 20)], 6) for i in range(0, len(text), 1 << 20))  # ~3 MiB of deflate streams
    for name, want in (("stored", inner * 3), ("mixed", text[: 3 << 20] + inner + text[3 << 20: 6 << 20] + inner)):
        for level in (0, 6):
            raw = raw_deflate(want, level)
            before = eng.spec_counts()
            out = eng.inflate_stream_host(raw, len(want))
            after = eng.spec_counts()
            assert out == want, (name, level)
            assert after[0] == before[0] + 1 and after[1] == before[1], (name, level, before, after)


def test_capacity_and_tails(eng):
    import zlib_amd
    data = corpus(eng, 0, 33, 128)
    raw = raw_deflate(data, 6)
    # a destination that is too small: the size that would have been needed comes back
    with pytest.raises(zlib_amd.EngineError) as ei:
        eng.inflate_stream_host(raw, len(data) - 1)
    assert ei.value.code == -5 and eng.last_inflate.out_bytes == len(data)
    # a far larger destination than the stream fills
    assert eng.inflate_stream_host(raw, 64 * len(data)) == data
    # stream mode: bytes behind the final block stay with the caller
    from zlib_amd import gpu
    out = eng.inflate_stream_host(raw + b"TRAILER" * 1000, len(data), flags=1)
    assert out == data and eng.last_inflate.stream_end == 1 and eng.last_inflate.in_used == len(raw)
    # strict mode: the same bytes are an error of the body
    with pytest.raises(zlib_amd.EngineError):
        eng.inflate_stream_host(raw + b"TRAILER" * 1000, len(data))


def test_damaged_and_cut_streams_keep_their_verdicts(eng):
    import zlib_amd
    data = corpus(eng, 1, 77, 64)
    raw = bytearray(raw_deflate(data, 6))
    for at in (len(raw) // 7, len(raw) // 2, len(raw) - 40000):
        bad = bytearray(raw)
        bad[at] ^= 0x10
        try:
            want = zlib.decompress(bytes(bad), -15)
        except zlib.error:
            want = None
        if want is None:
            with pytest.raises(zlib_amd.EngineError) as ei:
                eng.inflate_stream_host(bytes(bad), len(data) + 65536)
            assert ei.value.code == -3
        else:  # the flip landed where the format does not care (a literal, an extra bit): both decode to the same bytes
            assert eng.inflate_stream_host(bytes(bad), len(want) + 65536) == want
    cut = bytes(raw[: len(raw) * 2 // 3])
    with pytest.raises(zlib_amd.EngineError):
        eng.inflate_stream_host(cut, len(data))
    out = eng.inflate_stream_host(cut, len(data), flags=1)  # stream mode: not an error -- the whole pieces in front of the cut are delivered (round 3)
    r = eng.last_inflate
    assert r.incomplete == 1 and r.stream_end == 0 and 0 < r.in_used < len(cut) and r.in_used_bits < 8
    assert len(out) > len(data) // 2 and out == data[: len(out)]


def test_zlib_api_uncompress_of_a_foreign_stream(eng):
    from tests import zhost as Z
    data = corpus(eng, 0, 500, 512)  # 32 MiB
    z = zlib.compress(data, 6)
    before = eng.spec_counts()
    rc, out = Z.uncompress(z, len(data))
    assert rc == 0 and out == data
    assert eng.spec_counts()[0] == before[0] + 1


def test_preset_dictionary_in_front_of_the_first_piece(eng):
    """inflateSetDictionary (qcsrc/inflate.c:1200-1236) with a stream decoded in pieces: the first piece may reach back into the dictionary, the others
    into what the pieces in front of them produced; a stream that needs a dictionary it does not get is an error of the stream."""
    import zlib_amd
    text = corpus(eng, 1, 900, 200)
    zdict = text[5000:5000 + 32768]
    data = zdict[1000:9000] * 3 + text[1 << 20: 9 << 20]  # starts with bytes that are cheapest as copies from the dictionary
    co = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_DEFAULT_STRATEGY, zdict)
    raw = co.compress(data) + co.flush()
    eng.inflate_set_dictionary(zdict)
    try:
        before = eng.spec_counts()
        out = eng.inflate_stream_host(raw, len(data))
        assert out == data
        assert eng.spec_counts()[0] == before[0] + 1
    finally:
        eng.inflate_set_dictionary(b"")
    with pytest.raises(zlib_amd.EngineError) as ei:
        eng.inflate_stream_host(raw, len(data))
    assert ei.value.code == -3


def test_a_stream_longer_than_the_one_workgroup_decoder_takes(eng):
    """640 MiB of compressed input (the one-workgroup decoder stops at 512 MiB: its bit counts are 32 bits wide): stored blocks of random bytes with
    compressible stretches between them, level 1.  Only the pieces can decode it; bytes and CRC-32 must be the input's."""
    rnd = np.random.default_rng(11).integers(0, 256, 160 << 20, dtype=np.uint8).tobytes()
    text = corpus(eng, 1, 123, 256)
    data = b"".join(rnd[i * (40 << 20):(i + 1) * (40 << 20)] + text for i in range(4)) * 4   # 4 x (160 MiB random + 64 MiB text) = 896 MiB
    co = zlib.compressobj(1, zlib.DEFLATED, -15)
    raw = b"".join(co.compress(data[i:i + (64 << 20)]) for i in range(0, len(data), 64 << 20)) + co.flush()
    assert len(raw) > (1 << 29) + (64 << 20)
    dst = np.zeros(len(data), dtype=np.uint8)
    before = eng.spec_counts()
    out = eng.inflate_stream_host(raw, len(data), out=dst)
    assert eng.spec_counts() == (before[0] + 1, before[1])
    assert eng.last_inflate.crc32 == zlib.crc32(data) and out.tobytes() == data


def test_many_streams_around_the_threshold_of_the_pieces(eng):
    """Forty streams of the system zlib between 64 KiB and 6 MiB of compressed size -- levels, strategies, window sizes, sync and full flushes at random
    places, inputs from text to noise, some with bytes behind the end: whichever way a stream goes (too short for pieces, too few block starts,
    pieces, repair, one workgroup), the bytes are the input's and the end of the stream is where the system zlib says it is."""
    rng = np.random.default_rng(2024)
    text = corpus(eng, 1, 31, 128)
    mix = corpus(eng, 0, 32, 128)
    noise = rng.integers(0, 256, 8 << 20, dtype=np.uint8).tobytes()
    for k in range(40):
        kind = int(rng.integers(0, 4))
        n = int(rng.integers(150_000, 8_000_000))
        src = (text, mix, noise, text)[kind]
        o = int(rng.integers(0, len(src) - n))
        data = src[o:o + n]
        if kind == 3:  # stretches of noise inside text
            cut = n // 3
            data = data[:cut] + noise[o % 1000: o % 1000 + cut] + data[cut:]
        level = int(rng.choice([1, 3, 6, 9]))
        strategy = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_RLE, zlib.Z_FIXED]))
        wbits = -int(rng.choice([15, 15, 15, 12, 9]))
        co = zlib.compressobj(level, zlib.DEFLATED, wbits, 8, strategy)
        parts, pos = [], 0
        while pos < len(data):
            step = int(rng.integers(50_000, 2_000_000))
            parts.append(co.compress(data[pos:pos + step]))
            pos += step
            r = rng.random()
            if pos < len(data) and r < 0.15:
                parts.append(co.flush(zlib.Z_SYNC_FLUSH))
            elif pos < len(data) and r < 0.2:
                parts.append(co.flush(zlib.Z_FULL_FLUSH))
        parts.append(co.flush())
        raw = b"".join(parts)
        tail = b"" if k % 3 else bytes(rng.integers(0, 256, int(rng.integers(1, 5000)), dtype=np.uint8))
        out = eng.inflate_stream_host(raw + tail, len(data) + 100, flags=1)
        assert out == data, (k, level, strategy, wbits, len(raw))
        assert eng.last_inflate.stream_end == 1 and eng.last_inflate.in_used == len(raw), (k, eng.last_inflate.in_used, len(raw))
        assert eng.last_inflate.crc32 == zlib.crc32(data), k
