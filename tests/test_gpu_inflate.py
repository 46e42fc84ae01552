"""GPU parity tests for the inflate path (through the C ABI): bytes identical to the input / to the CPU oracle,
error class and message identical to the reference's verdicts stored in tests/golden/inflate_cases.json."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import cases, corpus_py as CP, oracle_py as O  # noqa: E402


@pytest.fixture(scope="module")
def eng():
    import zlib_amd
    e = zlib_amd.Engine(0)
    yield e
    e.close()


def oracle_stream(data, level, chunk=65536):
    """raw body (no zlib header) of the mode-B stream + segment offsets, from the CPU oracle"""
    n = len(data)
    nchunks = max(1, (n + chunk - 1) // chunk)
    segs = [O.deflate_chunk(data[k * chunk:(k + 1) * chunk], level, k == nchunks - 1) for k in range(nchunks)]
    offs = np.zeros(nchunks + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(s) for s in segs])
    return b"".join(segs), offs


@pytest.mark.parametrize("level", [0, 1, 6, 9])
def test_inflate_oracle_streams(eng, level):
    """Streams produced by the CPU oracle (= the reference's bytes): stored, static and dynamic blocks, multi-block
    chunks (high-entropy data), ragged tail."""
    for kind, n in (("mix", 300000), ("rand", 70000), ("text", 65536 * 3), ("zeros", 100000), ("ab", 40000), ("text", 5), ("rand", 0)):
        data = cases.make(kind, n, 21)
        body, offs = oracle_stream(data, level)
        out = eng.inflate_host(body, offs, out_len=max(len(data), 1))
        assert out == data, (kind, n, level)
        assert eng.last_inflate.adler32 == O.adler32(data)


def far_match_data(seed):
    """64 KiB chunks made of 33 KiB of noise followed by copies of pieces of it at distances around 8 KiB, 16 KiB and MAX_DIST, of every kind of
    length (up to 32 bytes a lane asks for ahead; longer ones are read when their turn comes), three bytes of noise between them"""
    g = cases.Lcg(seed)
    out = bytearray()
    dists = [8190, 8191, 8192, 8193, 8194, 9000, 12000, 12288, 16382, 16383, 16384, 16385, 16386, 20000, 24576, 32000, 32505, 32506, 5000, 300, 1]
    lens = [3, 4, 5, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 257, 258]
    for c in range(3):
        chunk = bytearray(g.below(256) for _ in range(33 * 1024))
        while len(chunk) < 65536 - 300:
            d, n = dists[g.below(len(dists))], lens[g.below(len(lens))]
            for i in range(n):
                chunk.append(chunk[len(chunk) - d])
            chunk += bytes(g.below(256) for _ in range(3))
        chunk += bytes(g.below(256) for _ in range(65536 - len(chunk)))
        out += chunk
    return bytes(out[:-777])  # (a ragged last chunk)


def test_small_ring_far_matches(eng, monkeypatch):
    """The segment's ring is shorter than the farthest distance (8 KiB by default): matches that reach farther back read the destination.  The same
    streams through rings of 8, 16 and 32 KiB -- identical bytes; and a destination that ends early is not read past its end."""
    import zlib_amd
    data = far_match_data(77)
    for level in (9, 6, 1):
        body, offs = oracle_stream(data, level)
        far = 0
        for kb in ("8", "16", "32"):
            monkeypatch.setenv("ZGPU_INF_RING_KB", kb)
            out = eng.inflate_host(body, offs, out_len=len(data))
            assert out == data, (level, kb)
            assert eng.last_inflate.adler32 == O.adler32(data)
        monkeypatch.setenv("ZGPU_INF_RING_KB", "8")
        with pytest.raises(zlib_amd.EngineError) as ei:
            eng.inflate_host(body, offs, out_len=len(data) - 40000)
        assert ei.value.code == -5
    monkeypatch.delenv("ZGPU_INF_RING_KB")


def test_whole_stream_segment(eng):
    """chunk_size ZGPU_WHOLE_STREAM: one raw-deflate stream of any size as a single segment (system zlib as the producer);
    a destination that is too small gets ZGPU_BUF_ERROR and the size that was needed."""
    import zlib
    import zlib_amd
    from zlib_amd import gpu
    data = cases.make("mix", 400000, 3) + bytes(200000) + cases.make("rand", 70000, 4)
    for level in (0, 1, 6, 9):
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        raw = co.compress(data) + co.flush()
        offs = np.array([0, len(raw)], dtype=np.uint64)
        out = eng.inflate_host(raw, offs, chunk_size=gpu.WHOLE_STREAM, out_len=len(data))
        assert out == data, level
        assert eng.last_inflate.adler32 == O.adler32(data)
    with pytest.raises(zlib_amd.EngineError) as ei:
        eng.inflate_host(raw, offs, chunk_size=gpu.WHOLE_STREAM, out_len=len(data) - 5)
    assert ei.value.code == -5 and eng.last_inflate.out_bytes == len(data)


def test_whole_stream_many_alignments(eng):
    """Many streams of assorted sizes, contents and levels through the end-to-end decoder: block boundaries, input-ring refills
    and token hand-overs fall on ever different alignments (a refill that ran two dwords too far showed up once in thousands
    of blocks)."""
    import zlib
    from zlib_amd import gpu
    rng = np.random.default_rng(2024)
    kinds = ("mix", "text", "rand", "ab")
    for i in range(36):
        n = int(rng.integers(150000, 900000))
        data = cases.make(kinds[i % 4], n, 100 + i) if i % 4 != 2 else cases.make("text", n // 2, 100 + i) + cases.make("rand", n // 2, i)
        co = zlib.compressobj(int(rng.integers(1, 10)), zlib.DEFLATED, -15)
        raw = co.compress(data) + co.flush()
        out = eng.inflate_host(raw, np.array([0, len(raw)], dtype=np.uint64), chunk_size=gpu.WHOLE_STREAM, out_len=len(data))
        assert out == data, (i, n)


def test_whole_stream_damaged(eng):
    """Damaged streams of another producer through the end-to-end decoder: it must come to a verdict (no hang, no crash) and the
    verdict must be the system zlib's -- an error, or the very same bytes when the damage happens to leave a valid stream (a raw
    stream has no checksum; 48 cases were run once, 18 are kept: a damaged stream may decode to many megabytes before it ends)."""
    import zlib
    import zlib_amd
    from zlib_amd import gpu
    rng = np.random.default_rng(99)
    data = cases.make("mix", 220000, 12)
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = co.compress(data) + co.flush()
    seen_ok = seen_err = 0
    for i in range(18):
        bad = bytearray(raw)
        if i % 3 == 2:
            bad = bad[: int(rng.integers(1, len(bad)))]                      # cut short
        else:
            for _ in range(1 + i % 3):
                bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        bad = bytes(bad)
        d = zlib.decompressobj(-15)
        try:
            want = d.decompress(bad, 4 * len(data))
            complete = d.eof and not d.unconsumed_tail
        except zlib.error:
            want, complete = None, False
        offs = np.array([0, len(bad)], dtype=np.uint64)
        try:
            got = eng.inflate_host(bad, offs, chunk_size=gpu.WHOLE_STREAM, out_len=4 * len(data))
        except zlib_amd.EngineError as ex:
            got = None
            assert ex.code in (-3, -5), ex
        if want is not None and complete and not d.unused_data:
            assert got == want, i
            seen_ok += 1
        else:
            assert got is None, i                                            # error, truncated, or data behind the final block
            seen_err += 1
    assert seen_err >= 1 and seen_ok + seen_err == 18


def test_far_matches_between_literals(eng):
    """Matches at distances close to the full 32 KiB window with fresh literals right behind them: the output ring in LDS is
    exactly one window long, so a literal stored ahead of its turn would land on bytes such a match still has to read."""
    rng = np.random.default_rng(77)
    base = rng.integers(0, 256, 40000, dtype=np.uint8).tobytes()
    parts, total = [base], len(base)
    while total < 3 * 65536 + 12345:
        back = int(rng.integers(32768 - 700, 32769))          # distance of the next copy
        ln = int(rng.integers(3, 40))
        flat = b"".join(parts)
        parts = [flat, flat[len(flat) - back: len(flat) - back + ln], rng.integers(0, 256, int(rng.integers(1, 30)), dtype=np.uint8).tobytes()]
        total = sum(len(x) for x in parts)
    data = b"".join(parts)
    for level in (1, 6, 9):
        body, offs = oracle_stream(data, level)
        out = eng.inflate_host(body, offs, out_len=len(data))
        assert out == data, level


@pytest.mark.parametrize("level", [1, 6, 9])
def test_roundtrip_engine_to_engine(eng, level):
    data = CP.chunks(CP.KIND_SILESIA, 200, 48).tobytes()[: 48 * 65536 - 777]
    z, offs = eng.deflate_host(data, level, want_offsets=True)
    out = eng.inflate_host(z, offs, out_len=len(data))
    assert out == data
    assert eng.last_inflate.adler32 == int.from_bytes(z[-4:], "big")
    # and the CPU oracle accepts the engine's stream
    rc, ref_out, used, msg = O.inflate_zlib(z, len(data))
    assert rc == 1 and ref_out == data and used == len(z)


def test_smaller_chunk_size_roundtrip(eng):
    data = cases.make("mix", 200000, 5)
    for cs in (4096, 32768):
        z, offs = eng.deflate_host(data, 6, chunk_size=cs, want_offsets=True)
        assert eng.inflate_host(z, offs, chunk_size=cs, out_len=len(data)) == data


def test_corrupt_streams_match_reference_verdict(eng, golden):
    import zlib_amd
    g = golden("inflate_cases.json")
    seen = set()
    for stream_hex, cap, rc, msg, sha_out, len_out in g["rows"]:
        raw = bytes.fromhex(stream_hex)
        offs = np.array([0, len(raw)], dtype=np.uint64)
        if rc == 1:
            # the reference stops at the final block; when the corruption left bytes after it the segment API reports that
            # inconsistency (a segment must be exactly one run of blocks) -- the oracle tells which case this is
            o_rc, o_out, o_used, _ = O.inflate_raw(raw, cap)
            assert o_rc == 1
            if o_used < len(raw):
                with pytest.raises(zlib_amd.EngineError) as ei:
                    eng.inflate_host(raw, offs, chunk_size=min(cap, 65536), out_len=cap)
                assert "after its last block" in str(ei.value)
            else:
                out = eng.inflate_host(raw, offs, chunk_size=min(cap, 65536), out_len=cap)
                assert [len(out), hashlib.sha256(out).hexdigest()[:16]] == [len_out, sha_out]
        else:
            with pytest.raises(zlib_amd.EngineError) as ei:
                eng.inflate_host(raw, offs, chunk_size=min(cap, 65536), out_len=cap)
            if rc == -3:
                assert ei.value.code == -3 and msg in str(ei.value), (msg, str(ei.value))
                seen.add(msg)
            else:
                assert ei.value.code in (-3, -5)
    assert len(seen) >= 6


def test_segment_rules(eng):
    import zlib_amd
    data = cases.make("text", 65536 * 2, 3)
    body, offs = oracle_stream(data, 6)
    # wrong split point
    bad = offs.copy(); bad[1] += 1
    with pytest.raises(zlib_amd.EngineError):
        eng.inflate_host(body, bad, out_len=len(data))
    # last segment without a final block
    segs = [O.deflate_chunk(data[:65536], 6, False), O.deflate_chunk(data[65536:], 6, False)]
    o2 = np.array([0, len(segs[0]), len(segs[0]) + len(segs[1])], dtype=np.uint64)
    with pytest.raises(zlib_amd.EngineError):
        eng.inflate_host(b"".join(segs), o2, out_len=len(data))


def test_checks_of_the_output_can_be_chosen(eng):
    """zgpu_inflate_set_checks: Adler-32 only (a zlib stream), CRC-32 only (a gzip member), both (the default), none (raw deflate); a check that is
    not computed reads 1 / 0, the bytes are the same; a mask beyond the two bits is refused; the setting stays until replaced."""
    import zlib as syszlib
    from zlib_amd import gpu
    data = cases.make("text", 300000, 77) + cases.make("rand", 70000, 78)
    z, offs = eng.deflate_host(data, 6, want_offsets=True)
    ad, cr = syszlib.adler32(data), syszlib.crc32(data)
    try:
        for mask, want in ((gpu.CHECK_ADLER32, (ad, 0)), (gpu.CHECK_CRC32, (1, cr)), (0, (1, 0)), (gpu.CHECK_ADLER32 | gpu.CHECK_CRC32, (ad, cr))):
            eng.inflate_set_checks(mask)
            for _ in range(2):
                assert eng.inflate_host(z, offs, out_len=len(data)) == data
                assert (eng.last_inflate.adler32, eng.last_inflate.crc32) == want, mask
            raw = syszlib.compressobj(6, syszlib.DEFLATED, -15)
            body = raw.compress(data) + raw.flush()
            assert eng.inflate_stream_host(body, len(data)) == data  # (a stream of another producer: the same pass behind another decoder)
            assert (eng.last_inflate.adler32, eng.last_inflate.crc32) == want, mask
        with pytest.raises(Exception):
            eng.inflate_set_checks(4)
    finally:
        eng.inflate_set_checks(gpu.CHECK_ADLER32 | gpu.CHECK_CRC32)
