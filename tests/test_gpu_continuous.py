"""ONE continuous deflate stream of any size on the device (zlib_amd/csrc/zgpu_cont.hip): what plain compress2() / un-flushed deflate() of
the reference emits -- the window slides through the whole input (qcsrc/deflate.c:1266-1358), matches cross every 64 KiB boundary (:1027-1168, 1554-1674),
blocks are cut every 16383 tokens from the stream's start (h/deflate.h:313).  Checked against tests/golden/continuous_kat.json and kat.json (the compiled
reference's streams, oracle/gen_golden_continuous.py) and against the CPU restatement oracle/deflate_oracle.c (ora_deflate_cont) on seeded inputs."""
import hashlib
import json
import os
import random

import pytest

pytestmark = pytest.mark.gpu

from oracle import cases, corpus_py as CP, oracle_py as O  # noqa: E402
import zhost as Z  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SEEDS = {0: 0x5EED5117, 1: 0x10C7E47}


def corpus(kind, seed, nbytes):
    return CP.chunks(kind, seed, (nbytes + 65535) // 65536).tobytes()[:nbytes]


@pytest.fixture(scope="module")
def eng():
    import zlib_amd
    e = zlib_amd.Engine(0)
    yield e
    e.close()


def test_compress2_of_config_1_is_the_references_stream():
    """VERDICT round 3, missing item 1: compress2(1 MiB of "hello, hello! ") = the reference's 2071 / 2071 / 5632 bytes at levels 6 / 9 / 1."""
    kat = json.load(open(os.path.join(GOLD, "kat.json")))
    big = cases.hello_1mib()
    for level, want_len in ((6, 2071), (9, 2071), (1, 5632)):
        rc, z = Z.compress2(big, level)
        ref = kat["hello_1mib"][str(level)]
        assert rc == 0 and len(z) == want_len == ref["compress2_len"] and hashlib.sha256(z).hexdigest() == ref["compress2_sha256"], level
        assert Z.uncompress(z, len(big)) == (0, big)
    L = Z.lib()
    for n, b in kat["compressBound"].items():
        assert L.compressBound(int(n)) == b  # compress.c:75-79, to the byte


def test_golden_streams_of_both_corpora_through_the_z_stream_api():
    """65537 bytes, 1 MiB and 16 MiB, un-flushed and with one Z_SYNC_FLUSH in the middle (the window is kept, deflate.c:808-819): length and SHA-256 of the
    compiled reference's streams."""
    kat = json.load(open(os.path.join(GOLD, "continuous_kat.json")))
    data = {}
    bad = []
    for r in kat["rows"]:
        first = r.get("first_chunk", SEEDS[r["corpus"]])  # (the corpus from this chunk on: the bench's 256 MiB rows start at chunk 0)
        key = (r["corpus"], first, r["n"])
        if key not in data:
            data = {key: corpus(r["corpus"], first, r["n"])}
        d = data[key]
        plan = [(len(d), Z.Z_FINISH)] if r["sync_at"] is None else [(r["sync_at"], Z.Z_SYNC_FLUSH), (len(d) - r["sync_at"], Z.Z_FINISH)]
        z, codes, info = Z.deflate_stream(d, r["level"], plan)
        if len(z) != r["len"] or hashlib.sha256(z).hexdigest() != r["sha256"]:
            bad.append((r["corpus"], r["n"], r["level"], r["sync_at"], len(z), r["len"]))
        assert info["adler"] == O.adler32(d)
    assert not bad, bad[:10]


def test_fast_levels_round_zero_in_phases(eng, monkeypatch):
    """Levels 1-3: only every K-th tile of a batch starts from a guess, the K - 1 behind it are parsed phase by phase from the tile in front's results
    (zgpu_engine.hip lz_tiles_fast; K = tiles / 256 by default, 1 for inputs this small).  Runs of 1, 3, 4 and 7 tiles over 3 MiB of both corpora and over data
    whose parse never forgets where it started (a period, zeros): the restatement's bytes whatever K is."""
    from zlib_amd import gpu
    F = gpu.F_FINAL | gpu.F_CONTINUOUS
    rnd = random.Random(11)
    per = bytes(rnd.getrandbits(8) for _ in range(5000))
    srcs = {"sil": corpus(0, 31, 3 << 20), "log": corpus(1, 32, 3 << 20), "per": (per * 700)[: 3 << 20], "zeros+": bytes(1 << 20) + corpus(0, 33, 1 << 20) + bytes(700000)}
    for name, d in srcs.items():
        for level in (1, 2, 3):
            want = O.deflate_cont(d, level)
            for k in ("1", "3", "4", "7"):
                monkeypatch.setenv("ZGPU_FAST_RUN", k)
                got = eng.deflate_host(d, level, flags=F)
                assert got == want, (name, level, k, len(got), len(want))
    monkeypatch.delenv("ZGPU_FAST_RUN")


def test_engine_one_shot_against_the_restatement(eng):
    """zgpu_deflate_host(ZGPU_F_CONTINUOUS): sizes around the tiles' ranges (32512 positions each, the first 65024) and the window's slides, every level,
    strategies, several batches of tiles per feed."""
    from zlib_amd import gpu
    F = gpu.F_FINAL | gpu.F_CONTINUOUS
    rnd = random.Random(3)
    base = bytes(rnd.getrandbits(8) for _ in range(32506))
    words = [bytes(rnd.choice(b"abcdefghij ") for _ in range(rnd.randrange(2, 9))) for _ in range(300)]
    tb = b"".join(rnd.choice(words) for _ in range(9000))[:32506]
    srcs = {"per": base * 5, "pert": tb * 5, "sil": corpus(0, 21, 140000)}  # (a period of MAX_DIST: every first candidate lies exactly 32506 back)
    sizes = set()
    for ps in (65274, 98042, 130810):
        for d in (0, 1, 100, 260, 261, 262, 263):
            sizes.add(ps + d)
    sizes |= {0, 1, 2, 3, 4, 262, 263, 32512, 32513, 65023, 65024, 65025, 65536, 65537, 97536, 97537}
    for name, src in srcs.items():
        for n in sorted(sizes):
            if n > len(src):
                continue
            for level in (1, 3, 4, 6, 9):
                assert eng.deflate_host(src[:n], level, flags=F) == O.deflate_cont(src[:n], level), (name, n, level)
    d = corpus(0, 22, 2 << 20) + bytes(rnd.getrandbits(8) for _ in range(200000)) + corpus(1, 5, 1 << 20)
    for level in range(1, 10):
        assert eng.deflate_host(d, level, flags=F | gpu.F_ZLIB_WRAP) == O.cont_stream(d, level), level
    for strategy, levels in ((1, (1, 6)), (2, (6,)), (3, (6, 9)), (4, (2, 6))):
        for level in levels:
            assert eng.deflate_host(d[: 1 << 20], level, flags=F, strategy=strategy) == O.deflate_cont(d[: 1 << 20], level, strategy=strategy), (strategy, level)
    for bt, pipe in (("1", "0"), ("3", "2"), ("7", "2"), ("64", "0")):  # (ZGPU_CONT_PIPE=2: a batch's blocks are made on a second stream under the next batch's walkers)
        os.environ["ZGPU_CONT_BATCH_TILES"] = bt
        os.environ["ZGPU_CONT_PIPE"] = pipe
        try:
            for level in (1, 6, 9):
                assert eng.deflate_host(d, level, flags=F) == O.deflate_cont(d, level), (bt, pipe, level)
        finally:
            del os.environ["ZGPU_CONT_BATCH_TILES"], os.environ["ZGPU_CONT_PIPE"]
    for name, d2 in (("a", b"a" * (1 << 20)), ("ab", b"ab" * (1 << 19)), ("zeros", bytes(200000))):  # walkers that never meet: the exits' serial path
        for level in (1, 4, 9):
            assert eng.deflate_host(d2, level, flags=F) == O.deflate_cont(d2, level), (name, level)
    gz = eng.deflate_host(d[:300000], 6, flags=F | gpu.F_GZIP_WRAP)
    assert gz == O.cont_stream(d[:300000], 6, wbits=31)


def test_deflate_calls_slices_and_flushes_against_the_restatement():
    """deflate() driven like an application drives it: slices with Z_NO_FLUSH, Z_SYNC_FLUSH / Z_PARTIAL_FLUSH (the window stays), Z_FULL_FLUSH (it is
    forgotten), small output space -- the reference's bytes for the same calls (the restatement is checked against it on the CPU)."""
    rnd = random.Random(17)
    d = corpus(0, 11, 3 << 20)
    plans = [[(len(d) // 2, 2), (len(d) - len(d) // 2, 4)], [(1000000, 0), (500000, 3), (700000, 1), (len(d) - 2200000, 4)], [(100000, 2)] * 20 + [(len(d) - 2000000, 4)]]
    for it in range(3):
        plan = []; pos = 0
        while pos < len(d):
            n = min(len(d) - pos, rnd.choice([1, 5, 100, 4096, 32768 - 262 + rnd.randrange(0, 300), 65536, 200000, 900000]))
            pos += n
            plan.append((n, 4 if pos == len(d) else rnd.choice([0, 0, 0, 1, 2, 3])))
        plans.append(plan)
    for pi, plan in enumerate(plans):
        # (every other plan with feeds of the engine in the middle of Z_NO_FLUSH input: ZGPU_CONT_MORE)
        os.environ["ZAMD_FEED_BYTES"] = "150000" if pi % 2 else str(16 << 20)
        for level, wbits in ((6, 15), (9, -15), (4, 31), (0, 15), (1, 15), (3, -15)):
            for in_step, out_step in ((None, None), (30011, 4099)):
                calls = []; pos = 0
                for n, f in plan:
                    if in_step:
                        calls += [(q, 0) for q in range(pos + in_step, pos + n, in_step)]
                    pos += n
                    if f != 4:
                        calls.append((pos, f))
                z, codes, info = Z.deflate_stream(d, level, plan, in_step=in_step, out_step=out_step, window_bits=wbits)
                assert z == O.cont_stream(d, level, calls, wbits=wbits), (level, wbits, len(plan), in_step)
    del os.environ["ZAMD_FEED_BYTES"]


def test_feed_interface_state_travels_with_the_caller(eng):
    """zgpu_deflate_cont_host: feeds that stop 512 bytes in front of their input's end (ZGPU_CONT_MORE), flush feeds, the finishing feed; the block that is
    filling, the unfinished byte and (levels 1-3) the chains' bits go from feed to feed through the caller."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import cont_feed as CF
    d = corpus(0, 11, 2 << 20)
    rnd = random.Random(5)
    for it in range(4):
        calls = []; pos = 0
        while True:
            pos += rnd.choice([5, 100, 4096, 32768 - 262 + rnd.randrange(0, 300), 65536, 200000])
            if pos >= len(d):
                break
            calls.append((pos, rnd.choice([0, 0, 0, 1, 2, 3])))
        for level in (1, 3, 6, 9):
            assert CF.stream(eng, d, level, calls, more_at=rnd.choice([70000, 200000])) == O.deflate_cont(d, level, calls), (it, level)
