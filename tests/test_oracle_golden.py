"""The CPU restatement (oracle/) against the committed golden vectors, which were produced by the real
reference (oracle/gen_golden.py -> tests/golden/).  Runs anywhere, no GPU, no /root/reference."""
import hashlib
import os

import pytest

from oracle import cases, corpus_py as CP, oracle_py as O


def h16(b):
    return hashlib.sha256(b).hexdigest()[:16]


def check(expected, got: bytes, what):
    if isinstance(expected, str):
        assert got.hex() == expected, what
    else:
        assert [len(got), h16(got)] == expected, what


def test_known_answers(golden):
    kat = golden("kat.json")
    assert kat["reference"].endswith("1.2.3")
    # "hello, hello!" fits one chunk, so the mode-B stream equals plain compress2 (levels 1, 6, 9)
    for lvl in (1, 6, 9):
        assert O.deflate_stream(cases.HELLO, lvl).hex() == kat["hello"][str(lvl)]
    assert O.deflate_stream(cases.HELLO, 6).hex() == kat["hello"]["-1"]
    assert O.deflate_stream(cases.HELLO, 0).hex() == kat["hello"]["0"]
    big = cases.hello_1mib()
    assert hashlib.sha256(big).hexdigest() == kat["hello_1mib"]["sha256_input"]
    assert "%08x" % O.adler32(big) == kat["hello_1mib"]["adler32"] == "c08f758f"
    for lvl in (1, 6, 9):
        z = O.deflate_stream(big, lvl)
        e = kat["hello_1mib"][str(lvl)]
        assert (len(z), hashlib.sha256(z).hexdigest()) == (e["mode_b_len"], e["mode_b_sha256"])
        rc, out, used, msg = O.inflate_zlib(z, len(big))
        assert rc == 1 and out == big and used == len(z)
    for a, b, n, want in kat["adler32_combine"]:
        assert O.adler32_combine(a, b, n) == want


def test_small_chunks_all_levels(golden):
    exp = golden("chunk_small.json")
    n = 0
    for name, data in cases.small_cases():
        e = exp[name]
        for lvl in range(0, 10):
            for last in (0, 1):
                check(e["L%d-last%d" % (lvl, last)], O.deflate_chunk(data, lvl, bool(last)), (name, lvl, last))
                if lvl > 0:
                    check(e["L%d-last%d-p0" % (lvl, last)], O.deflate_chunk(data, lvl, bool(last), True), (name, lvl, last, "p0"))
                n += 1
    assert n == len(exp) * 20


@pytest.mark.parametrize("kind", cases.KINDS)
def test_chunk_size_edges(golden, kind):
    exp = golden("chunk_big.json")
    for name, data in cases.big_cases():
        if not name.startswith(kind + "-"):
            continue
        e = exp[name]
        for lvl in range(0, 10):
            for last in (0, 1):
                check(e["L%d-last%d" % (lvl, last)], O.deflate_chunk(data, lvl, bool(last)), (name, lvl, last))
            if lvl in (1, 6, 9):
                check(e["L%d-last0-p0" % lvl], O.deflate_chunk(data, lvl, False, True), (name, lvl, "p0"))


@pytest.mark.parametrize("fname,stride", [("corpus_silesia.json", 16), ("corpus_logtext.json", 8)])
def test_corpus_sample(golden, fname, stride):
    g = golden(fname)
    assert g["seed"] == CP.default_seed(g["kind"])
    rows = g["rows"][::stride]
    for row in rows:
        data = CP.chunk(g["kind"], row[0])
        assert h16(data) == row[1], "corpus generator drifted from the fixtures"
        for j, lvl in enumerate((1, 6, 9)):
            o = O.deflate_chunk(data, lvl, False)
            assert [len(o), h16(o)] == row[2 + 2 * j: 4 + 2 * j], (row[0], lvl)
    for row in g["last_rows"]:  # the chunks a run ends on, with BFINAL (oracle/gen_golden_last.py)
        data = CP.chunk(g["kind"], row[0])
        assert h16(data) == row[1]
        for j, lvl in enumerate((1, 6, 9)):
            o = O.deflate_chunk(data, lvl, True)
            assert [len(o), h16(o)] == row[2 + 2 * j: 4 + 2 * j], (row[0], lvl, "last")


def test_inflate_cases(golden):
    g = golden("inflate_cases.json")
    seen = set()
    for stream_hex, cap, rc, msg, sha_out, len_out in g["rows"]:
        got_rc, out, used, got_msg = O.inflate_raw(bytes.fromhex(stream_hex), cap)
        assert (got_rc, got_msg) == (rc, msg)
        if rc == 1:
            assert [len(out), h16(out)] == [len_out, sha_out]
        seen.add(msg)
    assert {"invalid block type", "invalid stored block lengths", "invalid distance too far back"} <= seen


def test_roundtrip_every_level():
    data = cases.make("mix", 200000, 9)
    for lvl in range(0, 10):
        z = O.deflate_stream(data, lvl)
        rc, out, used, msg = O.inflate_zlib(z, len(data))
        assert rc == 1 and out == data and used == len(z), lvl


def test_oracle_on_chunks_whose_token_count_fills_the_buffer_exactly():
    """tests/golden/fullblock_kat.json (the compiled reference, oracle/gen_golden_fullblock.py): deflate_slow's trailing literal does not cut a block."""
    import hashlib
    import json
    from oracle import gen_golden_fullblock as G
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fullblock_kat.json")))
    ins = dict(G.inputs())
    for c in kat:
        for last in (0, 1):
            z = O.deflate_chunk(ins[c["name"]], c["level"], bool(last))
            assert (len(z), hashlib.sha256(z).hexdigest()[:16]) == (c["len"][last], c["sha"][last]), (c["name"], c["level"], last)
