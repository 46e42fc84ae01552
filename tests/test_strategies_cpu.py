"""CPU: the oracle's restatement of the deflate strategies against the compiled reference (all four, levels 1-9, both kinds of
chunk boundary, position-0 matchable or not) and against golden hashes made by it (tests/golden/strategy_kat.json)."""
import hashlib

import pytest

from oracle import cases, oracle_py as O, refzlib as R


def h16(b):
    return hashlib.sha256(b).hexdigest()[:16]


def kat_cases():
    g = cases.Lcg(2024)
    for strategy in (1, 2, 3, 4):
        for level in (1, 2, 3, 4, 6, 9):
            for kind in cases.KINDS:
                for n in (0, 1, 7, 300, 5000, 65536, 65536 - g.below(300)):
                    yield strategy, level, kind, n, g.below(1000)


def test_strategies_golden(golden):
    kat = golden("strategy_kat.json")
    for strategy, level, kind, n, seed in kat_cases():
        d = cases.make(kind, n, seed)
        key = "%d/%d/%s/%d/%d" % (strategy, level, kind, n, seed)
        got = [h16(O.deflate_chunk(d, level, last, p0, strategy=strategy)) for last in (False, True) for p0 in (False, True)]
        assert got == kat[key], key


@pytest.mark.skipif(not R.available(), reason="oracle/_ref/libzref.so not built")
def test_strategies_against_reference():
    n_checked = 0
    for strategy, level, kind, n, seed in kat_cases():
        if (n_checked % 3) != 0 and n > 1000:  # a third of the large cases: the golden test covers all of them
            n_checked += 1
            continue
        n_checked += 1
        d = cases.make(kind, n, seed)
        for last in (False, True):
            for p0 in (False, True):
                assert O.deflate_chunk(d, level, last, p0, strategy=strategy) == R.deflate_chunk_raw(d, level, last, p0, strategy=strategy), (strategy, level, kind, n, last, p0)
    d = cases.make("text", 3000, 1)
    for strategy in (1, 2, 3, 4):
        for level in (1, 6, 9):
            assert O.deflate_stream(d, level, strategy=strategy) == R.deflate_wbits(d, level, 15, strategy)  # header flags, deflate.c:628
