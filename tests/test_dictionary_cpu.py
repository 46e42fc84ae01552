"""CPU: the oracle's chunk function with a preset dictionary (deflateSetDictionary, deflate.c:315-354) against the compiled
reference and against golden hashes made by it (tests/golden/dict_kat.json, oracle/gen_golden_dict.py)."""
import hashlib

import pytest

from oracle import cases, oracle_py as O, refzlib as R


def dict_cases():
    g = cases.Lcg(11)
    for lvl in (0, 1, 3, 4, 6, 9):
        for strat in (0, 1, 3):
            for kind in ("text", "rand", "runs", "mix", "period"):
                for dn in (3, 6, 100, 5000, 32506, 40000):
                    for n in (0, 14, 3000, 65536 - min(dn, 32506)):
                        d = cases.make(kind, dn, g.below(1000))
                        x = cases.make(kind, n, g.below(1000))
                        if kind == "text" and n > 100:
                            x = d[:50] + x[50:]  # data that starts like the dictionary
                        yield "%d/%d/%s/%d/%d" % (lvl, strat, kind, dn, n), lvl, strat, d, x


def h16(b):
    return hashlib.sha256(b).hexdigest()[:16]


def test_dictionary_golden(golden):
    kat = golden("dict_kat.json")
    for key, lvl, strat, d, x in dict_cases():
        assert [h16(O.deflate_chunk_dict(d, x, lvl, last, strat)) for last in (False, True)] == kat[key], key


@pytest.mark.skipif(not R.available(), reason="oracle/_ref/libzref.so not built")
def test_dictionary_against_reference():
    for i, (key, lvl, strat, d, x) in enumerate(dict_cases()):
        if i % 4 and len(x) > 5000:
            continue  # a quarter of the large cases here; the golden test covers all
        for last in (False, True):
            assert O.deflate_chunk_dict(d, x, lvl, last, strat) == R.deflate_chunk_dict_raw(d, x, lvl, last, strat), (key, last)
