"""The continuous stream on the CPU: the restatement (oracle/deflate_oracle.c, ora_deflate_cont) against the compiled reference and the golden streams, and
the tile decomposition the device uses (tests/tools/cont_tile_model.c) against the restatement."""
import hashlib
import json
import os
import random
import subprocess

import pytest

from oracle import corpus_py as CP, oracle_py as O, refzlib as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
SEEDS = {0: 0x5EED5117, 1: 0x10C7E47}


def corpus(kind, seed, nbytes):
    return CP.chunks(kind, seed, (nbytes + 65535) // 65536).tobytes()[:nbytes]


def test_restatement_reproduces_the_golden_streams():
    kat = json.load(open(os.path.join(GOLD, "continuous_kat.json")))
    for r in kat["rows"]:
        if (r["n"] > (1 << 20) and r["level"] not in (1, 6)) or r["n"] > (16 << 20):
            continue  # (the 16 MiB rows at two levels, the bench's 256 MiB rows not at all: the suite stays short)
        d = corpus(r["corpus"], SEEDS[r["corpus"]], r["n"])
        calls = () if r["sync_at"] is None else ((r["sync_at"], 2),)
        z = O.cont_stream(d, r["level"], calls)
        assert (len(z), hashlib.sha256(z).hexdigest()) == (r["len"], r["sha256"]), r


@pytest.mark.skipif(not R.available(), reason="the compiled reference (oracle/_ref) is not here")
def test_restatement_against_the_compiled_reference_call_by_call():
    rnd = random.Random(7)
    d = corpus(0, 11, 2 << 20)
    for it in range(10):
        level = rnd.choice([0, 1, 2, 3, 4, 5, 6, 7, 8, 9])
        strat = rnd.choice([0, 0, 1, 2, 3, 4])
        calls, pos = [], 0
        while True:
            pos += rnd.choice([1, 5, 100, 4096, 32768 - 262 + rnd.randrange(0, 300), 65536, 200000])
            if pos >= len(d):
                break
            calls.append((pos, rnd.choice([0, 0, 0, 1, 2, 3])))
        dic = rnd.choice([None, None, d[5000:9000], d[:40000]])
        assert R.deflate_calls(d, level, calls, strategy=strat, dictionary=dic) == O.deflate_cont(d, level, calls, strategy=strat, dictionary=dic or b""), (level, strat)
    # deflateParams between the calls (deflate.c:416-451), the compress function changing or not
    for l0, l1, st in ((1, 9, 1), (6, 1, 0), (6, 9, 0), (0, 6, 0), (6, 0, 0), (9, 3, 0)):
        for calls in (((100000, 0), (200000, 2)), ((100000, 3), (150000, 1))):
            for k in (1, 2):
                assert R.deflate_calls(d[:300001], l0, calls, params={k: (l1, st)}) == O.deflate_cont(d[:300001], l0, calls, params={k: (l1, st)}), (l0, l1, st, k)
    for wbits in (15, -15, 31):
        assert R.deflate_calls(d[:200000], 6, ((100000, 2),), wbits=wbits) == O.cont_stream(d[:200000], 6, ((100000, 2),), wbits=wbits)


def test_tile_model_is_the_reference_loop(tmp_path):
    """The decomposition the device uses -- 64 KiB tiles that parse 32512 positions each with 32512 bytes of history, slides / NIL / stored-block veto as
    functions of the absolute position -- gives the bytes of the reference's loop, at sizes around every threshold, on data whose first chain candidate lies
    exactly MAX_DIST back."""
    exe = str(tmp_path / "ctm")
    subprocess.check_call(["gcc", "-O2", "-std=c99", "-o", exe, os.path.join(ROOT, "tests", "tools", "cont_tile_model.c"), os.path.join(ROOT, "oracle", "checksum_oracle.c")])
    rnd = random.Random(3)
    base = bytes(rnd.getrandbits(8) for _ in range(32506))
    srcs = {"per": base * 5, "sil": corpus(0, 21, 200000), "rnd": bytes(rnd.getrandbits(8) for _ in range(140000))}
    sizes = {65274 + d for d in (0, 1, 259, 260, 261, 262, 263)} | {98042 + d for d in (0, 261, 262)} | {1, 3, 262, 32512, 65024, 65025, 65536, 97537, 140000}
    f = str(tmp_path / "in.bin")
    for name, src in srcs.items():
        for n in sorted(sizes):
            if n > len(src):
                continue
            open(f, "wb").write(src[:n])
            for level in (4, 6, 9):
                p = subprocess.run([exe, str(level), f], capture_output=True, text=True)
                assert p.returncode == 0, (name, n, level, p.stdout, p.stderr)
