"""GPU: the alternative kernel paths of the levels 4-9 pipeline must produce the same bytes as the default one.

  ZGPU_SORT=1            sort_kernel (ballots only) instead of sort3_kernel (ordered LDS atomics + self-check)
  ZTEST_SORT_FAULT=1 (read by the child script, which calls zgpu_debug_inject_sort_fault): sort3's self-check reports a fault -> the engine redoes the call with sort_kernel
  ZGPU_PARSE=1           parse_kernel (the reference loop, one lane per chunk) instead of parse2_kernel

The switches are read once per process, so every variant runs in a child process (one at a time) and prints the
SHA-256 of its streams; the default variant's digests are additionally pinned to the oracle in this process."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import hashlib, json, sys
sys.path.insert(0, %r)
from oracle import cases, corpus_py as CP
import zlib_amd
e = zlib_amd.Engine(0)
import os
if os.environ.get("ZTEST_SORT_FAULT"):
    e.L.zgpu_debug_inject_sort_fault.restype = None
    e.L.zgpu_debug_inject_sort_fault()
out = {}
inputs = {
    "corpus0": CP.chunks(0, 0, 24).tobytes(),
    "corpus1": CP.chunks(1, 3, 8).tobytes(),
    "hello": cases.hello_1mib()[: 5 * 65536 + 77],
    "zeros": bytes(3 * 65536 + 5),
    "ragged": CP.chunks(0, 40, 2).tobytes()[: 65536 + 2],
}
for name, data in inputs.items():
    for lvl in (4, 6, 9):
        out["%%s/%%d" %% (name, lvl)] = hashlib.sha256(e.deflate_host(data, lvl)).hexdigest()
print("DIGESTS " + json.dumps(out, sort_keys=True))
""" % ROOT


def run_variant(env_extra):
    env = dict(os.environ)
    for k in ("ZGPU_SORT", "ZTEST_SORT_FAULT", "ZGPU_PARSE", "ZGPU_BATCH_CHUNKS"):
        env.pop(k, None)
    env.update(env_extra)
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("DIGESTS ")][-1]
    return json.loads(line[len("DIGESTS "):])


def test_alternative_paths_agree():
    base = run_variant({})
    # pin the default path to the oracle for two of the inputs (the rest of the suite does this at length)
    from oracle import corpus_py as CP, oracle_py as O
    data = CP.chunks(0, 0, 24).tobytes()
    assert base["corpus0/6"] == hashlib.sha256(O.deflate_stream(data, 6, 65536)).hexdigest()
    assert base["zeros/9"] == hashlib.sha256(O.deflate_stream(bytes(3 * 65536 + 5), 9, 65536)).hexdigest()
    # ... and a batch size of 7 chunks: every multi-chunk input above then takes several launches of every stage
    for env in ({"ZGPU_SORT": "1"}, {"ZTEST_SORT_FAULT": "1"}, {"ZGPU_PARSE": "1"}, {"ZGPU_SORT": "1", "ZGPU_PARSE": "1"}, {"ZGPU_BATCH_CHUNKS": "7"}):
        got = run_variant(env)
        assert got == base, "variant %r differs: %s" % (env, [k for k in base if got.get(k) != base[k]])


def test_chunks_the_loop_hands_on_come_out_the_same(monkeypatch):
    """Levels 1-3, large calls: the lane-per-chunk loop gives chunks that do not compress to the wave-per-chunk kernel (lz_serial_chunk's hand_on, a list
    launch of sort3 + fastwin over ChunkGeom::chunk_map).  Forced here at test size (ZGPU_HAND_ON=2): a call of compressible, incompressible and
    half-and-half chunks, ragged at the end, against the same call with the hand-on switched off and against the oracle; the counter says it happened."""
    import ctypes as C
    import numpy as np
    import zlib_amd
    from zlib_amd import gpu
    from oracle import cases, corpus_py as CP, oracle_py as O
    parts = [CP.chunks(CP.KIND_SILESIA, 5, 6).reshape(-1).tobytes(), cases.make("rand", 3 * 65536, 9), cases.make("text", 65536, 3),
             cases.make("rand", 40000, 4) + cases.make("text", 25536, 5), cases.make("text", 30000, 6) + cases.make("rand", 35536, 7),
             cases.make("rand", 65536, 8), cases.make("mix", 65536, 10), cases.make("rand", 12345, 11)]
    data = np.frombuffer(b"".join(parts), dtype=np.uint8)
    e = zlib_amd.Engine(0)
    e.L.zgpu_debug_handed_on.argtypes = [C.c_void_p]
    e.L.zgpu_debug_handed_on.restype = C.c_uint64
    try:
        for level in (1, 2, 3):
            plain, offs = e.deflate_host(data, level, want_offsets=True, lz_impl=gpu.LZ_SERIAL)  # (asked for by name, the loop keeps every chunk)
            monkeypatch.setenv("ZGPU_HAND_ON", "2")
            before = e.L.zgpu_debug_handed_on(e.h)
            handed, offs2 = e.deflate_host(data, level, want_offsets=True)
            n_handed = e.L.zgpu_debug_handed_on(e.h) - before
            assert handed == plain and list(offs) == list(offs2)
            assert 5 <= n_handed <= 9, n_handed  # the chunks that begin with random bytes, and those that turn random later
            assert handed == O.deflate_stream(data.tobytes(), level)
            monkeypatch.delenv("ZGPU_HAND_ON")
    finally:
        e.close()
