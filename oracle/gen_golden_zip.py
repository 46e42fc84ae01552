#!/usr/bin/env python3
"""Golden PKZIP archives written by the reference's minizip (qcsrc/minizip.c + zip.c + ioapi.c, compiled from the mount into
oracle/_ref/minizip_ref by oracle/Makefile, linked with the compiled reference library):  python oracle/gen_golden_zip.py
->  tests/golden/zip_kat.json
Every member is at most one 64 KiB chunk, so the reference's single stream and the product's chunked stream are the same bytes and the whole
archive is comparable byte for byte.  File times are fixed and TZ=UTC, so the DOS dates are reproducible.
TEST INFRASTRUCTURE ONLY."""
import base64
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cases  # noqa: E402

MEMBERS = [("a.txt", "text", 20000, 11), ("b.bin", "rand", 3000, 12), ("empty", "zeros", 0, 13), ("sub/c.dat", "ab", 65536, 14), ("d.mix", "mix", 40000, 15),
           ("e.run", "runs", 65535, 16)]
MTIME = 1121690096  # 2005-07-18 12:34:56 UTC
LEVELS = [0, 1, 2, 6, 9]


def main():
    exe = os.path.join(ROOT, "oracle", "_ref", "minizip_ref")
    out = {"comment": "archives written by the reference's minizip (oracle/gen_golden_zip.py)", "mtime": MTIME, "members": [list(m) for m in MEMBERS], "archives": []}
    env = dict(os.environ, TZ="UTC", LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref"))
    for level in LEVELS:
        with tempfile.TemporaryDirectory() as d:
            os.makedirs(os.path.join(d, "sub"))
            for name, kind, n, seed in MEMBERS:
                p = os.path.join(d, name)
                open(p, "wb").write(cases.make(kind, n, seed))
                os.utime(p, (MTIME, MTIME))
            subprocess.run([exe, "-o", "-%d" % level, "t.zip"] + [m[0] for m in MEMBERS], cwd=d, env=env, check=True, stdout=subprocess.DEVNULL)
            z = open(os.path.join(d, "t.zip"), "rb").read()
        dos = int.from_bytes(z[10:14], "little")
        out["archives"].append({"level": level, "dos_date": dos, "zip_b64": base64.b64encode(z).decode()})
        print("level %d: %d bytes, dos date %#x" % (level, len(z), dos))
    with open(os.path.join(ROOT, "tests", "golden", "zip_kat.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
