"""The PKZIP fixture (tests/golden/zip_kat.json, archives written by the reference's minizip) against an independent reader:
Python's zipfile must list and extract every member with the inputs oracle/cases.py regenerates.  Pins the fixture the GPU tests
(tests/test_gpu_zip.py) compare the product's archives with."""
import base64
import io
import json
import os
import zipfile

import pytest

from oracle import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KAT = json.load(open(os.path.join(ROOT, "tests", "golden", "zip_kat.json")))


@pytest.mark.parametrize("arc", KAT["archives"], ids=lambda a: "L%d" % a["level"])
def test_fixture_archives_hold_the_case_inputs(arc):
    zf = zipfile.ZipFile(io.BytesIO(base64.b64decode(arc["zip_b64"])))
    assert zf.testzip() is None
    assert [i.filename for i in zf.infolist()] == [m[0] for m in KAT["members"]]
    for (name, kind, n, seed), info in zip(KAT["members"], zf.infolist()):
        assert zf.read(name) == cases.make(kind, n, seed)
        assert info.compress_type == (zipfile.ZIP_DEFLATED if arc["level"] else zipfile.ZIP_STORED)
        assert info.date_time == (2005, 7, 18, 12, 34, 56)
        want_flag = {8: 2, 9: 2, 2: 4, 1: 6}.get(arc["level"], 0)  # zip.c:760-766
        assert info.flag_bits == want_flag
