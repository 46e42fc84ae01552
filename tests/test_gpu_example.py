"""The reference's own test program, example.c (qcsrc/example.c:59-565: compress/uncompress, gz* file I/O, deflate and inflate with
small buffers, large buffers with deflateParams, full flush + inflateSync, preset dictionaries), compiled UNMODIFIED from the mount by
oracle/Makefile and linked with the product's libzamd_z.so.  Its output must be the output of the same program linked with the
reference (tests/golden/example_out.txt, produced by oracle/_ref/example_ref)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_example_c_runs_against_the_product(tmp_path):
    exe = os.path.join(ROOT, "oracle", "_ref", "example_zamd")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/example_zamd was not built (no reference mount at build time)")
    libdir = os.path.join(ROOT, "zlib_amd")
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    p = subprocess.run([exe, str(tmp_path / "foo.gz")], cwd=str(tmp_path), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0, out
    want = open(os.path.join(ROOT, "tests", "golden", "example_out.txt")).read()
    assert out.splitlines() == want.splitlines(), out
