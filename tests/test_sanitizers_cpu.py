"""ASan + UBSan over the C that parses untrusted input on the host: the product's host library (zip directories, gzip and
zlib headers, gz files) and the checker's C (oracle/*.c on the golden vectors, the corrupted streams among them).  The sanitized libraries are built by
`make -C oracle san` into build/san/ and the tests below are run again, in a child process, with those libraries and the sanitizer runtime preloaded.
CPU only: sanitizers never run on the GPU box."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "build", "san")


def _runtime(name):
    out = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


@pytest.fixture(scope="module")
def env():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not os.path.exists(os.path.join(ROOT, "zlib_amd", "libzamd_gpu.so")):
        pytest.skip("no sanitizer runtime, or the engine library is not built")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "san"])
    e = dict(os.environ)
    e.update(LD_PRELOAD=":".join(x for x in (asan, ubsan) if x), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:allocator_may_return_null=1",
             UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", ZAMD_Z_LIB=os.path.join(SAN, "libzamd_z.so"),
             ZAMD_ORACLE_LIB=os.path.join(SAN, "liboracle.so"), PYTHONPATH=ROOT)
    return e


def _child(env, *tests):
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", *tests], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "passed" in r.stdout


def test_host_parsers_under_asan_ubsan(env):
    _child(env, "tests/test_zip_cpu.py::test_directory_reader_survives_damaged_archives", "tests/test_headers_cpu.py")


def test_oracle_under_asan_ubsan(env):
    _child(env, "tests/test_oracle_golden.py")
