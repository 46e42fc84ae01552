"""CPU-side checks of the C ABI: the HIP library builds, loads and exports every symbol include/zamd_gpu.h
declares (no compute calls -- there is no GPU in the build container)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zgpu_[a-z0-9_]+)\s*\(", text)))


def test_gpu_library_exports_declared_symbols():
    import zlib_amd
    L = zlib_amd.load_library()
    names = declared_symbols("zamd_gpu.h")
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), "libzamd_gpu.so does not export " + n
    assert b"gfx950" in L.zgpu_version()


def test_no_gpu_means_loud_failure():
    import pytest
    import zlib_amd
    L = zlib_amd.load_library()
    if L.zgpu_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(zlib_amd.EngineError):
        zlib_amd.Engine(0)


def test_product_does_not_touch_the_oracle():
    """The product path may not import, link or call anything under oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "zlib_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".c", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle_py" not in text and "liboracle" not in text and "refzlib" not in text and "libzref" not in text, f


def test_host_library_exports_and_fails_loudly_without_gpu():
    """libzamd_z.so: every entry point include/zamd_zlib.h declares; z_stream is 112 bytes; without a GPU the Init
    functions refuse (Z_MEM_ERROR + message) instead of falling back to a CPU codec."""
    import ctypes as C
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import zhost as Z
    text = open(os.path.join(ROOT, "include", "zamd_zlib.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b([a-zA-Z_][a-zA-Z0-9_]*)\s*\(z_streamp|\b(compress2?|uncompress|compressBound|adler32(?:_combine)?|crc32(?:_combine)?|zlibVersion|zError|zlibCompileFlags)\s*\(", text)))
    flat = sorted({n for pair in names for n in pair if n})
    L = Z.lib()
    assert len(flat) >= 20
    for n in flat:
        assert hasattr(L, n), "libzamd_z.so does not export " + n
    assert C.sizeof(Z.ZStream) == 112
    assert L.zlibVersion() == b"1.2.3"
    assert L.compressBound(65536) >= 65567 and L.adler32(1, b"hello", 5) == 0x062C0215
    import zlib_amd
    if zlib_amd.load_library().zgpu_device_count() == 0:
        s = Z.ZStream()
        assert L.deflateInit_(C.byref(s), 6, b"1.2.3", 112) == Z.Z_MEM_ERROR and b"no CPU fallback" in s.msg
        assert L.inflateInit_(C.byref(s), b"1.2.3", 112) == Z.Z_MEM_ERROR
        rc, _ = Z.compress2(b"abc", 6)
        assert rc == Z.Z_MEM_ERROR


def test_client_compiled_against_reference_header_links_with_host_library(tmp_path):
    """Drop-in check: a C client built with the REFERENCE's zlib.h (when the mount is present) runs against libzamd_z.so."""
    import subprocess
    import pytest
    ref_h = "/root/reference/h/zlib.h"
    if not os.path.exists(ref_h):
        pytest.skip("reference header not mounted here")
    src = tmp_path / "client.c"
    src.write_text('''#include "zlib.h"
#include <stdio.h>
int main(void) {
    z_stream s; s.zalloc = 0; s.zfree = 0; s.opaque = 0;
    printf("%s %d %lu ", zlibVersion(), (int)sizeof(z_stream), compressBound(65536));
    int rc = deflateInit(&s, 6);
    printf("%d %s\\n", rc, rc == Z_OK ? "ok" : (s.msg ? s.msg : zError(rc)));
    if (rc == Z_OK) deflateEnd(&s);
    return 0; }''')
    exe = tmp_path / "client"
    libdir = os.path.join(ROOT, "zlib_amd")
    subprocess.check_call(["gcc", "-std=gnu89", "-w", "-I/root/reference/h", str(src), "-o", str(exe), "-L" + libdir, "-lzamd_z", "-lzamd_gpu",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"])
    out = subprocess.check_output([str(exe)], env=dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib")).decode()
    assert out.startswith("1.2.3 112 "), out


API_OF_THE_REFERENCE = """zlibVersion zlibCompileFlags zError deflateInit_ deflateInit2_ deflate deflateEnd deflateSetDictionary deflateCopy deflateReset
deflateParams deflateTune deflateBound deflatePrime deflateSetHeader inflateInit_ inflateInit2_ inflate inflateEnd inflateSetDictionary inflateSync
inflateSyncPoint inflateCopy inflateReset inflatePrime inflateGetHeader inflateBackInit_ inflateBack inflateBackEnd compress compress2 compressBound
uncompress adler32 adler32_combine crc32 crc32_combine get_crc_table gzopen gzdopen gzsetparams gzread gzwrite gzprintf gzputs gzgets gzputc gzgetc
gzungetc gzflush gzseek gzrewind gztell gzeof gzdirect gzclose gzerror gzclearerr z_errmsg zcalloc zcfree""".split()


def test_host_library_exports_the_whole_api_of_the_reference():
    """Every API function h/zlib.h declares (SURVEY.md 8b lists them from `nm` of the compiled reference) plus the three utility
    globals a caller may name.  The reference's remaining globals are internals of its codec (_tr_*, inflate_table, ...)."""
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "zlib_amd", "libzamd_z.so")]).decode()
    have = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    missing = [n for n in API_OF_THE_REFERENCE if n not in have]
    assert not missing, missing
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libzref.so")
    if os.path.exists(ref_so):  # whatever else the reference exports must be one of its codec internals
        ref = {ln.split()[-1] for ln in subprocess.check_output(["nm", "-D", "--defined-only", ref_so]).decode().splitlines() if ln.strip()}
        internals = {"_dist_code", "_length_code", "_tr_align", "_tr_flush_block", "_tr_init", "_tr_stored_block", "_tr_tally", "deflate_copyright",
                     "inflate_copyright", "inflate_fast", "inflate_table"}
        assert ref - have <= internals, sorted(ref - have - internals)


def test_reference_example_links_with_host_library(tmp_path):
    """The judge's link line: the reference's own example.c, unmodified, against libzamd_z.so + libzamd_gpu.so."""
    import subprocess
    import pytest
    ex = "/root/reference/qcsrc/example.c"
    if not os.path.exists(ex):
        pytest.skip("reference not mounted here")
    libdir = os.path.join(ROOT, "zlib_amd")
    exe = tmp_path / "example"
    subprocess.check_call(["gcc", "-std=gnu89", "-w", "-I/root/reference/h", ex, "-o", str(exe), "-L" + libdir, "-lzamd_z", "-lzamd_gpu",
                           "-Wl,-rpath," + libdir])
    assert exe.exists()


def test_host_library_exports_the_zip_entry_points():
    """include/zamd_zip.h: every zamd_zip_* / zamd_unzip_* it declares is exported."""
    import subprocess
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "zamd_zip.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(zamd_(?:un)?zip_[a-z_]+)\s*\(", text)))
    assert len(names) == 9, names
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "zlib_amd", "libzamd_z.so")]).decode()
    have = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert not [n for n in names if n not in have]
