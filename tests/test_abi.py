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
