"""zlib_amd -- MI355X-native DEFLATE engine (zlib 1.2.3 bit-exact, 64 KiB independent chunks).

The product is native: zlib_amd/csrc/*.hip (HIP kernels + C ABI, include/zamd_gpu.h) and the
zlib-compatible C host library (include/zamd_zlib.h).  This package only loads those shared objects through
ctypes for tests and benchmarks; it contains no codec logic and no CPU fallback -- if the HIP library is
missing or no GPU is present, engine creation fails loudly.
"""
from .gpu import Engine, EngineError, DeflateResult, load_library, library_path  # noqa: F401

__all__ = ["Engine", "EngineError", "DeflateResult", "load_library", "library_path"]
