"""Host-side access to the seeded corpus generator (oracle/libcorpus.so).  TEST/BENCH INFRASTRUCTURE."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcorpus.so")
CHUNK = 65536
SEED_SILESIA = 0x5EED5117
SEED_LOGTEXT = 0x10C7E47
KIND_SILESIA, KIND_LOGTEXT = 0, 1
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-s", "-C", _HERE, "libcorpus.so"])
        L = C.CDLL(_SO)
        L.zc_host_fill.argtypes = [C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
        L.zc_host_fill.restype = None
        L.zc_host_class_of.argtypes = [C.c_uint64]
        L.zc_host_class_of.restype = C.c_uint32
        _lib = L
    return _lib


def default_seed(kind):
    return SEED_SILESIA if kind == KIND_SILESIA else SEED_LOGTEXT


def chunks(kind, first_chunk, nchunks, seed=None) -> np.ndarray:
    """uint8 array of nchunks*65536 bytes: chunks first_chunk.. of corpus `kind`."""
    seed = default_seed(kind) if seed is None else seed
    out = np.empty(nchunks * CHUNK // 8, dtype=np.uint64).view(np.uint8)
    lib().zc_host_fill(kind, seed, first_chunk, nchunks, out.ctypes.data)
    return out


def chunk(kind, index, seed=None) -> bytes:
    return chunks(kind, index, 1, seed).tobytes()


def class_of(index):
    return lib().zc_host_class_of(index)
