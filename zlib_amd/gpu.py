"""ctypes binding of libzamd_gpu.so (C ABI: include/zamd_gpu.h).  Plumbing only."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None

F_FINAL, F_ZLIB_WRAP, F_POS0, F_POS0_ALL, F_GZIP_WRAP, F_CRC32, F_CONTINUOUS = 1, 2, 4, 8, 16, 32, 64
CONT_MORE, CONT_FLUSH, CONT_FINISH = 0, 1, 2  # zgpu_deflate_cont_host modes
WHOLE_STREAM = 0xFFFFFFFF  # inflate chunk_size: the one segment is a complete stream of any size
LZ_AUTO, LZ_SERIAL, LZ_PARALLEL, LZ_SORTED, LZ_WALK, LZ_FAST, LZ_FASTWIN = 0, 1, 2, 3, 4, 5, 6
CHECK_ADLER32, CHECK_CRC32 = 1, 2  # zgpu_inflate_set_checks
STAGES = ["chain", "match", "parse", "lz_serial", "huffman", "stitch", "inflate"]
CHUNK = 65536


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("zgpu error %d: %s" % (code, msg))
        self.code = code


class _Params(C.Structure):
    _fields_ = [("level", C.c_int32), ("chunk_size", C.c_uint32), ("flags", C.c_uint32), ("lz_impl", C.c_int32),
                ("strategy", C.c_int32), ("prime", C.c_uint32)]


class DeflateResult(C.Structure):
    _fields_ = [("out_bytes", C.c_uint64), ("nchunks", C.c_uint64), ("adler32", C.c_uint32), ("data_type", C.c_uint32),
                ("ntokens", C.c_uint64), ("crc32", C.c_uint32), ("reserved", C.c_uint32)]


class ContState(C.Structure):
    """zgpu_cont_state: where a continuous stream stands between two feeds (stream positions)."""
    _fields_ = [("abs0", C.c_uint64), ("entry", C.c_uint64), ("block_start", C.c_uint64), ("carry_ntok", C.c_uint32), ("bit_count", C.c_uint32),
                ("bit_value", C.c_uint32), ("data_type", C.c_uint32), ("first_block", C.c_uint32), ("last_eob", C.c_uint32)]


class InflateResult(C.Structure):
    _fields_ = [("out_bytes", C.c_uint64), ("adler32", C.c_uint32), ("first_bad_chunk", C.c_int32),
                ("error_code", C.c_int32), ("error_msg", C.c_uint32), ("crc32", C.c_uint32), ("in_used_bits", C.c_uint32),
                ("in_used", C.c_uint64), ("stream_end", C.c_uint32), ("incomplete", C.c_uint32)]


def library_path():
    # ZAMD_GPU_LIB: load another build of the engine (A/B measurements of kernel variants)
    return os.environ.get("ZAMD_GPU_LIB") or os.path.join(_HERE, "libzamd_gpu.so")


def load_library():
    """Load the HIP engine.  Fails loudly when it has not been built (no fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process.  The torch wheel bundles its own libamdhip64.so.7 (same SONAME as /opt/rocm's);
    # whichever is mapped first serves both, so when torch is installed it must come first -- loading the system
    # runtime first and torch's afterwards leaves two runtimes in the process and torch then sees no GPU.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = library_path()
    if not os.path.exists(path):
        raise ImportError("%s is missing: build it with `make -C zlib_amd/csrc` (or __graft_entry__.build())" % path)
    L = C.CDLL(path)
    vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
    L.zgpu_device_count.restype = C.c_int
    L.zgpu_engine_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.zgpu_engine_destroy.argtypes = [vp]
    L.zgpu_engine_destroy.restype = None
    L.zgpu_engine_error.argtypes = [vp]
    L.zgpu_engine_error.restype = C.c_char_p
    L.zgpu_version.restype = C.c_char_p
    L.zgpu_deflate_bound.argtypes = [u64, u32]
    L.zgpu_deflate_bound.restype = u64
    L.zgpu_deflate_bound_geometry.argtypes = [u64, u32, C.c_int, C.c_int]
    L.zgpu_deflate_bound_geometry.restype = u64
    L.zgpu_deflate_set_geometry.argtypes = [vp, C.c_int, C.c_int]
    L.zgpu_deflate_device.argtypes = [vp, vp, u64, C.POINTER(_Params), vp, u64, vp, C.POINTER(DeflateResult), vp]
    L.zgpu_deflate_host.argtypes = [vp, vp, u64, C.POINTER(_Params), vp, u64, vp, C.POINTER(DeflateResult)]
    L.zgpu_deflate_cont_bound.argtypes = [u64]
    L.zgpu_deflate_cont_bound.restype = u64
    L.zgpu_deflate_cont_host.argtypes = [vp, vp, u64, vp, u64, u64, C.POINTER(_Params), C.c_int, C.POINTER(ContState), vp, vp, vp, u32, vp, u64, C.POINTER(DeflateResult)]
    L.zgpu_deflate_segments_host.argtypes = [vp, vp, vp, u64, C.POINTER(_Params), vp, u64, vp, C.POINTER(DeflateResult)]
    L.zgpu_deflate_segments_device.argtypes = [vp, vp, u64, vp, u64, C.POINTER(_Params), vp, u64, vp, C.POINTER(DeflateResult), vp]
    L.zgpu_inflate_device.argtypes = [vp, vp, u64, vp, u64, u32, vp, u64, C.POINTER(InflateResult), vp]
    L.zgpu_inflate_host.argtypes = [vp, vp, u64, vp, u64, u32, vp, u64, C.POINTER(InflateResult)]
    L.zgpu_inflate_find_chunks_host.argtypes = [vp, vp, u64, u32, vp, u64, C.POINTER(u64)]
    L.zgpu_inflate_stream_host2.argtypes = [vp, vp, u64, u32, vp, u64, C.POINTER(InflateResult)]
    L.zgpu_inflate_stream_host3.argtypes = [vp, vp, u64, u32, u32, vp, u64, C.POINTER(InflateResult)]
    L.zgpu_comm_unique_id.argtypes = [vp]
    L.zgpu_comm_create.argtypes = [C.c_int, C.c_int, C.c_int, vp, C.POINTER(vp)]
    L.zgpu_comm_destroy.argtypes = [vp]
    L.zgpu_comm_destroy.restype = None
    L.zgpu_comm_error.restype = C.c_char_p
    L.zgpu_gather_layout.argtypes = [C.c_int, vp, vp, C.POINTER(u64)]
    L.zgpu_gather_layout.restype = None
    L.zgpu_deflate_gather_sizes.argtypes = [vp, u64, u32, u64, vp, C.POINTER(u64), vp]
    L.zgpu_deflate_gather.argtypes = [vp, vp, vp, C.c_int, vp, u64, C.POINTER(u32), vp]
    L.zgpu_inflate_spec_count.argtypes = [C.c_int]
    L.zgpu_inflate_spec_count.restype = u64
    L.zgpu_inflate_message.argtypes = [u32]
    L.zgpu_inflate_message.restype = C.c_char_p
    L.zgpu_adler32_device.argtypes = [vp, vp, u64, C.POINTER(u32), vp]
    L.zgpu_crc32_device.argtypes = [vp, vp, u64, C.POINTER(u32), vp]
    L.zgpu_deflate_dict_chunk_host.argtypes = [vp, vp, u32, u32, vp, vp, u64, vp]
    L.zgpu_inflate_set_dictionary.argtypes = [vp, vp, u32]
    L.zgpu_inflate_set_checks.argtypes = [vp, u32]
    L.zgpu_profile_enable.argtypes = [vp, C.c_int]
    L.zgpu_profile_enable.restype = None
    L.zgpu_profile_reset.argtypes = [vp]
    L.zgpu_profile_reset.restype = None
    L.zgpu_profile_get.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(u64)]
    L.zgpu_stage_name.argtypes = [C.c_int]
    L.zgpu_stage_name.restype = C.c_char_p
    L.zgpu_corpus_fill_device.argtypes = [vp, u32, u64, u64, u64, vp, vp]
    _lib = L
    return L


class Engine:
    """One engine per process and GPU (one process per GPU is the deployment model)."""

    def __init__(self, device=0):
        self.L = load_library()
        h = C.c_void_p()
        rc = self.L.zgpu_engine_create(device, C.byref(h))
        if rc != 0:
            raise EngineError(rc, "cannot create engine on device %d (%d visible)" % (device, self.L.zgpu_device_count()))
        self.h = h

    def close(self):
        if self.h:
            self.L.zgpu_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise EngineError(rc, self.L.zgpu_engine_error(self.h).decode())

    def inflate_set_dictionary(self, dictionary):
        d = bytes(dictionary) if dictionary else b""
        self._check(self.L.zgpu_inflate_set_dictionary(self.h, d if d else None, len(d)))

    def inflate_set_checks(self, mask):
        """Which checks of the decoded bytes the inflate calls that follow compute: CHECK_ADLER32 | CHECK_CRC32 (default both)."""
        self._check(self.L.zgpu_inflate_set_checks(self.h, mask))

    def deflate_dict_chunk_host(self, dictionary, chunk, level, final, strategy=0, flags=0):
        """One chunk behind a preset dictionary (its last 32506 bytes count); returns the raw deflate stream of `chunk`."""
        import numpy as np
        d = bytes(dictionary)[-32506:]
        buf = np.frombuffer(d + bytes(chunk) + b"\0", dtype=np.uint8)
        n = len(d) + len(chunk)
        cap = self.L.zgpu_deflate_bound(n, CHUNK)
        out = np.empty(cap, dtype=np.uint8)
        p = _Params(level, CHUNK, flags | (F_FINAL if final else 0), LZ_AUTO, strategy, 0)
        res = DeflateResult()
        self._check(self.L.zgpu_deflate_dict_chunk_host(self.h, buf.ctypes.data, n, len(d), C.byref(p), out.ctypes.data, cap, C.byref(res)))
        self.last = res
        return out[: res.out_bytes].tobytes()

    # ---- deflate ----
    def deflate_host(self, data, level, flags=F_FINAL | F_ZLIB_WRAP, chunk_size=CHUNK, lz_impl=LZ_AUTO, want_offsets=False, strategy=0, prime=(0, 0)):
        """data: bytes-like or numpy uint8 array.  Returns bytes (and the chunk offsets when asked)."""
        import numpy as np
        arr = np.frombuffer(data, dtype=np.uint8) if not hasattr(data, "ctypes") else data
        n = int(arr.size)
        cap = self.L.zgpu_deflate_bound_geometry(n, chunk_size, *getattr(self, "geometry", (15, 8)))
        if flags & F_CONTINUOUS:
            cap = self.L.zgpu_deflate_cont_bound(n) + 32
        out = np.empty(cap, dtype=np.uint8)
        nchunks = max(1, (n + chunk_size - 1) // chunk_size)
        offs = np.zeros(nchunks + 1, dtype=np.uint64)
        p = _Params(level, chunk_size, flags, lz_impl, strategy, (prime[0] << 16) | (prime[1] & 0xffff))  # prime = (nbits, value): deflatePrime
        res = DeflateResult()
        src = arr.ctypes.data if n else None
        self._check(self.L.zgpu_deflate_host(self.h, src, n, C.byref(p), out.ctypes.data, cap,
                                             offs.ctypes.data if want_offsets else None, C.byref(res)))
        self.last = res
        z = out[: res.out_bytes].tobytes()
        return (z, offs) if want_offsets else z

    def cont_new(self):
        """State of a fresh continuous stream (no dictionary) + its token carry."""
        import numpy as np
        cs = ContState(0, 0, 0, 0, 0, 0, 2, 1, 8)
        return cs, (np.zeros(16384, dtype=np.uint32), np.zeros(1040, dtype=np.uint32))  # the block's tokens so far; which history positions are in the chains (levels 1-3)

    def deflate_cont_host(self, buf, check_from, level, mode, cs, carry, strategy=0, flags=0, excl=(), split=None):
        """One feed of a continuous stream (zgpu_deflate_cont_host): buf = the history the parse can still reach + the unparsed bytes, buf[0] at stream
        position cs.abs0 (handed over as two pieces cut at `split`, default: all of it as history-less input).  Returns the whole bytes the feed
        wrote; cs and carry move on.  self.last has the checksums of buf[check_from:]."""
        import numpy as np
        arr = np.frombuffer(bytes(buf) + b"\0", dtype=np.uint8)
        n = int(arr.size) - 1
        split = 0 if split is None else min(max(split, 0), n)
        cap = self.L.zgpu_deflate_cont_bound(n) + 64
        out = np.empty(cap, dtype=np.uint8)
        p = _Params(level, 0, flags, LZ_AUTO, strategy, 0)
        res = DeflateResult()
        ex = np.ascontiguousarray(list(excl) + [0], dtype=np.uint64)
        self._check(self.L.zgpu_deflate_cont_host(self.h, arr.ctypes.data, split, arr.ctypes.data + split, n - split, check_from, C.byref(p), mode, C.byref(cs), carry[0].ctypes.data,
                                                  carry[1].ctypes.data, ex.ctypes.data, len(excl), out.ctypes.data, cap, C.byref(res)))
        self.last = res
        return out[: res.out_bytes].tobytes()

    def set_geometry(self, window_bits=15, mem_level=8):
        """deflateInit2's windowBits (9..15) and memLevel (1..9) for the deflate calls that follow; 15 / 8 is the default."""
        self._check(self.L.zgpu_deflate_set_geometry(self.h, window_bits, mem_level))
        self.geometry = (window_bits, mem_level)

    def deflate_segments_host(self, buffers, level, flags=0, lz_impl=LZ_AUTO):
        """Batch of independent buffers (each <= 65536 bytes) -> list of raw-deflate segments, one launch."""
        import numpy as np
        sizes = [len(b) for b in buffers]
        offs = np.zeros(len(buffers) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum(sizes)
        blob = np.frombuffer(b"".join(buffers) + b"\0", dtype=np.uint8)
        cap = int(offs[-1]) + 40 * len(buffers) + 64
        if getattr(self, "geometry", (15, 8)) != (15, 8):
            cap = int(offs[-1]) + len(buffers) * (12288 + 5 * 520)
        out = np.empty(cap, dtype=np.uint8)
        ooffs = np.zeros(len(buffers) + 1, dtype=np.uint64)
        p = _Params(level, 0, flags, lz_impl)
        res = DeflateResult()
        self._check(self.L.zgpu_deflate_segments_host(self.h, blob.ctypes.data, offs.ctypes.data, len(buffers), C.byref(p),
                                                      out.ctypes.data, cap, ooffs.ctypes.data, C.byref(res)))
        self.last = res
        raw = out[: res.out_bytes].tobytes()
        return [raw[int(ooffs[i]): int(ooffs[i + 1])] for i in range(len(buffers))]

    def deflate_device(self, d_in, n, level, d_out, out_cap, flags=F_FINAL | F_ZLIB_WRAP, chunk_size=CHUNK, lz_impl=LZ_AUTO,
                       d_offsets=None, stream=None):
        """d_in / d_out / d_offsets: device pointers as ints (e.g. torch.Tensor.data_ptr())."""
        p = _Params(level, chunk_size, flags, lz_impl)
        res = DeflateResult()
        self._check(self.L.zgpu_deflate_device(self.h, d_in, n, C.byref(p), d_out, out_cap, d_offsets, C.byref(res), stream))
        return res

    # ---- inflate ----
    def inflate_host(self, data, offsets, chunk_size=CHUNK, out_len=None):
        import numpy as np
        arr = np.frombuffer(data, dtype=np.uint8)
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        nchunks = len(offs) - 1
        cap = nchunks * chunk_size if out_len is None else out_len
        out = np.empty(max(cap, 1), dtype=np.uint8)
        res = InflateResult()
        rc = self.L.zgpu_inflate_host(self.h, arr.ctypes.data, arr.size, offs.ctypes.data, nchunks, chunk_size, out.ctypes.data, cap,
                                      C.byref(res))
        self.last_inflate = res
        if rc != 0:
            msg = self.L.zgpu_inflate_message(res.error_msg).decode() if rc == -3 else self.L.zgpu_engine_error(self.h).decode()
            raise EngineError(rc, msg)
        return out[: res.out_bytes].tobytes()

    def inflate_stream_host(self, body, out_cap, flags=0, out=None):
        """A raw deflate body without a side table (zgpu_inflate_stream_host2): split at its flush markers if it has them, in pieces at
        block starts found by search if it is long enough, by one workgroup otherwise.  Returns the bytes; self.last_inflate has the rest."""
        import numpy as np
        arr = np.frombuffer(body, dtype=np.uint8)
        given = out is not None
        if not given:
            out = np.empty(max(out_cap, 1), dtype=np.uint8)
        res = InflateResult()
        rc = self.L.zgpu_inflate_stream_host2(self.h, arr.ctypes.data, arr.size, flags, out.ctypes.data, out_cap, C.byref(res))
        self.last_inflate = res
        if rc != 0:
            msg = self.L.zgpu_inflate_message(res.error_msg).decode() if rc == -3 else self.L.zgpu_engine_error(self.h).decode()
            raise EngineError(rc, msg)
        return out[: res.out_bytes] if given else out[: res.out_bytes].tobytes()

    def spec_counts(self):
        return int(self.L.zgpu_inflate_spec_count(0)), int(self.L.zgpu_inflate_spec_count(1))

    def inflate_device(self, d_in, n, d_offsets, nchunks, d_out, out_cap, chunk_size=CHUNK, stream=None):
        res = InflateResult()
        rc = self.L.zgpu_inflate_device(self.h, d_in, n, d_offsets, nchunks, chunk_size, d_out, out_cap, C.byref(res), stream)
        if rc != 0:
            msg = self.L.zgpu_inflate_message(res.error_msg).decode() if rc == -3 else self.L.zgpu_engine_error(self.h).decode()
            raise EngineError(rc, msg)
        return res

    def find_chunks_host(self, body, chunk_size=CHUNK, max_chunks=None):
        import numpy as np
        arr = np.frombuffer(body, dtype=np.uint8)
        max_chunks = max_chunks or (arr.size // 5 + 2)
        offs = np.zeros(max_chunks + 1, dtype=np.uint64)
        n = C.c_uint64(0)
        self._check(self.L.zgpu_inflate_find_chunks_host(self.h, arr.ctypes.data, arr.size, chunk_size, offs.ctypes.data, max_chunks,
                                                         C.byref(n)))
        return offs[: n.value + 1]

    # ---- misc ----
    def adler32_device(self, d_in, n, stream=None):
        a = C.c_uint32(0)
        self._check(self.L.zgpu_adler32_device(self.h, d_in, n, C.byref(a), stream))
        return a.value

    def crc32_device(self, d_in, n, stream=None):
        a = C.c_uint32(0)
        self._check(self.L.zgpu_crc32_device(self.h, d_in, n, C.byref(a), stream))
        return a.value

    def corpus_fill_device(self, kind, seed, first_chunk, nchunks, d_out, stream=None):
        self._check(self.L.zgpu_corpus_fill_device(self.h, kind, seed, first_chunk, nchunks, d_out, stream))

    def profile(self, on=True):
        self.L.zgpu_profile_enable(self.h, int(on))
        self.L.zgpu_profile_reset(self.h)

    def profile_read(self):
        out = {}
        for i, name in enumerate(STAGES):
            ms, n = C.c_double(0), C.c_uint64(0)
            self.L.zgpu_profile_get(self.h, i, C.byref(ms), C.byref(n))
            out[name] = (ms.value, n.value)
        return out


def gather_layout(table):
    """table: world rows of (body bytes, Adler-32, input bytes).  Returns (offsets[world + 1], total): where every rank's body starts in the gathered
    stream, where the trailer goes, the stream's length -- the C library's arithmetic (zgpu_gather_layout), which the RCCL gather itself uses."""
    import numpy as np
    L = load_library()
    t = np.ascontiguousarray(table, dtype=np.uint64).reshape(-1, 3)
    offs = np.zeros(len(t) + 1, dtype=np.uint64)
    total = C.c_uint64(0)
    L.zgpu_gather_layout(len(t), t.ctypes.data, offs.ctypes.data, C.byref(total))
    return [int(x) for x in offs], int(total.value)


class Comm:
    """The RCCL communicator of the C library (include/zamd_gpu.h zgpu_comm_*): one per process and GPU.  `exchange_id` hands rank 0's 128-byte id
    to the other ranks -- any channel will do; bench.py uses the torch.distributed group it has for its barrier."""

    def __init__(self, device, world, rank, exchange_id):
        import numpy as np
        self.L = load_library()
        self.world, self.rank = world, rank
        ident = np.zeros(128, dtype=np.uint8)
        rc0 = self.L.zgpu_comm_unique_id(ident.ctypes.data) if rank == 0 else 0
        if rc0 != 0:
            ident[:] = 0
        ident = np.frombuffer(exchange_id(ident.tobytes()), dtype=np.uint8).copy()  # (also when rank 0 has no id: the others must not wait for it)
        if rc0 != 0 or not ident.any():
            raise EngineError(rc0 or -2, self.L.zgpu_comm_error().decode() if rank == 0 else "rank 0 could not make an RCCL id")
        h = C.c_void_p()
        rc = self.L.zgpu_comm_create(device, world, rank, ident.ctypes.data, C.byref(h))
        if rc != 0:
            raise EngineError(rc, self.L.zgpu_comm_error().decode())
        self.h = h

    def sizes(self, body_bytes, adler, in_bytes, stream=None):
        import numpy as np
        table = np.zeros(self.world * 3, dtype=np.uint64)
        total = C.c_uint64(0)
        rc = self.L.zgpu_deflate_gather_sizes(self.h, body_bytes, adler, in_bytes, table.ctypes.data, C.byref(total), stream)
        if rc != 0:
            raise EngineError(rc, self.L.zgpu_comm_error().decode())
        return table, int(total.value)

    def gather(self, d_body, table, level, d_out=None, out_cap=0, stream=None):
        adler = C.c_uint32(0)
        rc = self.L.zgpu_deflate_gather(self.h, d_body, table.ctypes.data, level, d_out, out_cap, C.byref(adler), stream)
        if rc != 0:
            raise EngineError(rc, self.L.zgpu_comm_error().decode())
        return int(adler.value)

    def close(self):
        if self.h:
            self.L.zgpu_comm_destroy(self.h)
            self.h = None
