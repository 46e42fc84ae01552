"""BASELINE.json configs 2-5 at their full size (4 GiB of the seeded Silesia-mix, generated in HBM): the engine's
chunk segments are checked against the reference's golden hashes for all 4096 sampled chunk ids, the stream inflates
back to the input on the device, and the Adler-32 trailer matches a checksum of checksums computed independently."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_4gib_level6_against_reference_hashes_and_roundtrip():
    import torch
    import zlib_amd
    from zlib_amd import gpu
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "corpus_silesia.json")))
    nchunks = g["total_chunks"]  # 65536 chunks = 4 GiB
    nbytes = nchunks * 65536
    eng = zlib_amd.Engine(0)
    dev = torch.device("cuda", 0)
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    eng.corpus_fill_device(g["kind"], g["seed"], 0, nchunks, src.data_ptr())
    cap = eng.L.zgpu_deflate_bound(nbytes, 65536)
    dst = torch.empty(cap, dtype=torch.uint8, device=dev)
    offs = torch.empty(nchunks + 1, dtype=torch.int64, device=dev)
    for level, col in ((6, 4), (1, 2), (9, 6)):
        res = eng.deflate_device(src.data_ptr(), nbytes, level, dst.data_ptr(), cap, flags=gpu.F_FINAL | gpu.F_ZLIB_WRAP, d_offsets=offs.data_ptr())
        assert res.nchunks == nchunks
        o = offs.cpu().numpy()
        assert o[0] == 2 and o[-1] + 4 == res.out_bytes
        # every sampled chunk; the stream's very last chunk carries BFINAL instead of the flush marker: the fixture's `last_rows` hold that variant
        last = [r for r in g["last_rows"] if r[0] == nchunks - 1]
        assert len(last) == 1
        for row in [r for r in g["rows"] if r[0] != nchunks - 1] + last:
            k = row[0]
            seg = dst[int(o[k]): int(o[k + 1])].cpu().numpy().tobytes()
            assert [len(seg), hashlib.sha256(seg).hexdigest()[:16]] == row[col: col + 2], (level, k)
        # trailer = Adler-32 of the whole input, recomputed on the device by a different code path
        trailer = int.from_bytes(dst[res.out_bytes - 4: res.out_bytes].cpu().numpy().tobytes(), "big")
        assert trailer == res.adler32 == eng.adler32_device(src.data_ptr(), nbytes)
        # inflate on the device: identical bytes
        back = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        r = eng.inflate_device(dst.data_ptr(), res.out_bytes, offs.data_ptr(), nchunks, back.data_ptr(), nbytes)
        assert r.out_bytes == nbytes and r.adler32 == trailer
        assert torch.equal(src, back)
        del back
    eng.close()


def test_reference_inflate_accepts_the_device_stream():
    """Config 4: the stock inflate must accept the stream.  The device's level-6 stream of the first 256 MiB of the workload through the compiled
    reference's uncompress() (oracle/_ref/libzref.so, prebuilt in the development container) and through the interpreter's zlib."""
    import zlib
    import torch
    import zlib_amd
    from zlib_amd import gpu
    from oracle import refzlib as R
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "corpus_silesia.json")))
    nchunks = 4096
    nbytes = nchunks * 65536
    eng = zlib_amd.Engine(0)
    dev = torch.device("cuda", 0)
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    eng.corpus_fill_device(g["kind"], g["seed"], 0, nchunks, src.data_ptr())
    cap = eng.L.zgpu_deflate_bound(nbytes, 65536)
    dst = torch.empty(cap, dtype=torch.uint8, device=dev)
    res = eng.deflate_device(src.data_ptr(), nbytes, 6, dst.data_ptr(), cap, flags=gpu.F_FINAL | gpu.F_ZLIB_WRAP)
    z = dst[: res.out_bytes].cpu().numpy().tobytes()
    data = src.cpu().numpy().tobytes()
    row = [r for r in g["last_rows"] if r[0] == nchunks - 1][0]
    assert hashlib.sha256(z[-4 - row[4]:-4]).hexdigest()[:16] == row[5]  # the BFINAL chunk: the reference's bytes
    assert zlib.decompress(z) == data
    if R.available():
        rc, out = R.uncompress(z, nbytes)
        assert rc == 0 and out == data
    else:
        pytest.skip("oracle/_ref/libzref.so is not in this checkout: checked with the interpreter's zlib only")
    eng.close()


@pytest.mark.parametrize("rank", [0, 7])
def test_8gib_logtext_rank_share_against_reference_hashes(rank):
    """Config 5's per-GPU share: 131072 chunks (8 GiB) of the 64 GiB log-text, as rank `rank` of 8 compresses them (raw body, BFINAL
    only on the last rank), every sampled chunk of the range against the reference's hashes, Adler-32 and round trip on the device."""
    import torch
    import zlib_amd
    from zlib_amd import gpu, shard
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "corpus_logtext.json")))
    world = 8
    lo, hi = shard.chunk_range(g["total_chunks"], rank, world)
    nchunks = hi - lo
    assert nchunks == 131072
    nbytes = nchunks * 65536
    eng = zlib_amd.Engine(0)
    dev = torch.device("cuda", 0)
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    eng.corpus_fill_device(g["kind"], g["seed"], lo, nchunks, src.data_ptr())
    cap = eng.L.zgpu_deflate_bound(nbytes, 65536)
    dst = torch.empty(cap, dtype=torch.uint8, device=dev)
    offs = torch.empty(nchunks + 1, dtype=torch.int64, device=dev)
    res = eng.deflate_device(src.data_ptr(), nbytes, 6, dst.data_ptr(), cap, flags=gpu.F_FINAL if rank == world - 1 else 0, d_offsets=offs.data_ptr())
    assert res.nchunks == nchunks
    o = offs.cpu().numpy()
    assert o[0] == 0 and o[-1] == res.out_bytes
    rows = [r for r in g["rows"] if lo <= r[0] < hi]
    assert len(rows) >= 32
    if rank == world - 1:  # the chunk that carries BFINAL: `last_rows` hold the reference's output for it
        last = [r for r in g["last_rows"] if r[0] == hi - 1]
        assert len(last) == 1
        rows = [r for r in rows if r[0] != hi - 1] + last
    for row in rows:
        k = row[0] - lo
        seg = dst[int(o[k]): int(o[k + 1])].cpu().numpy().tobytes()
        assert [len(seg), hashlib.sha256(seg).hexdigest()[:16]] == row[4:6], (rank, row[0])
    assert res.adler32 == eng.adler32_device(src.data_ptr(), nbytes)
    if rank == world - 1:  # a complete raw stream of its own: decodes back on the device
        back = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        r = eng.inflate_device(dst.data_ptr(), res.out_bytes, offs.data_ptr(), nchunks, back.data_ptr(), nbytes)
        assert r.out_bytes == nbytes and r.adler32 == res.adler32
        assert torch.equal(src, back)
    eng.close()


def test_rccl_gather_world1_on_device_tensors():
    """The N > 1 framing path of bench.py with the nccl (= RCCL) backend, world size 1: communicator set-up, the all_gather of
    (size, Adler-32, length) and the framing run on device tensors; the stream must equal the one-call stream of the engine."""
    import torch
    import torch.distributed as dist
    import zlib_amd
    from zlib_amd import gpu, shard
    from oracle import corpus_py as CP
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29541", world_size=1, rank=0, device_id=dev)
    try:
        eng = zlib_amd.Engine(0)
        n = 96
        data = CP.chunks(CP.KIND_LOGTEXT, 5, n)
        src = torch.from_numpy(data.copy()).to(dev)
        cap = eng.L.zgpu_deflate_bound(src.numel(), 65536)
        body = torch.empty(cap, dtype=torch.uint8, device=dev)
        res = eng.deflate_device(src.data_ptr(), src.numel(), 6, body.data_ptr(), cap, flags=gpu.F_FINAL)
        stream, total = shard.gather_stream(body[: res.out_bytes], res.adler32, src.numel(), 6)
        whole = torch.empty(cap, dtype=torch.uint8, device=dev)
        res2 = eng.deflate_device(src.data_ptr(), src.numel(), 6, whole.data_ptr(), cap, flags=gpu.F_FINAL | gpu.F_ZLIB_WRAP)
        assert total == res2.out_bytes and torch.equal(stream, whole[: res2.out_bytes])
        eng.close()
    finally:
        dist.destroy_process_group()


def test_rccl_gather_world1_through_the_c_abi():
    """The same exchange behind the C ABI (include/zamd_gpu.h zgpu_comm_*, zgpu_deflate_gather: RCCL opened by the library itself, no torch.distributed):
    world size 1 -- id, communicator, the all-gather of the sizes, exact sizing, framing on the device -- must give the engine's one-call stream."""
    import torch
    import zlib_amd
    from zlib_amd import gpu
    from oracle import corpus_py as CP
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    eng = zlib_amd.Engine(0)
    comm = gpu.Comm(0, 1, 0, lambda b: b)
    try:
        data = CP.chunks(CP.KIND_SILESIA, 9, 80)
        src = torch.from_numpy(data.copy()).to(dev)
        cap = eng.L.zgpu_deflate_bound(src.numel(), 65536)
        body = torch.empty(cap, dtype=torch.uint8, device=dev)
        whole = torch.empty(cap, dtype=torch.uint8, device=dev)
        for level in (1, 6):
            res = eng.deflate_device(src.data_ptr(), src.numel(), level, body.data_ptr(), cap, flags=gpu.F_FINAL)
            table, total = comm.sizes(res.out_bytes, res.adler32, src.numel())
            assert [int(x) for x in table] == [res.out_bytes, res.adler32, src.numel()] and total == res.out_bytes + 6
            out = torch.zeros(total, dtype=torch.uint8, device=dev)  # exactly as large as the stream
            adler = comm.gather(body.data_ptr(), table, level, out.data_ptr(), total)
            res2 = eng.deflate_device(src.data_ptr(), src.numel(), level, whole.data_ptr(), cap, flags=gpu.F_FINAL | gpu.F_ZLIB_WRAP)
            assert total == res2.out_bytes and adler == res2.adler32 and torch.equal(out, whole[: total])
            with pytest.raises(zlib_amd.EngineError):  # a buffer one byte short is refused, nothing is received into it
                comm.gather(body.data_ptr(), table, level, out.data_ptr(), total - 1)
    finally:
        comm.close()
        eng.close()
