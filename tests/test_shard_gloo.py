"""The multi-GPU stream assembly (zlib_amd/shard.py) on CPU: world_size 2 and 3 over gloo, per-rank compression done by
the CPU oracle.  The gathered stream must be byte-identical to the single-process mode-B stream of the whole input."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cases, oracle_py as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, data, level, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from zlib_amd import shard
    nchunks = max(1, (len(data) + 65535) // 65536)
    lo, hi = shard.chunk_range(nchunks, rank, world)
    mine = data[lo * 65536: hi * 65536]
    segs = [O.deflate_chunk(mine[k * 65536:(k + 1) * 65536], level, (lo + k == nchunks - 1)) for k in range(hi - lo)]
    body = torch.frombuffer(bytearray(b"".join(segs)), dtype=torch.uint8) if segs else torch.empty(0, dtype=torch.uint8)
    stream, total = shard.gather_stream(body, O.adler32(mine), len(mine), level)
    if rank == 0:
        q.put(bytes(stream.numpy().tobytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nbytes", [(2, 65536 * 5 + 1234), (3, 65536 * 7), (2, 65536 + 1)])
def test_gathered_stream_equals_single_process_stream(world, nbytes):
    data = cases.make("mix", nbytes, 17)
    level = 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, data, level, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == O.deflate_stream(data, level)
    rc, out, used, msg = O.inflate_zlib(got, len(data))
    assert rc == 1 and out == data


def test_chunk_ranges_cover_everything():
    from zlib_amd import shard
    for n in (1, 2, 7, 8, 65536, 1048576):
        for w in (1, 2, 3, 4, 8):
            r = [shard.chunk_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))


def test_adler_join_matches_oracle():
    from zlib_amd import shard
    a, b = cases.make("rand", 70001, 1), cases.make("text", 12345, 2)
    assert shard.adler_join(O.adler32(a), O.adler32(b), len(b)) == O.adler32(a + b)
    assert shard.adler_join(1, O.adler32(b), len(b)) == O.adler32(b)


def test_inflate_partition_decodes_disjoint_ranges_of_one_stream():
    """SURVEY.md 8e, last row: every rank decodes a contiguous chunk range of ONE stream from the compressor's chunk table; the ranges'
    outputs, side by side, are the input (decoded here by the CPU oracle, one rank after the other: there is nothing to exchange)."""
    from zlib_amd import shard
    data = cases.make("mix", 65536 * 9 + 4321, 23)
    nchunks = (len(data) + 65535) // 65536
    segs = [O.deflate_chunk(data[k * 65536:(k + 1) * 65536], 6, k == nchunks - 1) for k in range(nchunks)]
    body = b"".join(segs)
    offsets = [0]
    for s in segs:
        offsets.append(offsets[-1] + len(s))
    for world in (1, 2, 3, 4, 16):
        out = bytearray(len(data))
        covered = 0
        for rank in range(world):
            lo, hi, c0, c1, o0, table = shard.inflate_partition(offsets, 65536, rank, world)
            assert table[0] == 0 and table[-1] == c1 - c0 and len(table) == hi - lo + 1
            piece = body[c0:c1]
            for k in range(hi - lo):
                rc, dec, used, msg = O.inflate_raw(piece[table[k]:table[k + 1]], 65536)
                assert rc == 1 if lo + k == nchunks - 1 else rc in (0, -5), (rc, msg)  # (a chunk that is not the last ends with a flush marker: the decoder wants more)
                out[o0 + k * 65536: o0 + k * 65536 + len(dec)] = dec
                covered += len(dec)
        assert covered == len(data) and bytes(out) == data, world


def test_gather_layout_is_the_c_librarys_arithmetic():
    """Where every rank's body lands in the gathered stream: zgpu_gather_layout (the arithmetic zgpu_deflate_gather uses on the GPUs) against the
    obvious prefix sums, for empty bodies too."""
    import random
    from zlib_amd import gpu
    rnd = random.Random(5)
    for world in range(1, 9):
        for _ in range(50):
            table = [[rnd.choice([0, 1, rnd.randrange(1 << 34)]), rnd.randrange(1 << 32), rnd.randrange(1 << 36)] for _ in range(world)]
            offs, total = gpu.gather_layout(table)
            want = [2]
            for r in range(world):
                want.append(want[-1] + table[r][0])
            assert offs == want and total == want[-1] + 4
