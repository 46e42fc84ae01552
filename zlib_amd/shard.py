"""Multi-GPU assembly of ONE RFC 1950 stream from per-rank chunk ranges (one process per GPU, torch.distributed).

The reference has no distributed path; this is the only exchange step the chunked engine needs:
chunks are pure functions of their bytes, so rank r compresses the contiguous chunk range [r*M/G, (r+1)*M/G) into a
raw-deflate body of its own (only the last rank's last chunk carries BFINAL), and the bodies are gathered to rank 0:

  1. all_gather of (body bytes, Adler-32, input bytes) -- 3 x int64 per rank;
  2. gather-v of the bodies: rank 0 posts one irecv per peer at the prefix-summed offset, every peer sends once
     (RCCL has no gatherv; on xGMI each peer has its own link to GPU 0, so the transfers run side by side);
  3. rank 0 writes the 2-byte zlib header in front and the big-endian Adler-32 of the whole input behind, combining
     the per-rank checksums (Adler-32 of a concatenation: a = a1 + a2 - 1, b = b1 + b2 + len2 * (a1 - 1), mod 65521).

Works with any backend (nccl on GPUs, gloo on CPU tensors -- the latter is what tests/test_shard_gloo.py runs).  The same exchange without
torch, behind the C ABI: include/zamd_gpu.h zgpu_comm_* / zgpu_deflate_gather (zlib_amd/csrc/zgpu_comm.hip), which bench.py --gpus N uses; this
module stays as its torch.distributed twin (gloo on the CPU) and shares the offset arithmetic with it (zgpu_gather_layout).
"""
import torch
import torch.distributed as dist

ADLER_BASE = 65521


def chunk_range(nchunks: int, rank: int, world: int):
    """contiguous, balanced chunk range of a rank"""
    lo = nchunks * rank // world
    hi = nchunks * (rank + 1) // world
    return lo, hi


def inflate_partition(offsets, chunk_size: int, rank: int, world: int):
    """Multi-GPU inflate (last row): the compressor's chunk table cuts the stream wherever a chunk ends, so rank r decodes the
    contiguous chunk range chunk_range(M, r, world) -- compressed bytes [offsets[lo], offsets[hi]) -- into the output range
    [lo * chunk_size, hi * chunk_size) (the last chunk may be short): disjoint ranges, no collective.  Returns
    (lo, hi, first compressed byte, behind the last compressed byte, first output byte, the rank's table re-based to 0)."""
    m = len(offsets) - 1
    lo, hi = chunk_range(m, rank, world)
    base = int(offsets[lo])
    return lo, hi, base, int(offsets[hi]), lo * chunk_size, [int(o) - base for o in offsets[lo:hi + 1]]


def adler_join(x: int, y: int, len_y: int) -> int:
    ax, bx, ay, by = x & 0xFFFF, x >> 16, y & 0xFFFF, y >> 16
    a = (ax + ay + ADLER_BASE - 1) % ADLER_BASE
    b = (bx + by + (len_y % ADLER_BASE) * ((ax + ADLER_BASE - 1) % ADLER_BASE)) % ADLER_BASE
    return a | (b << 16)


def zlib_header(level: int) -> bytes:
    """/root/reference/qcsrc/deflate.c:625-641 for windowBits 15, no dictionary"""
    hdr = (8 + (7 << 4)) << 8
    hdr |= (0 if level < 2 else 1 if level < 6 else 2 if level == 6 else 3) << 6
    hdr += 31 - hdr % 31
    return bytes([hdr >> 8, hdr & 0xFF])


def gather_stream(body: torch.Tensor, adler: int, in_bytes: int, level: int, out: torch.Tensor = None, group=None):
    """body: 1-D uint8 tensor with this rank's raw-deflate bytes (device = the backend's device).  Returns on rank 0
    (stream tensor view, total bytes), on other ranks (None, total bytes).  `out` (rank 0, optional) is a reusable
    destination buffer of sufficient size."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = body.device
    mine = torch.tensor([body.numel(), adler, in_bytes], dtype=torch.int64, device=dev)
    allv = torch.empty(world * 3, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allv, mine, group=group)
    rows = allv.view(world, 3).tolist()
    sizes = [r[0] for r in rows]
    from . import gpu
    offsets, total = gpu.gather_layout(rows)  # (the C library's arithmetic: zgpu_gather_layout, which the RCCL gather of zgpu_comm.hip uses too)
    assert offsets[0] == 2 and total == 2 + sum(sizes) + 4
    if rank != 0:
        if body.numel():
            for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, body, 0, group)]):
                q.wait()
        return None, total
    if out is None or out.numel() < total:
        out = torch.empty(total, dtype=torch.uint8, device=dev)
    out[0:2] = torch.tensor(list(zlib_header(level)), dtype=torch.uint8, device=dev)
    off = 2
    out[off:off + sizes[0]].copy_(body)
    off += sizes[0]
    # one coalesced group of receives (ncclGroupStart/End under RCCL): posted one by one they would run one after the
    # other on the communicator's stream, and the peers' links would take turns instead of running side by side
    ops = []
    for r in range(1, world):
        if sizes[r]:
            ops.append(dist.P2POp(dist.irecv, out[off:off + sizes[r]], r, group))
        off += sizes[r]
    for q in (dist.batch_isend_irecv(ops) if ops else []):
        q.wait()
    a = 1
    for r in range(world):
        a = adler_join(a, rows[r][1], rows[r][2])
    out[off:off + 4] = torch.tensor(list(a.to_bytes(4, "big")), dtype=torch.uint8, device=dev)
    return out[:total], total
