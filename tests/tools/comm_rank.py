"""One rank of tests/test_gpu_comm_world.py: compresses its chunk range of the shared input on GPU 0 and takes part in the C library's gather.
usage: comm_rank.py WORLD RANK LEVEL INPUT_FILE ID_FILE OUT_FILE"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    world, rank, level = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    in_file, id_file, out_file = sys.argv[4:7]
    import torch
    import zlib_amd
    from zlib_amd import gpu, shard
    data = np.fromfile(in_file, dtype=np.uint8)
    nchunks = (data.size + 65535) // 65536
    c0, c1 = shard.chunk_range(nchunks, rank, world)
    mine = data[c0 * 65536: min(c1 * 65536, data.size)]
    eng = zlib_amd.Engine(0)

    def exchange(b):
        if rank == 0:
            with open(id_file + ".tmp", "wb") as f:
                f.write(b)
            os.replace(id_file + ".tmp", id_file)
            return b
        for _ in range(600):
            if os.path.exists(id_file):
                return open(id_file, "rb").read()
            time.sleep(0.1)
        raise RuntimeError("no id")

    comm = gpu.Comm(0, world, rank, exchange)
    dev = torch.device("cuda", 0)
    src = torch.from_numpy(mine.copy()).to(dev) if mine.size else torch.empty(1, dtype=torch.uint8, device=dev)
    cap = eng.L.zgpu_deflate_bound(int(mine.size), 65536)
    dst = torch.empty(cap, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    if mine.size or rank == world - 1:  # (a rank without chunks has no body; the last rank always writes the final block)
        res = eng.deflate_device(src.data_ptr(), int(mine.size), level, dst.data_ptr(), cap, flags=gpu.F_FINAL if rank == world - 1 else 0, stream=stream)
        body, adler = res.out_bytes, res.adler32
    else:
        body, adler = 0, 1
    for rep in range(2):  # twice: the communicator is reused call after call
        table, total = comm.sizes(body, adler, int(mine.size), stream=stream)
        out = torch.full((total + 64,), 0xA5, dtype=torch.uint8, device=dev) if rank == 0 else None
        a = comm.gather(dst.data_ptr(), table, level, out.data_ptr() if rank == 0 else None, total if rank == 0 else 0, stream=stream)
    if rank == 0:
        torch.cuda.synchronize()
        h = out.cpu().numpy()
        assert (h[total:] == 0xA5).all(), "bytes behind the stream were written"
        with open(out_file, "wb") as f:
            f.write(h[:total].tobytes())
        with open(out_file + ".adler", "w") as f:
            f.write("%d" % a)
    comm.close()
    eng.close()


if __name__ == "__main__":
    main()
