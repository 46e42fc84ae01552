import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, zlib_amd
e = zlib_amd.Engine(0)
n = 256
src = torch.empty(n * 65536, dtype=torch.uint8, device="cuda")
e.corpus_fill_device(0, 0x5EED5117, 100, n, src.data_ptr())
data = src.cpu().numpy().tobytes()
for level in (1, 6):
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    raw = co.compress(data) + co.flush()
    print("level", level, len(raw), flush=True)
    try:
        out = e.inflate_stream_host(raw, len(data))
        print("equal", out == data, e.spec_counts(), flush=True)
        if out != data:
            import numpy as np
            a = np.frombuffer(out, np.uint8); b = np.frombuffer(data, np.uint8)
            m = min(len(a), len(b)); d = np.nonzero(a[:m] != b[:m])[0]
            print("lens", len(a), len(b), "ndiff", len(d), d[:20])
    except Exception as ex:
        print("ERR", ex, flush=True)
