"""GPU box: length (and SHA-256) of the continuous stream of the first N MiB of the silesia-mix workload at a level, next to the fixture's (tests/golden/continuous_kat.json).
usage: cont_len.py [mib] [level]"""
import sys, os, json, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zlib_amd
from zlib_amd import gpu
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
eng = zlib_amd.Engine(0)
n = mib << 20
src = torch.empty(n, dtype=torch.uint8, device="cuda")
eng.corpus_fill_device(0, 0x5EED5117, 0, n // 65536, src.data_ptr())
cap = eng.L.zgpu_deflate_cont_bound(n) + 64
dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
res = eng.deflate_device(src.data_ptr(), n, level, dst.data_ptr(), cap, flags=gpu.F_FINAL | gpu.F_ZLIB_WRAP | gpu.F_CONTINUOUS)
z = dst[: res.out_bytes].cpu().numpy().tobytes()
rows = [r for r in json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "continuous_kat.json")))["rows"]
        if r["corpus"] == 0 and r["level"] == level and r["sync_at"] is None and r["n"] == n]
print("%d MiB level %d: %d bytes %s; fixture %s; env SORT=%s BATCH=%s" % (mib, level, len(z), hashlib.sha256(z).hexdigest()[:16],
      [(r["len"], r["sha256"][:16]) for r in rows], os.environ.get("ZGPU_SORT"), os.environ.get("ZGPU_CONT_BATCH_TILES")))
