import sys, os, ctypes as C
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import zhost as Z
from oracle import cases, corpus_py as CP, oracle_py as O
L = Z.lib()
text = cases.make("text", 5000, 4)
data = CP.chunks(0, 50, 3).tobytes()[:-321]
for dictionary, level in ((text[:2000], 6), (b"hello\0", 1), (text[:300], 0)):
    d = dictionary[-32506:]; room = 65536 - len(d)
    hdr = bytearray(O.deflate_stream(b"", level)[:2]); hdr[1] = (hdr[1] & 0xC0) | 0x20; hdr[1] += 31 - ((hdr[0] << 8) + hdr[1]) % 31
    first = O.deflate_chunk(data[:room], 0, False) if level == 0 else O.deflate_chunk_dict(d, data[:room], level, False)
    rest = data[room:]; nrest = (len(rest) + 65535) // 65536
    want = bytes(hdr) + O.adler32(dictionary).to_bytes(4, "big") + first + b"".join(O.deflate_chunk(rest[k * 65536:(k + 1) * 65536], level, k == nrest - 1) for k in range(nrest)) + O.adler32(data).to_bytes(4, "big")
    s = Z.ZStream(); L.inflateInit_(C.byref(s), b"1.2.3", C.sizeof(Z.ZStream))
    src = C.create_string_buffer(want, len(want)); cap = len(data) + 16; out = C.create_string_buffer(cap)
    s.next_in = C.addressof(src); s.avail_in = len(want); s.next_out = C.addressof(out); s.avail_out = cap
    r1 = L.inflate(C.byref(s), Z.Z_NO_FLUSH)
    r2 = L.inflateSetDictionary(C.byref(s), dictionary, len(dictionary))
    r3 = L.inflate(C.byref(s), Z.Z_FINISH)
    print("level", level, "rc", r1, r2, r3, "msg", s.msg, "total_out", s.total_out, "of", len(data), "total_in", s.total_in, "of", len(want), "avail_in", s.avail_in, "equal prefix", out.raw[:s.total_out] == data[:s.total_out])
    r4 = L.inflate(C.byref(s), Z.Z_FINISH)
    print("   again:", r4, s.total_out, s.msg)
    L.inflateEnd(C.byref(s))
