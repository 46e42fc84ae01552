"""Deterministic small/edge-case inputs shared by oracle/gen_golden.py and the tests.  TEST INFRASTRUCTURE.

Everything here is integer arithmetic on a 64-bit LCG so the inputs are reproducible anywhere."""
CHUNK = 65536


class Lcg:
    def __init__(self, seed):
        self.s = (seed * 0x9E3779B97F4A7C15 + 1) & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        return self.s >> 33

    def below(self, n):
        return self.next() % n


def make(kind: str, n: int, seed: int = 1) -> bytes:
    g = Lcg(seed * 1000003 + n)
    out = bytearray()
    if kind == "zeros":
        return bytes(n)
    if kind == "rand":
        while len(out) < n:
            out += (g.next() & 0xFFFFFFFF).to_bytes(4, "little")
    elif kind == "ab":
        while len(out) < n:
            v = g.next()
            out += bytes(97 + ((v >> i) & 1) for i in range(24))
    elif kind == "runs":
        while len(out) < n:
            out += bytes([g.below(4) * 60 + 32]) * (1 + g.below(600))
    elif kind == "text":
        words = []
        for i in range(400):
            ln = 2 + g.below(8)
            words.append(bytes(97 + g.below(26) for _ in range(ln)))
        while len(out) < n:
            r = min(g.below(400), g.below(400), g.below(400))
            out += words[r] + (b"\n" if g.below(12) == 0 else b" ")
    elif kind == "period":  # long matches at a fixed large distance (exercises MAX_DIST / TOO_FAR edges)
        period = [3, 4097, 32506, 32507, 32768, 258, 4096][seed % 7]
        base = bytearray()
        while len(base) < min(period, n):
            base += (g.next() & 0xFFFFFFFF).to_bytes(4, "little")
        base = base[:period]
        while len(out) < n:
            out += base
    elif kind == "mix":
        kinds = ["rand", "text", "runs", "ab", "zeros"]
        while len(out) < n:
            out += make(kinds[g.below(5)], 1 + g.below(6000), seed + len(out) + 1)
    else:
        raise ValueError(kind)
    return bytes(out[:n])


SMALL_SIZES = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 15, 16, 17, 31, 32, 33, 100, 255, 256, 257, 258, 259, 260, 261, 262,
               263, 264, 265, 300, 511, 512, 1000, 4095, 4096, 4097]
BIG_SIZES = [20000, 32767, 32768, 32769, 40000, 65273, 65274, 65275, 65276, 65277, 65278, 65279, 65280, 65400,
             65533, 65534, 65535, 65536]
KINDS = ["text", "rand", "runs", "ab", "zeros", "period", "mix"]


def small_cases():
    """(name, bytes) pairs small enough to store complete expected outputs."""
    for kind in KINDS:
        for n in SMALL_SIZES:
            yield "%s-%d" % (kind, n), make(kind, n)


def big_cases():
    """(name, bytes) pairs near/at the chunk size; expected outputs stored as hashes."""
    for kind in KINDS:
        for i, n in enumerate(BIG_SIZES):
            yield "%s-%d" % (kind, n), make(kind, n, seed=i + 2)


HELLO = b"hello, hello!\x00"


def hello_1mib() -> bytes:
    return (b"hello, hello! " * (CHUNK * 16 // 14 + 1))[: CHUNK * 16]


def nomatch(n: int) -> bytes:
    """n <= 32768 bytes in which no three-byte string occurs twice (pairs of a 7-bit digit and a 7-bit digit with the top bit set): every token is a
    literal, so the token count is the byte count -- the input for "the buffer fills exactly at ..." cases (tests/golden/fullblock_kat.json)."""
    b = bytearray()
    i = 0
    while len(b) < n:
        b += bytes([i & 127, 0x80 | ((i >> 7) & 127)])
        i += 1
    return bytes(b[:n])
