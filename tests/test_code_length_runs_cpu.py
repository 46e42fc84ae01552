"""The header's code-length sequences, run by run (zgpu_huffman.hip: run_items / walk_lengths_wave) against the reference's state machine
(scan_tree / send_tree, /root/reference/qcsrc/trees.c:707-797), both restated here: what the machine emits depends on nothing but the run of equal
lengths it stands in -- that is what lets a wave give every run a lane.  Exhaustive over run lengths 1..320 for every length value, every pair of
neighbouring runs, and 20 000 random sequences shaped like real trees (long zero runs, short runs of neighbouring lengths)."""
import random

REP_3_6, REPZ_3_10, REPZ_11_138 = 16, 17, 18


def machine(lens):
    """send_tree's item sequence for the code lengths lens[0..max_code] (trees.c:752-797): (symbol, extra value, extra bits)"""
    out = []
    prevlen, nextlen, count = -1, lens[0], 0
    max_count, min_count = (138, 3) if nextlen == 0 else (7, 4)
    for n in range(len(lens)):
        curlen = nextlen
        nextlen = lens[n + 1] if n + 1 < len(lens) else 0xFFFF  # the guard behind the last code (trees.c:719,764)
        count += 1
        if count < max_count and curlen == nextlen:
            continue
        if count < min_count:
            out += [(curlen, 0, 0)] * count
        elif curlen != 0:
            if curlen != prevlen:
                out.append((curlen, 0, 0))
                count -= 1
            out.append((REP_3_6, count - 3, 2))
        elif count <= 10:
            out.append((REPZ_3_10, count - 3, 3))
        else:
            out.append((REPZ_11_138, count - 11, 7))
        count, prevlen = 0, curlen
        if nextlen == 0:
            max_count, min_count = 138, 3
        elif curlen == nextlen:
            max_count, min_count = 6, 3
        else:
            max_count, min_count = 7, 4
    return out


def run_items(L, N):
    """the items of one maximal run of N codes of length L, as zgpu_huffman.hip's run_items plays them"""
    out, rem, first = [], N, True
    while rem:
        if L == 0:
            take = min(rem, 138)
            if take < 3:
                out += [(0, 0, 0)] * take
            elif take <= 10:
                out.append((REPZ_3_10, take - 3, 3))
            else:
                out.append((REPZ_11_138, take - 11, 7))
        else:
            maxc, minc = (7, 4) if first else (6, 3)
            take = min(rem, maxc)
            if take < minc:
                out += [(L, 0, 0)] * take
            elif first:
                out += [(L, 0, 0), (REP_3_6, take - 4, 2)]
            else:
                out.append((REP_3_6, take - 3, 2))
            first = False
        rem -= take
    return out


def by_runs(lens):
    out, i = [], 0
    while i < len(lens):
        j = i
        while j < len(lens) and lens[j] == lens[i]:
            j += 1
        out += run_items(lens[i], j - i)
        i = j
    return out


def test_single_runs_of_every_length_and_size():
    for L in range(0, 16):
        for N in range(1, 321):
            assert by_runs([L] * N) == machine([L] * N), (L, N)


def test_pairs_of_neighbouring_runs():
    for a in (0, 1, 7, 15):
        for b in (0, 2, 7, 8):
            if a == b:
                continue
            for na in list(range(1, 15)) + [137, 138, 139, 150]:
                for nb in list(range(1, 15)) + [138, 140]:
                    seq = [a] * na + [b] * nb
                    assert by_runs(seq) == machine(seq), (a, na, b, nb)


def test_random_sequences_shaped_like_trees():
    rnd = random.Random(707797)
    for _ in range(20000):
        seq = []
        target = rnd.choice((1, 2, 19, 30, 257, 286, 316))
        while len(seq) < target:
            kind = rnd.random()
            L = 0 if kind < 0.35 else rnd.choice((rnd.randint(1, 15), rnd.randint(6, 10)))
            n = rnd.choice((1, 1, 1, 2, 2, 3, 4, 5, 6, 7, 8, 9, 12, 13, 20, rnd.randint(1, 150)))
            seq += [L] * n
        seq = seq[:target]
        assert by_runs(seq) == machine(seq)
