// zgpu_huffman.hip -- block construction stage: per-block histogram, zlib-exact Huffman trees, block-type
// choice and bit emission, one 256-lane workgroup per chunk.
//
// Restates (file:line under /root/reference):
//   init_block / tally frequencies       qcsrc/trees.c:411-424, h/deflate.h:308-324
//   build_tree, pqdownheap, smaller      qcsrc/trees.c:434-478, 619-701   (exact heap order and depth tie-break)
//   gen_bitlen with overflow repair      qcsrc/trees.c:490-567
//   gen_codes / bi_reverse               qcsrc/trees.c:577-609, 1146-1156
//   scan_tree, build_bl_tree, send_tree, send_all_trees   qcsrc/trees.c:707-862
//   _tr_flush_block (stored / static / dynamic choice)    qcsrc/trees.c:921-1016
//   compress_block                       qcsrc/trees.c:1072-1118
//   _tr_stored_block, copy_block, bi_windup               qcsrc/trees.c:867-879, 1178-1219
//   full-flush marker                    qcsrc/deflate.c:811-812
//
// Work split inside the workgroup: all lanes histogram the block's tokens (LDS atomics); lane 0 builds the
// literal/length tree while lane 64 builds the distance tree (independent heaps in LDS), lane 0 then builds the
// bit-length tree and writes the block header; all lanes then emit: every lane owns a contiguous token range,
// sizes it, a workgroup prefix scan turns sizes into bit offsets, and each lane packs its range into 32-bit
// words (plain stores for interior words, atomic OR for the two words it may share with a neighbour).
#include "zgpu_common.h"

namespace zgpu {

#ifndef ZGPU_HUFF_THREADS
#define ZGPU_HUFF_THREADS 128
#endif
constexpr int kThreads = ZGPU_HUFF_THREADS;
constexpr uint32_t kEmitPer = 4, kEmitRound = kThreads * kEmitPer; // token emission: rounds of kEmitRound tokens, kEmitPer consecutive ones per lane

// Heap entries are packed: frequency << 16 | depth << 10 | node.  The reference orders nodes by (frequency, depth) with
// "<=" deciding ties (smaller(), trees.c:451-453), which on packed entries is one comparison of entry >> 10 -- and a
// sift-down step needs one 8-byte LDS read (both children) instead of two dependent rounds of 2-byte reads.
// Ranges: a block holds at most 16384 symbols, so frequencies fit 15 bits; the depth of a Huffman tree over that total
// weight is at most 21 (Fibonacci bound), 6 bits; nodes are numbered below 2*286+1, 10 bits.
template <int kNodes> // 2 * symbols + 1 (HEAP_SIZE of the tree in question)
struct TreeWorkT {
    static constexpr int kCap = kNodes;
    uint16_t freq[kNodes];
    uint16_t dad[kNodes];
    uint16_t len[kNodes];
    __attribute__((aligned(16))) uint32_t heap[kNodes + 5]; // [1..heap_len]: packed entries; [heap_max..]: node numbers in extraction order (+4: the grandchildren read of sift_down may start at 2 * heap_len)
    uint16_t bl_count[kMaxBits + 1];
    uint16_t next_code[kMaxBits + 1];
};
using TreeWork = TreeWorkT<kHeapSize>;          // literal/length tree, and the bit-length tree after it
using TreeWorkD = TreeWorkT<2 * kDCodes + 1>;    // distance tree: a tenth of the LDS, so more chunks are resident per CU
__device__ inline uint32_t heap_entry(uint32_t freq, uint32_t depth, uint32_t node) { return (freq << 16) | (depth << 10) | node; }

// LSB-first bit writer.  The buffer is written once, a whole word at a time, and never read: the bits of the word a writer stops in travel to the
// next writer as its `carry` (the workgroup's writers take turns: block header, token rounds, stored bytes, the chunk's end), so the slots need no
// zero-initialisation and no atomic OR.
struct BitWriter {
    uint32_t *words;
    uint32_t wi;    // index of the word being assembled
    uint64_t acc;   // bits not yet stored (low nacc bits valid, the rest zero)
    uint32_t nacc;
    __device__ void begin(uint32_t *base, uint64_t bitpos, uint32_t carry) { words = base; wi = (uint32_t)(bitpos >> 5); nacc = (uint32_t)(bitpos & 31); acc = carry; }
    __device__ uint64_t pos() const { return ((uint64_t)wi << 5) + nacc; }
    __device__ void put(uint32_t v, uint32_t nb)
    {
        acc |= (uint64_t)v << nacc; nacc += nb;
        if (nacc >= 32) { words[wi] = (uint32_t)acc; wi++; acc >>= 32; nacc -= 32; }
    }
    __device__ void put64(uint64_t v, uint32_t nb)
    {
        if (nb > 32) { put((uint32_t)v, 32); put((uint32_t)(v >> 32), nb - 32); } else put((uint32_t)v, nb);
    }
    __device__ void align_byte() { uint32_t k = (8 - (nacc & 7)) & 7; if (k) put(0, k); }
    __device__ uint32_t carry() const { return (uint32_t)acc; } // the bits of the unfinished word, for the writer that goes on at pos()
    __device__ void store_tail() { if (nacc) words[wi] = (uint32_t)acc; } // the last writer of a slot
};

template <class TW>
__device__ inline void sift_down(TW &t, int heap_len, int k) // pqdownheap, trees.c:461-478
{
    // Two levels per trip to LDS: the four grandchildren (16 bytes at 2 j) are read together with the two children, so the second of two steps decides from registers.
    // The comparisons and their order are pqdownheap's; what is read past the heap's end is never looked at (the j <= heap_len / j < heap_len tests stand).
    const uint32_t v = t.heap[k];
    int j = k << 1;
    while (j <= heap_len) {
        const uint2 ch = *reinterpret_cast<const uint2 *>(&t.heap[j]); // children j and j+1 (j is even); j+1 may lie past the heap
        const uint4 gc = *reinterpret_cast<const uint4 *>(&t.heap[2 * j]); // their children 2j .. 2j+3
        uint32_t c = ch.x; int j1 = j;
        if (j < heap_len && (ch.y >> 10) <= (ch.x >> 10)) { c = ch.y; j1 = j + 1; }
        if ((v >> 10) <= (c >> 10)) break;
        t.heap[k] = c; k = j1;
        const int j2 = j1 << 1;
        if (j2 > heap_len) break;
        const uint32_t g0 = j1 == j ? gc.x : gc.z, g1 = j1 == j ? gc.y : gc.w;
        uint32_t c2 = g0; int j3 = j2;
        if (j2 < heap_len && (g1 >> 10) <= (g0 >> 10)) { c2 = g1; j3 = j2 + 1; }
        if ((v >> 10) <= (c2 >> 10)) break;
        t.heap[k] = c2; k = j3; j = j3 << 1;
    }
    t.heap[k] = v;
}

// build_tree + gen_bitlen + gen_codes, executed by one lane.  t.freq[0..elems) holds the symbol counts.
// slen: static code lengths (constant memory) or nullptr; xbits/xbase: extra-bit table and first symbol using it.
// Results: out_len / out_code for symbols 0..elems-1; returns max_code.  opt_len / static_len accumulate mod 2^32
// exactly like the reference's unsigned long arithmetic does mod 2^64 (the transient "-1" of the forced codes).
// extra bits and static code length of symbol n of tree `kind` (0 literal/length, 1 distance, 2 bit lengths), in closed form
// (extra_lbits / extra_dbits / extra_blbits, trees.c:61-68; static_ltree lengths, trees.c:251-262; static distances are 5 bits):
// the serial lane would otherwise wait for a constant-memory load per symbol
__device__ inline uint32_t extra_bits_of(int kind, int n)
{
    if (kind == 0) { const int k = n - 257; return (k < 8 || k == 28) ? 0u : (uint32_t)((k >> 2) - 1); }
    if (kind == 1) return n < 4 ? 0u : (uint32_t)((n >> 1) - 1);
    return n == 16 ? 2u : n == 17 ? 3u : n == 18 ? 7u : 0u;
}
__device__ inline uint32_t static_len_of(int kind, int n) { return kind == 0 ? (n < 144 ? 8u : n < 256 ? 9u : n < 280 ? 7u : 8u) : 5u; }

template <class TW>
__device__ int build_tree(TW &t, int elems, int kind, int max_length,
                          uint16_t *out_code, uint8_t *out_len, uint32_t &opt_len, uint32_t &static_len)
{
    const bool has_static = kind != 2;
    int heap_len = 0, heap_max = TW::kCap, max_code = -1, n, m, node;
    for (n = 0; n < elems; n++) {
        if (t.freq[n] != 0) t.heap[++heap_len] = heap_entry(t.freq[n], 0, (uint32_t)(max_code = n));
        else t.len[n] = 0;
    }
    while (heap_len < 2) {
        node = max_code < 2 ? ++max_code : 0;
        t.heap[++heap_len] = heap_entry(1, 0, (uint32_t)node);
        t.freq[node] = 1; opt_len--;
        if (has_static) static_len -= static_len_of(kind, node);
    }
    for (n = heap_len / 2; n >= 1; n--) sift_down(t, heap_len, n);
    node = elems;
    do {
        const uint32_t en = t.heap[1];
        t.heap[1] = t.heap[heap_len--]; sift_down(t, heap_len, 1);
        const uint32_t em = t.heap[1];
        n = (int)(en & 1023u); m = (int)(em & 1023u);
        t.heap[--heap_max] = (uint32_t)n; t.heap[--heap_max] = (uint32_t)m;
        const uint32_t dn = (en >> 10) & 63u, dm = (em >> 10) & 63u;
        t.dad[n] = t.dad[m] = (uint16_t)node;
        t.heap[1] = heap_entry((en >> 16) + (em >> 16), (dn >= dm ? dn : dm) + 1, (uint32_t)node);
        node++;
        sift_down(t, heap_len, 1);
    } while (heap_len >= 2);
    t.heap[--heap_max] = t.heap[1] & 1023u;

    // gen_bitlen
    int h, bits, overflow = 0;
    for (bits = 0; bits <= kMaxBits; bits++) t.bl_count[bits] = 0;
    t.len[t.heap[heap_max]] = 0;
    for (h = heap_max + 1; h < TW::kCap; h++) {
        n = t.heap[h]; bits = t.len[t.dad[n]] + 1;
        if (bits > max_length) { bits = max_length; overflow++; }
        t.len[n] = (uint16_t)bits;
        if (n > max_code) continue;
        t.bl_count[bits]++;
        const uint32_t xb = (kind != 0 || n >= 257) ? extra_bits_of(kind, n) : 0u;
        opt_len += (uint32_t)t.freq[n] * ((uint32_t)bits + xb);
        if (has_static) static_len += (uint32_t)t.freq[n] * (static_len_of(kind, n) + xb);
    }
    if (overflow > 0) {
        do {
            bits = max_length - 1;
            while (t.bl_count[bits] == 0) bits--;
            t.bl_count[bits]--; t.bl_count[bits + 1] += 2; t.bl_count[max_length]--;
            overflow -= 2;
        } while (overflow > 0);
        for (bits = max_length; bits != 0; bits--) {
            n = t.bl_count[bits];
            while (n != 0) {
                m = t.heap[--h];
                if (m > max_code) continue;
                if (t.len[m] != (unsigned)bits) {
                    opt_len += ((uint32_t)bits - (uint32_t)t.len[m]) * (uint32_t)t.freq[m];
                    t.len[m] = (uint16_t)bits;
                }
                n--;
            }
        }
    }
    // gen_codes
    uint32_t c = 0; // (next_code in LDS: a private array indexed by a code length would live in scratch memory)
    for (bits = 1; bits <= kMaxBits; bits++) { c = (c + t.bl_count[bits - 1]) << 1; t.next_code[bits] = (uint16_t)c; }
    for (n = 0; n < elems; n++) {
        int l = (n <= max_code) ? t.len[n] : 0;
        out_len[n] = (uint8_t)l;
        out_code[n] = l ? (uint16_t)(__brev((uint32_t)t.next_code[l]++) >> (32 - l)) : 0;
    }
    return max_code;
}

// build_tree for a whole wave (round 2): the heap stays one lane's -- zlib's order of leaving it decides code lengths (see above) -- but what stands
// in front of it and behind it is every lane's: the heap is filled by ballot compaction, the depths of the finished tree come from pointer
// jumping (dad[] doubles as the ancestor at distance 2^k, len[] as the distance: six rounds reach depth 64, a tree over 16 383 tokens is
// no deeper than 21), counts per length and the cost sums are LDS atomics and a wave reduction, a symbol's code is the first code of its
// length plus its rank among the symbols of that length (ballots).  Same results as build_tree; every lane returns them.
__device__ inline uint32_t wave_sum(uint32_t v) { for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o); return v; }
template <class TW>
__device__ int build_tree_wave(TW &t, int elems, int kind, int max_length, uint16_t *out_code, uint8_t *out_len, uint32_t &opt_len, uint32_t &static_len, uint32_t lane)
{
    constexpr int kGroups = (TW::kCap / 2 + 63) / 64; // symbols: at most (kCap - 1) / 2
    constexpr int kNodeRounds = (TW::kCap + 63) / 64; // nodes per lane
    auto sync = [&]() { __builtin_amdgcn_wave_barrier(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
    int heap_len = 0, max_code = -1;
#pragma unroll
    for (int g = 0; g < kGroups; g++) {
        const int n = g * 64 + (int)lane;
        const uint32_t f = n < elems ? t.freq[n] : 0u;
        const uint64_t m = __ballot(f != 0);
        if (f) t.heap[heap_len + 1 + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = heap_entry(f, 0, (uint32_t)n);
        else if (n < elems) t.len[n] = 0;
        heap_len += (int)__builtin_popcountll(m);
        if (m) max_code = g * 64 + 63 - (int)__builtin_clzll(m);
    }
    sync();
    int heap_max = TW::kCap, node = elems;
    if (lane == 0) { // ---- one lane: trees.c:629-669 ----
        const bool has_static = kind != 2;
        while (heap_len < 2) {
            const int nn = max_code < 2 ? ++max_code : 0;
            t.heap[++heap_len] = heap_entry(1, 0, (uint32_t)nn);
            t.freq[nn] = 1; opt_len--;
            if (has_static) static_len -= static_len_of(kind, nn);
        }
        for (int n = heap_len / 2; n >= 1; n--) sift_down(t, heap_len, n);
        do {
            const uint32_t en = t.heap[1];
            t.heap[1] = t.heap[heap_len--]; sift_down(t, heap_len, 1);
            const uint32_t em = t.heap[1];
            const int n = (int)(en & 1023u), m = (int)(em & 1023u);
            t.heap[--heap_max] = (uint32_t)n; t.heap[--heap_max] = (uint32_t)m;
            const uint32_t dn = (en >> 10) & 63u, dm = (em >> 10) & 63u;
            t.dad[n] = t.dad[m] = (uint16_t)node;
            t.heap[1] = heap_entry((en >> 16) + (em >> 16), (dn >= dm ? dn : dm) + 1, (uint32_t)node);
            node++;
            sift_down(t, heap_len, 1);
        } while (heap_len >= 2);
        t.heap[--heap_max] = t.heap[1] & 1023u;
        for (int bits = 0; bits <= kMaxBits; bits++) t.bl_count[bits] = 0;
        const uint32_t root = t.heap[heap_max];
        t.len[root] = 0; t.dad[root] = (uint16_t)root;
    }
    heap_max = __builtin_amdgcn_readfirstlane(heap_max); max_code = __builtin_amdgcn_readfirstlane(max_code);
    opt_len = (uint32_t)__builtin_amdgcn_readfirstlane((int)opt_len); static_len = (uint32_t)__builtin_amdgcn_readfirstlane((int)static_len);
    sync();
    // ---- gen_bitlen (trees.c:490-567): depth of every node below the root ----
    uint32_t nd[kNodeRounds];
    const int first_h = heap_max + 1, nnodes = TW::kCap - first_h;
#pragma unroll
    for (int r = 0; r < kNodeRounds; r++) { const int i = r * 64 + (int)lane; nd[r] = i < nnodes ? t.heap[first_h + i] : 0xffffu; if (i < nnodes) t.len[nd[r]] = 1; }
    sync();
    for (int round = 0; round < 6; round++) {
        uint32_t pk[kNodeRounds]; // new distance | new ancestor << 16
#pragma unroll
        for (int r = 0; r < kNodeRounds; r++) {
            pk[r] = 0;
            if (nd[r] != 0xffffu) { const uint32_t a = t.dad[nd[r]]; pk[r] = ((uint32_t)t.len[nd[r]] + t.len[a]) | ((uint32_t)t.dad[a] << 16); }
        }
        sync(); // every lane has read what it needs of this round before anybody writes
#pragma unroll
        for (int r = 0; r < kNodeRounds; r++) if (nd[r] != 0xffffu) { t.len[nd[r]] = (uint16_t)pk[r]; t.dad[nd[r]] = (uint16_t)(pk[r] >> 16); }
        sync();
    }
    uint32_t over = 0, osum = 0, ssum = 0;
#pragma unroll
    for (int r = 0; r < kNodeRounds; r++) {
        if (nd[r] == 0xffffu) continue;
        const int n = (int)nd[r];
        uint32_t bits = t.len[n];
        if (bits > (uint32_t)max_length) { bits = (uint32_t)max_length; over++; t.len[n] = (uint16_t)bits; }
        if (n > max_code) continue; // not a leaf
        atomicAdd(reinterpret_cast<unsigned int *>(&t.bl_count[bits & ~1u]), 1u << ((bits & 1u) * 16)); // (two 16-bit counts per word)
        const uint32_t xb = (kind != 0 || n >= 257) ? extra_bits_of(kind, n) : 0u;
        osum += (uint32_t)t.freq[n] * (bits + xb);
        if (kind != 2) ssum += (uint32_t)t.freq[n] * (static_len_of(kind, n) + xb);
    }
    const int overflow = (int)wave_sum(over);
    opt_len += wave_sum(osum); static_len += wave_sum(ssum);
    sync();
    if (overflow > 0) { // (rare: a code longer than max_length; the repair walks the nodes in the heap's order)
        if (lane == 0) {
            int ov = overflow, bits, h = TW::kCap;
            do {
                bits = max_length - 1;
                while (t.bl_count[bits] == 0) bits--;
                t.bl_count[bits]--; t.bl_count[bits + 1] += 2; t.bl_count[max_length]--;
                ov -= 2;
            } while (ov > 0);
            for (bits = max_length; bits != 0; bits--) {
                int n = t.bl_count[bits];
                while (n != 0) {
                    const int m = (int)t.heap[--h];
                    if (m > max_code) continue;
                    if (t.len[m] != (unsigned)bits) {
                        opt_len += ((uint32_t)bits - (uint32_t)t.len[m]) * (uint32_t)t.freq[m];
                        t.len[m] = (uint16_t)bits;
                    }
                    n--;
                }
            }
        }
        opt_len = (uint32_t)__builtin_amdgcn_readfirstlane((int)opt_len);
        sync();
    }
    // ---- gen_codes (trees.c:577-609) ----
    uint32_t nc[kMaxBits + 1], c = 0;
    nc[0] = 0;
#pragma unroll
    for (int bits = 1; bits <= kMaxBits; bits++) { c = (c + t.bl_count[bits - 1]) << 1; nc[bits] = c; }
#pragma unroll
    for (int g = 0; g < kGroups; g++) {
        const int n = g * 64 + (int)lane;
        const uint32_t l = (n < elems && n <= max_code) ? t.len[n] : 0u;
        uint32_t code = 0;
#pragma unroll
        for (int bits = 1; bits <= kMaxBits; bits++) {
            const uint64_t m = __ballot(l == (uint32_t)bits);
            if (l == (uint32_t)bits) code = nc[bits] + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            nc[bits] += (uint32_t)__builtin_popcountll(m);
        }
        if (n < elems) { out_len[n] = (uint8_t)l; out_code[n] = l ? (uint16_t)(__brev(code) >> (32 - l)) : 0; }
    }
    sync();
    return max_code;
}

// scan_tree (kEmit false: count into blfreq) / send_tree (kEmit true), trees.c:707-797.  Inlined, the writer by reference: a
// writer whose address is passed to a real call lives in scratch memory, and every put() is then a trip to HBM and back.
template <bool kEmit>
__device__ __forceinline__ void walk_lengths(const uint8_t *len, int max_code, uint16_t *blfreq, BitWriter &bw, const uint16_t *blcode,
                                             const uint8_t *bllen)
{
    int prevlen = -1, curlen, nextlen = len[0], count = 0, max_count = 7, min_count = 4;
    if (nextlen == 0) { max_count = 138; min_count = 3; }
    for (int n = 0; n <= max_code; n++) {
        curlen = nextlen; nextlen = (n == max_code) ? 0xffff : len[n + 1];
        if (++count < max_count && curlen == nextlen) continue;
        else if (count < min_count) {
            if (kEmit) { do { bw.put(blcode[curlen], bllen[curlen]); } while (--count != 0); }
            else blfreq[curlen] += (uint16_t)count;
        } else if (curlen != 0) {
            if (curlen != prevlen) { if (kEmit) { bw.put(blcode[curlen], bllen[curlen]); count--; } else blfreq[curlen]++; }
            if (kEmit) { bw.put(blcode[16], bllen[16]); bw.put((uint32_t)(count - 3), 2); } else blfreq[16]++;
        } else if (count <= 10) {
            if (kEmit) { bw.put(blcode[17], bllen[17]); bw.put((uint32_t)(count - 3), 3); } else blfreq[17]++;
        } else {
            if (kEmit) { bw.put(blcode[18], bllen[18]); bw.put((uint32_t)(count - 11), 7); } else blfreq[18]++;
        }
        count = 0; prevlen = curlen;
        if (nextlen == 0) { max_count = 138; min_count = 3; }
        else if (curlen == nextlen) { max_count = 6; min_count = 3; }
        else { max_count = 7; min_count = 4; }
    }
}

// scan_tree / send_tree for a whole wave.  The state machine above depends on nothing but the run of equal lengths it stands in: a run of zeros is
// cut into pieces of 138 (REPZ_11_138) and what is left is REPZ_11_138 / REPZ_3_10 / one or two zeros; a run of a length L is L, REP_3_6 of up to 6
// for the first seven (four to seven: L and a repeat; fewer: literals), then REP_3_6 of 6 as long as six are left, then a repeat of 3..5 or one
// or two literals (trees.c:721-741: max_count / min_count 138/3 inside zeros, 7/4 at the start of a run, 6/3 inside it).  So the runs are found
// by ballot, compacted (one lane per run), and every lane plays its run's pieces: MODE 0 counts the symbols of the bit-length alphabet (LDS atomics),
// MODE 1 returns the bits the run's items take, MODE 2 ORs them into `obuf` at bit offset `at`.
template <int MODE>
__device__ __forceinline__ uint32_t run_items(uint32_t L, uint32_t N, uint32_t *blf, const uint16_t *blcode, const uint8_t *bllen, uint32_t *obuf, uint32_t at)
{
    uint32_t bits = 0, rem = N;
    bool first = true;
    auto item = [&](uint32_t sym, uint32_t xv, uint32_t xb) {
        if (MODE == 0) atomicAdd(&blf[sym], 1u);
        else {
            const uint32_t cl = bllen[sym], nb = cl + xb;
            if (MODE == 2) {
                const uint32_t v = (uint32_t)blcode[sym] | (xv << cl), w = at >> 5, sh = at & 31; // 14 bits at most
                atomicOr(&obuf[w], v << sh);
                if (sh + nb > 32) atomicOr(&obuf[w + 1], v >> (32 - sh));
                at += nb;
            }
            bits += nb;
        }
    };
    while (rem) {
        if (L == 0) {
            const uint32_t take = rem < 138 ? rem : 138;
            if (take < 3) { for (uint32_t k = 0; k < take; k++) item(0, 0, 0); }
            else if (take <= 10) item(17, take - 3, 3);
            else item(18, take - 11, 7);
            rem -= take;
        } else {
            const uint32_t maxc = first ? 7 : 6, minc = first ? 4 : 3, take = rem < maxc ? rem : maxc;
            if (take < minc) { for (uint32_t k = 0; k < take; k++) item(L, 0, 0); }
            else if (first) { item(L, 0, 0); item(16, take - 4, 2); }
            else item(16, take - 3, 2);
            rem -= take; first = false;
        }
    }
    return bits;
}

// the runs of len[0..max_code], compacted into spos[] (start positions, spos[nruns] = max_code + 1); returns nruns.  One wave.
__device__ __forceinline__ uint32_t find_runs(const uint8_t *len, int max_code, uint16_t *spos, uint32_t lane)
{
    uint32_t nruns = 0;
    for (int g = 0; g * 64 <= max_code; g++) {
        const int n = g * 64 + (int)lane;
        const bool valid = n <= max_code;
        const uint32_t cur = valid ? len[n] : 0u, prv = (valid && n > 0) ? len[n - 1] : 0u;
        const bool st = valid && (n == 0 || cur != prv);
        const uint64_t m = __ballot(st);
        if (st) spos[nruns + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)n;
        nruns += (uint32_t)__builtin_popcountll(m);
    }
    if (lane == 0) spos[nruns] = (uint16_t)(max_code + 1);
    __builtin_amdgcn_wave_barrier(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return nruns;
}

// MODE 0: scan_tree into blf[]; MODE 2: send_tree into obuf from bit offset at0 on, returns the bits written.  One wave; spos is scratch.
template <int MODE>
__device__ __forceinline__ uint32_t walk_lengths_wave(const uint8_t *len, int max_code, uint16_t *spos, uint32_t *blf, const uint16_t *blcode, const uint8_t *bllen,
                                                      uint32_t *obuf, uint32_t at0, uint32_t lane)
{
    const uint32_t nruns = find_runs(len, max_code, spos, lane);
    uint32_t done = 0;
    for (uint32_t r0 = 0; r0 < nruns; r0 += 64) {
        const uint32_t r = r0 + lane;
        uint32_t L = 0, N = 0;
        if (r < nruns) { const uint32_t p = spos[r]; N = (uint32_t)spos[r + 1] - p; L = len[p]; }
        if (MODE == 0) run_items<0>(L, N, blf, blcode, bllen, obuf, 0);
        else {
            const uint32_t mybits = run_items<1>(L, N, blf, blcode, bllen, obuf, 0);
            uint32_t x = mybits;
            for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if ((int)lane >= o) x += y; }
            run_items<2>(L, N, blf, blcode, bllen, obuf, at0 + done + x - mybits);
            done += __shfl(x, 63);
        }
    }
    __builtin_amdgcn_wave_barrier(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return done;
}

// Length and distance codes in closed form (the maps behind _length_code / _dist_code / base_length / base_dist / extra_lbits /
// extra_dbits of trees.h and trees.c:61-68): code and number of extra bits from the position of the leading one; the extra
// bits' value is the low bits of the length or distance itself, because every code's base is a multiple of its range.
// Per-lane lookups in constant memory cost a vector-memory round trip each, six of them per match token.
__device__ inline void len_code_of(uint32_t lc, uint32_t &code, uint32_t &extra) // lc = match length - 3
{
    const uint32_t e = (31u - (uint32_t)__clz((int)(lc | 8u))) - 2u; // 1..5 for lc >= 8
    const bool small = lc < 8, top = lc == 255;                       // length 258 has its own code without extra bits
    code = small ? lc : top ? 28u : 4u * e + 4u + ((lc >> e) & 3u);
    extra = (small || top) ? 0u : e;
}
__device__ inline void dist_code_bits(uint32_t d, uint32_t &code, uint32_t &extra) // d = match distance - 1
{
    const uint32_t msb = 31u - (uint32_t)__clz((int)(d | 2u));
    const bool small = d < 4;
    code = small ? d : 2u * msb + ((d >> (msb - 1u)) & 1u);
    extra = small ? 0u : msb - 1u;
}

__device__ inline void token_bits(uint32_t t, const uint16_t *lcode, const uint8_t *llen, const uint16_t *dcode, const uint8_t *dlen,
                                  uint64_t &bits, uint32_t &nb)
{
    uint32_t dist = t >> 8, lc = t & 255;
    if (dist == 0) { bits = lcode[lc]; nb = llen[lc]; return; }
    uint32_t c, xl, dc, xd;
    len_code_of(lc, c, xl);
    const uint32_t s = 257 + c;
    bits = lcode[s]; nb = llen[s];
    bits |= (uint64_t)(lc & ((1u << xl) - 1u)) << nb; nb += xl;
    dist--;
    dist_code_bits(dist, dc, xd);
    bits |= (uint64_t)dcode[dc] << nb; nb += dlen[dc];
    bits |= (uint64_t)(dist & ((1u << xd) - 1u)) << nb; nb += xd;
}

__device__ inline uint32_t block_reduce_add(uint32_t v, uint32_t *tmp)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) tmp[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t s = 0;
    for (int w = 0; w < kThreads / 64; w++) s += tmp[w];
    return s;
}

// exclusive prefix sum over the 256 lanes; *total receives the sum
__device__ inline uint32_t block_exclusive_scan(uint32_t v, uint32_t *tmp, uint32_t *total)
{
    uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, x = v;
    for (int o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(x, o); if (lane >= (uint32_t)o) x += y; }
    __syncthreads();
    if (lane == 63) tmp[wave] = x;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < wave; w++) base += tmp[w];
    uint32_t s = 0;
    for (int w = 0; w < kThreads / 64; w++) s += tmp[w];
    *total = s;
    return base + x - v;
}

#ifdef ZGPU_HUF_TIME // debug build only (scripts/huf_time.py): clock per phase, summed over chunks (lane 0's clock)
__device__ unsigned long long huf_time[8];
extern "C" __attribute__((visibility("default"))) void zgpu_debug_huf_time(unsigned long long *out, int reset)
{
    unsigned long long z[8] = {};
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(huf_time), sizeof z);
    if (reset) (void)hipMemcpyToSymbol(HIP_SYMBOL(huf_time), z, sizeof z);
}
#define HUF_T(i) do { if (tid == 0) { const unsigned long long t_ = wall_clock64(); t_acc[i] += t_ - t_prev; t_prev = t_; } } while (0)
#define HUF_T0() unsigned long long t_prev = wall_clock64(), t_acc[8] = {}
#define HUF_TEND() do { if (tid == 0) for (int i_ = 0; i_ < 8; i_++) atomicAdd(&huf_time[i_], t_acc[i_]); } while (0)
#else
#define HUF_T(i) do { } while (0)
#define HUF_T0() do { } while (0)
#define HUF_TEND() do { } while (0)
#endif

#ifndef ZGPU_HUF_WAVES
#define ZGPU_HUF_WAVES 8 // waves per SIMD the register budget is cut for (A/B builds: -DZGPU_HUF_WAVES=n; 6 and 7 spill less and keep fewer chunks resident)
#endif
// CONT (continuous stream, zgpu_cont.hip): the workgroup codes ONE block -- blk[blockIdx.x], its tokens a run of the batch's compact token array -- into a
// slot of its own from bit 0; where the block starts in the stream (a bit position that depends on every block in front of it) is the stitcher's business,
// and so are the bytes of a stored block (copied from the input there); the block's size, type and last_eob_len go back into blk[].
template <bool CONT>
__global__ void __launch_bounds__(kThreads, ZGPU_HUF_WAVES) huffman_kernel(ChunkGeom g, const uint32_t *__restrict__ tokens, ChunkMeta *meta, uint8_t *slots, uint32_t fixed_trees,
                                                                            ContBlk *blk, ContState *cst)
{
    __shared__ __attribute__((aligned(16))) TreeWork work0;
    __shared__ TreeWorkD work1;
    __shared__ uint32_t hist[kLCodes + kDCodes + 2];
    __shared__ uint16_t lcode[kLCodes + 2], dcode[kDCodes + 2], blcode[kBLCodes + 1];
    __shared__ uint8_t llen[kLCodes + 2], dlen[kDCodes + 2], bllen[kBLCodes + 1];
    __shared__ uint32_t tmp[4];
    // token emission works in the LDS of the literal/length tree's workspace (idle by then): one round of tokens, so that global reads are coalesced
    // and lanes still own runs, and the round's bits (a token is 48 bits at most), assembled with LDS atomics.  Sixteen workgroups per CU depend on it (and on 64 registers: eight waves per SIMD in the launch bounds).
    constexpr uint32_t kObufWords = (kEmitRound * 48 + 31 + 31) / 32 + 1;
    static_assert(sizeof(TreeWork) >= (kEmitRound + kObufWords) * 4 && alignof(TreeWork) >= 8, "emit buffers live in the tree workspace");
    uint32_t *tokbuf = reinterpret_cast<uint32_t *>(&work0), *obuf = tokbuf + kEmitRound;
    __shared__ uint32_t sh_optl, sh_statl, sh_optd, sh_statd, sh_btype, sh_lmax, sh_dmax;
    __shared__ uint64_t sh_bitpos;
    __shared__ uint32_t sh_carry; // the bits of the word at sh_bitpos that are not in memory yet (BitWriter)

    const uint32_t c = blockIdx.x, tid = threadIdx.x;
    if (!CONT && c >= g.nchunks) return;
    if (CONT && c >= cst->nblk) return; // (the launch is sized for the most blocks the batch can have)
    uint64_t lo = 0; uint32_t nbytes = 0;
    if (!CONT) chunk_span(g, c, lo, nbytes);
    const uint8_t *src = g.in + lo;
    const uint32_t *tok = CONT ? tokens + blk[c].tok0 : tokens + (size_t)c * kChunkMax;
    uint32_t *out = reinterpret_cast<uint32_t *>(slots + (size_t)c * g.slot_stride);
    const uint32_t ntok = CONT ? blk[c].nt : meta[c].ntok, nostore = CONT ? blk[c].nostore : meta[c].nostore;
    const bool final_chunk_here = CONT ? blk[c].eof != 0 : chunk_is_final(g, c);
    const uint32_t btok = g.block_tokens, nblocks = CONT ? 1u : ntok / btok + ((nostore & kFullFinalBlock) ? 0u : 1u); // (a last block filled by deflate_slow's trailing literal has no empty block behind it)
    const uint32_t *nostore_bits = (!CONT && g.nostore_bits) ? g.nostore_bits + (size_t)c * kGeoNostoreWords : nullptr;
    uint32_t block_start = CONT ? 0u : chunk_skip(g, c), data_type = 2; // (behind a preset dictionary)
    if (tid == 0) { const uint32_t pr = CONT ? 0u : chunk_prime(g, c); sh_bitpos = pr >> 16; sh_carry = pr & 0xffffu; } // (deflatePrime: bi_valid and bi_buf as the first block finds them, deflate.c:411-412)
    HUF_T0();

    for (uint32_t b = 0; b < nblocks; b++) {
        const uint32_t t0 = b * btok, t1 = (b + 1 == nblocks) ? ntok : t0 + btok, nt = t1 - t0;
        const uint32_t eof = (final_chunk_here && b + 1 == nblocks) ? 1u : 0u;
        // ---- histogram (init_block + tally) ----
        for (uint32_t i = tid; i < kLCodes + kDCodes + 2; i += kThreads) hist[i] = 0;
        __syncthreads();
        uint32_t bytes = 0;
        for (uint32_t i0 = t0 + tid; i0 < t1; i0 += 4 * kThreads) { // (four loads in flight: one at a time this loop is one memory latency per token)
            uint32_t tv[4];
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) tv[k] = i0 + k * kThreads < t1 ? tok[i0 + k * kThreads] : 0xFFFFFFFFu;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                if (i0 + k * kThreads >= t1) continue;
                const uint32_t t = tv[k], dist = t >> 8, lc = t & 255;
                if (dist == 0) { atomicAdd(&hist[lc], 1u); bytes += 1; }
                else { uint32_t c, xl, dc, xd; len_code_of(lc, c, xl); dist_code_bits(dist - 1, dc, xd); atomicAdd(&hist[257 + c], 1u); atomicAdd(&hist[kLCodes + dc], 1u); bytes += lc + 3; }
            }
        }
        const uint32_t stored_len = block_reduce_add(bytes, tmp);
        __syncthreads();
        if (tid == 0) hist[kEndBlock] = 1;
        __syncthreads();
        if ((CONT ? blk[c].first != 0 : b == 0) && stored_len > 0) { // set_data_type, trees.c:1126-1139 (first block of the stream decides)
            bool bin = false;
            if (tid < 32 && (tid < 9 || tid >= 14)) bin = hist[tid] != 0;
            data_type = __syncthreads_or(bin) ? 0u : 1u;
        }
        for (uint32_t i = tid; i < kLCodes; i += kThreads) work0.freq[i] = (uint16_t)hist[i];
        if (tid < kDCodes) work1.freq[tid] = (uint16_t)hist[kLCodes + tid];
        __syncthreads();
        HUF_T(0);
        // ---- trees ----
        if (tid < 64) { // wave 0: the literal/length tree; wave 1: the distance tree
            uint32_t o = 0, s = 0;
            const int mc = build_tree_wave(work0, kLCodes, 0, kMaxBits, lcode, llen, o, s, tid);
            if (tid == 0) { sh_lmax = (uint32_t)mc; sh_optl = o; sh_statl = s; }
        } else {
            uint32_t o = 0, s = 0;
            const int mc = build_tree_wave(work1, kDCodes, 1, kMaxBits, dcode, dlen, o, s, tid - 64);
            if (tid == 64) { sh_dmax = (uint32_t)mc; sh_optd = o; sh_statd = s; }
        }
        __syncthreads();
        HUF_T(1);
        if (tid < 64) { // wave 0: the bit-length alphabet, the block type, the header (the other wave waits at the barrier below)
            const uint32_t lane = tid;
            auto wsync = [&]() { __builtin_amdgcn_wave_barrier(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
            const int lmax = (int)sh_lmax, dmax = (int)sh_dmax;
            uint32_t *blf = hist;                                         // (the histogram is dead until the next block clears it)
            uint16_t *spos = reinterpret_cast<uint16_t *>(hist + 32);     // run starts of a code-length sequence, kLCodes + 1 at most
            static_assert((kLCodes + kDCodes + 2 - 32) * 2 >= kLCodes + 2, "run starts fit behind the counts");
            if (lane < kBLCodes) blf[lane] = 0;
            wsync();
            walk_lengths_wave<0>(llen, lmax, spos, blf, nullptr, nullptr, nullptr, 0, lane);
            walk_lengths_wave<0>(dlen, dmax, spos, blf, nullptr, nullptr, nullptr, 0, lane);
            TreeWork &w = work0;
            if (lane < kBLCodes) w.freq[lane] = (uint16_t)blf[lane];
            wsync();
            if (lane == 0) {
                uint32_t opt_len = sh_optl + sh_optd, static_len = sh_statl + sh_statd;
                uint32_t dummy = 0;
                build_tree(w, kBLCodes, 2, kMaxBLBits, blcode, bllen, opt_len, dummy);
                int max_blindex;
                for (max_blindex = kBLCodes - 1; max_blindex >= 3; max_blindex--) if (bllen[kTables.bl_order[max_blindex]] != 0) break;
                opt_len += 3 * (uint32_t)(max_blindex + 1) + 5 + 5 + 4;
                uint32_t opt_lenb = (opt_len + 3 + 7) >> 3, static_lenb = (static_len + 3 + 7) >> 3;
                if (static_lenb <= opt_lenb) opt_lenb = static_lenb;
                uint32_t btype;
                if (stored_len + 4 <= opt_lenb && !(nostore_bits ? (nostore_bits[b >> 5] >> (b & 31u)) & 1u : (nostore >> b) & 1u)) btype = 0;
                else if (fixed_trees || static_lenb == opt_lenb) btype = 1; // Z_FIXED: trees.c:986
                else btype = 2;
                sh_btype = btype;
                HUF_T(2);
                // ---- block header: the fixed part ----
                BitWriter bw; bw.begin(out, sh_bitpos, sh_carry);
                bw.put((btype << 1) + eof, 3);
                if (btype == 0) {
                    bw.align_byte(); bw.put(stored_len & 0xffff, 16); bw.put(~stored_len & 0xffff, 16);
                } else if (btype == 2) {
                    bw.put((uint32_t)(lmax + 1 - 257), 5); bw.put((uint32_t)(dmax + 1 - 1), 5); bw.put((uint32_t)(max_blindex + 1 - 4), 4);
                    for (int r = 0; r <= max_blindex; r++) bw.put(bllen[kTables.bl_order[r]], 3);
                }
                sh_bitpos = bw.pos();
                sh_carry = bw.carry();
            }
            wsync();
            if (sh_btype == 2) { // ---- the two code-length sequences behind it: every lane its runs, the bits put together in LDS as a round of tokens is ----
                constexpr uint32_t kHdrWords = (31 + (kLCodes + kDCodes) * 14 + 31) / 32 + 2;
                static_assert(kHdrWords <= kObufWords, "the header's bits fit the round buffer");
                const uint64_t pos0 = sh_bitpos;
                const uint32_t sh0 = (uint32_t)(pos0 & 31);
                for (uint32_t i = lane; i < kHdrWords; i += 64) obuf[i] = 0;
                wsync();
                if (lane == 0) obuf[0] = sh_carry;
                wsync();
                uint32_t nbits = walk_lengths_wave<2>(llen, lmax, spos, blf, blcode, bllen, obuf, sh0, lane);
                nbits += walk_lengths_wave<2>(dlen, dmax, spos, blf, blcode, bllen, obuf, sh0 + nbits, lane);
                const uint32_t nfull = (sh0 + nbits) >> 5, nwords = (sh0 + nbits + 31) >> 5;
                uint32_t *dstw = out + (pos0 >> 5);
                for (uint32_t wd = lane; wd < nwords; wd += 64) { const uint32_t x = obuf[wd]; if (wd < nfull) dstw[wd] = x; else sh_carry = x; }
                if (lane == 0) { if (nfull == nwords) sh_carry = 0; sh_bitpos = pos0 + nbits; }
            }
        }
        __syncthreads();
        HUF_T(3);
        const uint32_t btype = sh_btype;
        uint64_t bitpos = sh_bitpos;
        if (CONT) {
            if (tid == 0) { // (the slot holds a coded block from bit 0; a stored block's bytes are taken from the input by the stitcher)
                ContBlk &o = blk[c];
                o.btype = btype; o.stored_len = stored_len; o.eob_len = btype == 0 ? 8u : btype == 1 ? 7u : (uint32_t)llen[kEndBlock];
                if (o.first) cst->data_type = data_type;
            }
        }
        if (btype == 0 && CONT) { }
        else if (btype == 0) {
            uint8_t *dst = reinterpret_cast<uint8_t *>(out) + (bitpos >> 3);
            if (tid == 0) { // the header's last bytes, still in the carry (the stored bytes follow them inside the same word)
                const uint32_t k = (uint32_t)(bitpos >> 3) & 3u, cw = sh_carry;
                for (uint32_t j = 0; j < k; j++) dst[(int)j - (int)k] = (uint8_t)(cw >> (8 * j));
            }
            for (uint32_t i = tid; i < stored_len; i += kThreads) dst[i] = src[block_start + i];
            bitpos += (uint64_t)stored_len * 8;
        } else {
            if (btype == 1) { // static trees: load the fixed codes into the same LDS tables
                for (uint32_t i = tid; i < kLCodes; i += kThreads) { lcode[i] = kTables.sl_code[i]; llen[i] = kTables.sl_len[i]; }
                if (tid < kDCodes) { dcode[tid] = kTables.sd_code[tid]; dlen[tid] = 5; }
                __syncthreads();
            }
            // Rounds of kEmitRound tokens (the end-of-block code is the token behind the last): read coalesced into LDS, kEmitPer consecutive ones
            // per lane sized and coded once, a prefix scan for the bit offsets, the bits put together in LDS (ds_or), the round's words stored
            // coalesced; the bits of the word the round stops in stay behind as the carry of the next round (or of the next writer).
            for (uint32_t i = tid; i < kObufWords; i += kThreads) obuf[i] = 0; // (the first round's barrier stands between this and the first bits)
            for (uint32_t r0 = 0; r0 <= nt; r0 += kEmitRound) {
                uint32_t tv[kEmitPer];
#pragma unroll
                for (uint32_t k = 0; k < kEmitPer; k++) { const uint32_t j = r0 + k * kThreads + tid; tv[k] = j < nt ? tok[t0 + j] : 0u; }
#pragma unroll
                for (uint32_t k = 0; k < kEmitPer; k++) tokbuf[k * kThreads + tid] = tv[k];
                __syncthreads();
                if (tid == 0) obuf[0] = sh_carry; // what the writer before this round left of its last word (lane 0 cleared the word; the scan's barriers stand before the round's bits)
                const uint4 mine = reinterpret_cast<const uint4 *>(tokbuf)[tid];
                const uint32_t mt[4] = {mine.x, mine.y, mine.z, mine.w};
                uint64_t v[kEmitPer]; uint32_t nb[kEmitPer], mybits = 0;
#pragma unroll
                for (uint32_t k = 0; k < kEmitPer; k++) {
                    const uint32_t j = r0 + tid * kEmitPer + k;
                    v[k] = 0; nb[k] = 0;
                    if (j < nt) token_bits(mt[k], lcode, llen, dcode, dlen, v[k], nb[k]);
                    else if (j == nt) { v[k] = lcode[kEndBlock]; nb[k] = llen[kEndBlock]; }
                    mybits += nb[k];
                }
                uint32_t total, off = block_exclusive_scan(mybits, tmp, &total);
                const uint32_t sh0 = (uint32_t)(bitpos & 31);
                uint32_t at = sh0 + off;
#pragma unroll
                for (uint32_t k = 0; k < kEmitPer; k++) {
                    if (nb[k]) {
                        const uint32_t w = at >> 5, sh = at & 31;
                        atomicOr(&obuf[w], (uint32_t)(v[k] << sh));
                        if (sh + nb[k] > 32) atomicOr(&obuf[w + 1], (uint32_t)(v[k] >> (32 - sh)));
                        if (sh + nb[k] > 64) atomicOr(&obuf[w + 2], (uint32_t)(v[k] >> (64 - sh)));
                    }
                    at += nb[k];
                }
                __syncthreads();
                const uint32_t nfull = (sh0 + total) >> 5, nwords = (sh0 + total + 31) >> 5;
                uint32_t *dstw = out + (bitpos >> 5);
                for (uint32_t w = tid; w < nwords; w += kThreads) {
                    const uint32_t x = obuf[w];
                    obuf[w] = 0;
                    if (w < nfull) dstw[w] = x; else sh_carry = x; // (one lane at most: the word the round stops in)
                }
                if (tid == 0 && nfull == nwords) sh_carry = 0;
                bitpos += total;
            }
        }
        block_start += stored_len;
        __syncthreads();
        HUF_T(4);
        if (tid == 0) {
            sh_bitpos = bitpos;
            if (btype == 0) { // the stored bytes that share their word with what follows: back from memory (this workgroup wrote them in front of the barrier)
                const uint32_t k = (uint32_t)(bitpos >> 3) & 3u;
                sh_carry = k ? *reinterpret_cast<volatile uint32_t *>(out + (bitpos >> 5)) & ((1u << (8 * k)) - 1u) : 0u;
            }
        }
        __syncthreads();
    }
    if (CONT) {
        if (tid == 0) { BitWriter bw; bw.begin(out, sh_bitpos, sh_carry); bw.store_tail(); blk[c].nbits = (uint32_t)sh_bitpos; }
    } else if (tid == 0) {
        BitWriter bw; bw.begin(out, sh_bitpos, sh_carry);
        if (!final_chunk_here) { bw.put(0, 3); bw.align_byte(); bw.put(0, 16); bw.put(0xffff, 16); } // flush marker
        else bw.align_byte();                                                                        // bi_windup
        uint64_t endpos = bw.pos();
        bw.store_tail();
        meta[c].out_bytes = (uint32_t)(endpos >> 3);
        meta[c].data_type = data_type;
    }
    HUF_TEND();
}

void launch_huffman(const ChunkGeom &g, const uint32_t *tokens, ChunkMeta *meta, uint8_t *slots, hipStream_t st, bool fixed_trees)
{
    hipLaunchKernelGGL(huffman_kernel<false>, dim3(g.nchunks), dim3(kThreads), 0, st, g, tokens, meta, slots, fixed_trees ? 1u : 0u, nullptr, nullptr);
}
// the blocks of a continuous stream's batch: g carries block_tokens and slot_stride only
void launch_huffman_cont(const ChunkGeom &g, const uint32_t *compact_tokens, uint32_t nblk, ContBlk *blk, ContState *cst, uint8_t *slots, hipStream_t st, bool fixed_trees)
{
    if (nblk) hipLaunchKernelGGL(huffman_kernel<true>, dim3(nblk), dim3(kThreads), 0, st, g, compact_tokens, nullptr, slots, fixed_trees ? 1u : 0u, blk, cst);
}

} // namespace zgpu
