// zgpu_lz_serial.hip -- LZ77 stage, serial form: one lane per chunk, hash tables in HBM.
//
// This is the reference's deflate_fast (levels 1-3) / deflate_slow (levels 4-9) control flow
// (/root/reference/qcsrc/deflate.c:1448-1546, 1554-1674) with longest_match (deflate.c:1027-1168),
// expressed on chunk offsets.  It is the only LZ77 form for levels 1-3, whose hash chains depend on
// the parse (insertions are skipped after long matches, deflate.c:1510-1534), and the cross-check
// form for levels 4-9 (zgpu_lz_parallel.hip is the fast one there).
//
// Window bookkeeping without moving memory.  The reference keeps 16-bit window indices in head[]/prev[]
// and rebases all of them when it slides the window (deflate.c:1293-1326).  A chunk is at most 64 KiB, so
// at most one slide happens, late (window index >= 65274).  Here tables hold p+1 (p = chunk offset, 0 =
// never inserted) and the window index is derived on the fly: w = p + base - off, base = 3 for a
// position-0-matchable chunk, off = 32768 once the slide has happened.  An entry is the
// reference's NIL exactly when its derived index is <= 0.
#include "zgpu_common.h"
#include <cstdlib>

namespace zgpu {

struct __attribute__((packed, aligned(1))) U32s { uint32_t v; };
struct __attribute__((packed, aligned(1))) U64s { uint64_t v; };
struct __attribute__((packed, aligned(1))) U128s { uint4 v; };
constexpr uint32_t kSerialWin = 128, kSerialWinStride = kSerialWin + 16; // bytes of input a lane keeps in LDS; the lanes' windows 144 bytes apart

// kGeo: deflateInit2's windowBits / memLevel are not the default (deflate.c:222-297): window size, hash width and shift and the tokens of a block
// are run-time values, the window may slide many times in a chunk (the derived index covers that: `off` grows by a window each time), and the
// "may not be stored" flags go to a bit array of their own (a chunk can have more than 32 blocks).
template <bool kGeo>
struct SerialLzT {
    const uint8_t *__restrict__ in;
    uint32_t n, base, off, start; // start: a preset dictionary occupies positions [0, start) (deflate.c:315-354)
    // The tables (see insert): 16 bytes per hash bucket and 16 bytes per window position, each holding what the walk needs of TWO chain members.
    uint4 *head;  // x: e1 | e2 << 16 (the bucket's newest position + 1, and that position's prev[] entry), y: the launch tag, z/w: e1's first six bytes
    uint4 *link;  // per window position q: x,y = prev[q] (e2) | q's six bytes << 16; z,w = prev[e2] (e3) | e2's six bytes << 16
    uint32_t tag; // a bucket whose tag is not this launch's is empty (the tables are zeroed once, when they are allocated; the tag counts launches)
    uint32_t in_e2; uint64_t in_pfx; // what insert() found in the bucket beside the position it returns: that position's prev[] entry and six bytes
    uint32_t *tok;
    uint32_t ntok, blk_tok0, nblk, nostore, block_start;
    LevelCfg cfg;
    uint32_t g_wsize, g_hmask, g_hshift, g_btok; // (kGeo)
    uint32_t *g_nostore;
    __device__ uint32_t wsize() const { return kGeo ? g_wsize : kWSize; }
    __device__ uint32_t wmask() const { return wsize() - 1u; }
    __device__ uint32_t maxdist() const { return wsize() - kMinLookahead; }
    __device__ uint32_t btok() const { return kGeo ? g_btok : kBlockTokens; }
    __device__ uint32_t hash(uint32_t b0, uint32_t b1, uint32_t b2) const { return kGeo ? ((b0 << (2 * g_hshift)) ^ (b1 << g_hshift) ^ b2) & g_hmask : hash3(b0, b1, b2); } // UPDATE_HASH three times (deflate.c:181)

    __device__ int widx(uint32_t p) const { return (int)(p + base) - (int)off; }
    // derived window index of a table entry; <= 0 means NIL
    __device__ int entry_w(uint32_t e) const { return e == 0 ? 0 : (int)(e - 1 + base) - (int)off; }

    // the first eight bytes of the string at p (zeros behind the end of the chunk).  They come from a window of the input in LDS, 128 bytes per lane, refilled
    // every 64: the loop asks for them once per inserted position, and between two of a lane's steps the line they lie in has long left the caches (the
    // other 8 000 lanes of the XCD pull three table lines per step through a 4 MiB L2), so every insert fetched its input line from the fabric again.
    uint32_t win_lds, win_base; // LDS byte address of this lane's window; the chunk offset it starts at (0x80000000: nothing in it yet)
    __device__ void win_fill(uint32_t p)
    {
        win_base = p & ~63u;
#pragma unroll
        for (uint32_t k = 0; k < kSerialWin / 16; k++) {
            const uint32_t o = win_base + 16 * k;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (o + 16 <= n) v = reinterpret_cast<const U128s *>(in + o)->v;
            else if (o < n) { uint32_t w[4] = {0, 0, 0, 0}; for (uint32_t b = 0; o + b < n && b < 16; b++) w[b >> 2] |= (uint32_t)in[o + b] << (8 * (b & 3)); v = make_uint4(w[0], w[1], w[2], w[3]); }
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            *reinterpret_cast<__attribute__((address_space(3))) u32x4 *>((uintptr_t)(win_lds + 16 * k)) = u32x4{v.x, v.y, v.z, v.w};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // (the window is this lane's alone; the reads below are asm and must stay behind these stores)
    }
    __device__ uint64_t bytes8(uint32_t p)
    {
        if (p - win_base > kSerialWin - 8) win_fill(p); // (also when p lies in front of the window, or nothing is in it: the difference wraps)
        uint64_t v;
        asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(win_lds + (p - win_base)) : "memory");
        return v;
    }
    __device__ uint32_t insert(uint32_t p)
    {
        // Every global load of this lane-per-chunk loop is a dependent round trip of one to two microseconds under load, and pulls a whole line for the
        // two bytes it wants: with head[] / prev[] as the reference has them 870 GB crossed the fabric per 4 GiB launch at level 1 -- 4 TB/s, what HBM gives
        // for random lines -- so what counts is LINES PER TOKEN.  A bucket therefore holds, beside its newest position e1, that position's prev[] entry e2 and
        // its first six bytes: the first candidate of a search is judged from the line that had to be read anyway.  A window position's entry holds its own
        // prev[] entry and six bytes AND those of the position it links to: every line read during longest_match's walk serves two candidates, and the
        // input itself is read only for a candidate that agrees in all six bytes.  (All of it is copies of values that never change once written: prev[q]
        // and the bytes at q.  Entries are written in position order, four to a line.)
        const uint64_t v = bytes8(p);
        const uint32_t h = hash((uint32_t)v & 255u, (uint32_t)(v >> 8) & 255u, (uint32_t)(v >> 16) & 255u);
        const uint4 b = head[h];
        const bool live = b.y == tag;
        const uint32_t e1 = live ? b.x & 0xffffu : 0u, e2 = (live && e1) ? b.x >> 16 : 0u;
        const uint64_t pfx1 = ((uint64_t)b.w << 32 | b.z) & 0xffffffffffffull;
        const uint64_t me = (uint64_t)e1 | (v << 16), nx = (uint64_t)e2 | (pfx1 << 16);
        link[(p + base) & wmask()] = make_uint4((uint32_t)me, (uint32_t)(me >> 32), (uint32_t)nx, (uint32_t)(nx >> 32));
        head[h] = make_uint4((p + 1) | (e1 << 16), tag, (uint32_t)v, (uint32_t)(v >> 32) & 0xffffu);
        in_e2 = e2; in_pfx = pfx1;
        return e1;
    }

    // the slide test of fill_window (deflate.c:1293); called wherever the reference calls fill_window
    __device__ void refill(uint32_t p) { if (widx(p) >= (int)(wsize() + maxdist())) off += wsize(); }
    // fill_window (deflate.c:1265-1350) with the whole chunk at hand: the slide if it is due, then as much input as the window has room for; returns
    // the new count of buffered bytes
    __device__ uint32_t fill(uint32_t p, uint32_t buffered)
    {
        if (!kGeo) { refill(p); return n; } // (one slide at most, late: everything that is left fits behind it)
        do {
            refill(p);
            const uint32_t more = 2 * wsize() - (buffered - p) - (uint32_t)widx(p), left = n - buffered;
            buffered += more < left ? more : left;
        } while (buffered - p < kMinLookahead && buffered < n);
        return buffered;
    }

    __device__ void cut_block(uint32_t p_end)
    {
        if ((int)(block_start + base) - (int)off < 0) { // buf == NULL, deflate.c:1365-1367: the block's first byte has left the window
            if (kGeo) g_nostore[nblk >> 5] |= 1u << (nblk & 31u); else nostore |= 1u << nblk;
        }
        nblk++; blk_tok0 = ntok; block_start = p_end;
    }
    __device__ bool emit(uint32_t t) { tok[ntok++] = t; return ntok - blk_tok0 == btok(); }

    // longest_match; e0 is the table entry of the first candidate.  Returns the match length and sets mstart.
    __device__ uint32_t longest(uint32_t p, uint32_t e0, uint32_t prev_length, uint32_t &mstart)
    {
        uint32_t chain = cfg.chain, look = n - p, nice = cfg.nice, best = prev_length;
        int w = widx(p), limit = w > (int)maxdist() ? w - (int)maxdist() : 0;
        uint32_t cap = look < kMaxMatch ? look : kMaxMatch;
        if (prev_length >= cfg.good) chain >>= 2;
        if (nice > look) nice = look;
        const uint8_t *scan = in + p;
        const uint64_t scan6 = bytes8(p) & 0xffffffffffffull;
        // the candidate in hand: its position + 1, six bytes and prev[] entry; `ahead`: the same of the candidate it links to are known already
        uint32_t e = e0, elink = in_e2, alink = 0;
        uint64_t epfx = in_pfx, apfx = 0;
        bool ahead = false;
        for (;;) {
            const uint32_t q = e - 1;
            const uint64_t x6 = epfx ^ scan6;
            uint32_t l = x6 ? (uint32_t)__builtin_ctzll(x6) >> 3 : 6u;
            if (l == 6 && cap > 6 && best < cap) { // all six agree: the rest from the input (a candidate no longer than the best so far changes nothing, deflate.c:1121-1124)
                const uint8_t *m = in + q;
                if (best < 8 || m[best] == scan[best]) {
                    while (l + 8 <= cap) { // (cap <= look: both reads stay inside the chunk)
                        const uint64_t x = reinterpret_cast<const U64s *>(m + l)->v ^ reinterpret_cast<const U64s *>(scan + l)->v;
                        if (x) { l += (uint32_t)__builtin_ctzll(x) >> 3; goto compared; }
                        l += 8;
                    }
                    while (l < cap && m[l] == scan[l]) l++;
                }
            }
        compared:
            if (l > cap) l = cap;
            if (l > best) { mstart = q; best = l; if (l >= nice) break; }
            e = elink;
            if (e == 0 || entry_w(e) <= limit) break;
            if (--chain == 0) break;
            if (ahead) { epfx = apfx; elink = alink; ahead = false; }
            else {
                const uint4 t = link[(e - 1 + base) & wmask()];
                elink = t.x & 0xffffu; epfx = ((uint64_t)t.y << 16) | (t.x >> 16);
                alink = t.z & 0xffffu; apfx = ((uint64_t)t.w << 16) | (t.z >> 16);
                ahead = true;
            }
        }
        return best <= look ? best : look;
    }
    // longest_match_fast (deflate.c:1173-1228; reached with Z_RLE only in this build): the common prefix with the one candidate
    __device__ uint32_t fast_match(uint32_t p, uint32_t e0, uint32_t &mstart) const
    {
        const uint32_t q = e0 - 1, look = n - p, cap = look < kMaxMatch ? look : kMaxMatch;
        uint32_t l = 0;
        if (cap < 2 || in[q] != in[p] || in[q + 1] != in[p + 1]) return kMinMatch - 1;
        while (l < cap && in[q + l] == in[p + l]) l++;
        if (l < kMinMatch) return kMinMatch - 1;
        mstart = q;
        return l;
    }
};

// `hand_on` (deflate_fast only): this loop's time goes with the tokens of a chunk -- each one a chain of dependent memory accesses -- and a launch takes
// as long as its slowest chunks: 4 GiB of the Silesia-mix take 377 ms at level 1, 232 ms without the 5 % of its chunks that do not compress (65 000
// literals each), which the wave-per-chunk kernel does in 3.5 ms apiece (scripts/serial_classes.py).  So a lane that finds, every 4096 bytes, more than
// one token per two bytes behind it (text has one per five) gives the chunk up (returns false, the kernel marks it kHandedOn) and the engine runs zgpu_lz_fastwin.hip
// over the chunks given up.  Either kernel's tokens are the reference's, so who compresses a chunk shows in the time only.
template <bool kSlow, bool kGeo>
__device__ bool lz_serial_chunk(SerialLzT<kGeo> &s, bool hand_on)
{
    const uint32_t n = s.n, maxd = s.maxdist();
    uint32_t room = 2 * s.wsize() - s.base, buffered = n < room ? n : room; // first fill_window (deflate.c:1275,1342)
    uint32_t p = s.start, match_len = kMinMatch - 1, prev_len = 0, mstart = 0, prev_match = 0, hh = 0;
    bool pending = false;
    for (uint32_t q = 0; q + kMinMatch <= s.start; q++) s.insert(q); // all dictionary strings but the last two (deflate.c:345-351)
    uint32_t next_check = 4096;
    for (;;) {
        if (!kSlow && hand_on && p >= next_check) {
            if (s.ntok * 2 > p) return false;
            next_check += 4096;
        }
        if (buffered - p < kMinLookahead) { buffered = s.fill(p, buffered); if (n == p) break; }
        uint32_t look = n - p;
        if (look >= kMinMatch) hh = s.insert(p);
        int hw = s.entry_w(hh), w = s.widx(p);
        bool cut = false;
        if (kSlow) {
            prev_len = match_len; prev_match = mstart; match_len = kMinMatch - 1;
            if (hw > 0 && prev_len < s.cfg.lazy && (uint32_t)(w - hw) <= maxd) {
                if (s.cfg.strategy != kHuffmanOnly && s.cfg.strategy != kRle) match_len = s.longest(p, hh, prev_len, mstart);
                else if (s.cfg.strategy == kRle && w - hw == 1) match_len = s.fast_match(p, hh, mstart);
                if (match_len <= 5 && (s.cfg.strategy == kFiltered || (match_len == kMinMatch && p - mstart > kTooFar))) match_len = kMinMatch - 1;
            }
            if (prev_len >= kMinMatch && match_len <= prev_len) {
                uint32_t max_insert = p + look - kMinMatch, k = prev_len - 2;
                cut = s.emit(tok_match(p - 1 - prev_match, prev_len - kMinMatch));
                do { if (++p <= max_insert) hh = s.insert(p); } while (--k != 0);
                pending = false; match_len = kMinMatch - 1; p++;
                if (cut) s.cut_block(p);
            } else if (pending) {
                cut = s.emit(tok_lit(s.in[p - 1]));
                if (cut) s.cut_block(p);
                p++;
            } else { pending = true; p++; }
        } else {
            if (hw > 0 && (uint32_t)(w - hw) <= maxd) {
                if (s.cfg.strategy != kHuffmanOnly && s.cfg.strategy != kRle) match_len = s.longest(p, hh, kMinMatch - 1, mstart);
                else if (s.cfg.strategy == kRle && w - hw == 1) match_len = s.fast_match(p, hh, mstart);
            }
            if (match_len >= kMinMatch) {
                cut = s.emit(tok_match(p - mstart, match_len - kMinMatch));
                look -= match_len;
                if (match_len <= s.cfg.lazy && look >= kMinMatch) { // cfg.lazy is max_insert_length here (h/deflate.h:176)
                    match_len--;
                    do { p++; hh = s.insert(p); } while (--match_len != 0);
                    p++;
                } else { p += match_len; match_len = 0; }
            } else { cut = s.emit(tok_lit(s.in[p])); p++; }
            if (cut) s.cut_block(p);
        }
    }
    // the last byte's literal is tallied behind the loop and its "buffer full" is not looked at (deflate.c:1660-1665): when it is the token that fills
    // the block, that block -- all its tokens -- is the final one, with no empty block behind it
    if (kSlow && pending && s.emit(tok_lit(s.in[p - 1]))) s.nostore |= kFullFinalBlock;
    s.cut_block(p); // the final block (its emission happens in the Huffman stage)
    return true;
}

// grid: one lane per chunk of the batch.  tables: per chunk 16 bytes per hash bucket, then 16 bytes per window position (SerialLzT::insert): 2 x 32768 uint4,
// or (kGeo) 2^hash_bits + 2^w_bits at a stride of kGeoTableEntries.  Zeroed when allocated; `tag` (never 0, never repeated on these tables) marks this launch's buckets.
template <bool kGeo>
__global__ void __launch_bounds__(64) lz_serial_kernel(ChunkGeom g, LevelCfg cfg, uint4 *tables, uint32_t *tokens, ChunkMeta *meta, uint32_t lanes, uint32_t *nostore_bits, uint32_t hand_on, uint32_t tag)
{
    // `lanes` chunks per wave: a wave's step takes as long as its slowest lane's memory access, and fewer lanes per wave
    // means more waves to overlap those waits (the vector work per step is next to nothing)
    if (threadIdx.x >= lanes) return;
    uint32_t c = blockIdx.x * lanes + threadIdx.x;
    if (c >= g.nchunks) return;
    uint64_t lo; uint32_t n;
    chunk_span(g, c, lo, n);
    SerialLzT<kGeo> s;
    s.in = g.in + lo; s.n = n;
    s.base = chunk_base(g, c); s.off = 0; s.start = chunk_skip(g, c);
    if (kGeo) {
        s.g_wsize = 1u << cfg.w_bits; s.g_hmask = (1u << cfg.hash_bits) - 1u; s.g_hshift = (cfg.hash_bits + kMinMatch - 1) / kMinMatch; s.g_btok = g.block_tokens;
        s.g_nostore = nostore_bits + (size_t)c * kGeoNostoreWords;
        for (uint32_t i = 0; i < kGeoNostoreWords; i++) s.g_nostore[i] = 0;
        s.head = tables + (size_t)c * kGeoTableEntries; s.link = s.head + (1u << cfg.hash_bits);
    } else {
        s.g_wsize = kWSize; s.g_hmask = kHashMask; s.g_hshift = 5; s.g_btok = kBlockTokens; s.g_nostore = nullptr;
        s.head = tables + (size_t)c * kSerialTableEntries; s.link = s.head + kHashSize;
    }
    extern __shared__ __attribute__((aligned(16))) uint8_t serial_win[];
    s.win_lds = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)serial_win + threadIdx.x * kSerialWinStride; s.win_base = 0x80000000u;
    s.tag = tag; s.in_e2 = 0; s.in_pfx = 0;
    s.tok = tokens + (size_t)c * kChunkMax;
    s.ntok = 0; s.blk_tok0 = 0; s.nblk = 0; s.nostore = 0; s.block_start = s.start; s.cfg = cfg;
    const bool done = cfg.slow ? lz_serial_chunk<true, kGeo>(s, false) : lz_serial_chunk<false, kGeo>(s, hand_on != 0 && s.start == 0);
    meta[c].ntok = done ? s.ntok : kHandedOn; meta[c].nostore = s.nostore; meta[c].in_bytes = s.n;
}

void launch_lz_serial(const ChunkGeom &g, LevelCfg cfg, uint4 *tables, uint32_t *tokens, ChunkMeta *meta, hipStream_t st, uint32_t *nostore_bits, bool hand_on, uint32_t tag)
{
    // chunks per wave: measured best (MI355X, level 1) where the launch has about 4096 waves -- 16 per CU; 64 chunks per wave
    // (1024 waves at 4 GiB) is 30 % slower, 8192 waves again slower.  ZGPU_SERIAL_LANES overrides.
    static int forced = -1;
    if (forced < 0) { const char *e = getenv("ZGPU_SERIAL_LANES"); forced = e ? atoi(e) : 0; if (forced < 0 || forced > 64) forced = 0; }
    uint32_t lanes = (uint32_t)forced;
    if (!lanes) { lanes = 1; while (lanes < 64 && (uint64_t)lanes * 4096 < g.nchunks) lanes <<= 1; }
    if (cfg.w_bits) hipLaunchKernelGGL(lz_serial_kernel<true>, dim3((g.nchunks + lanes - 1) / lanes), dim3(64), lanes * kSerialWinStride, st, g, cfg, tables, tokens, meta, lanes, nostore_bits, 0u, tag);
    else hipLaunchKernelGGL(lz_serial_kernel<false>, dim3((g.nchunks + lanes - 1) / lanes), dim3(64), lanes * kSerialWinStride, st, g, cfg, tables, tokens, meta, lanes, nostore_bits, hand_on ? 1u : 0u, tag);
}

} // namespace zgpu
