// zgpu_common.h -- shared device/host definitions of the MI355X DEFLATE engine.
//
// Constants follow /root/reference (zlib 1.2.3): h/zutil.h:81-82 (MIN/MAX_MATCH), h/deflate.h:30-48,271-276
// (L_CODES.., MIN_LOOKAHEAD, MAX_DIST), qcsrc/deflate.c:105-110 (NIL, TOO_FAR), qcsrc/deflate.c:137-149
// (level table), qcsrc/trees.c:61-71 (extra bits, bl_order).  The code tables of h/trees.h are
// regenerated here at compile time from RFC 1951 instead of being copied.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zgpu {

constexpr uint32_t kWSize = 32768, kWMask = kWSize - 1, kHashSize = 32768, kHashMask = kHashSize - 1;
constexpr uint32_t kMinMatch = 3, kMaxMatch = 258, kMinLookahead = kMaxMatch + kMinMatch + 1;
constexpr uint32_t kMaxDist = kWSize - kMinLookahead; // 32506
constexpr uint32_t kTooFar = 4096;
constexpr uint32_t kChunkMax = 65536;
constexpr uint32_t kBlockTokens = 16383;       // lit_bufsize-1: a block is cut after this many tokens
constexpr uint32_t kSerialTableEntries = 32768 + 32768; // uint4 entries per chunk of the lane-per-chunk loop's tables: one per hash bucket, one per window position (1 MiB)
constexpr uint32_t kGeoTableEntries = 65536 + 32768;   // the same at memLevel 9 / windowBits 15
constexpr uint32_t kGeoNostoreWords = 18;     // memLevel 1: 127 tokens a block, up to 517 blocks in a chunk
constexpr uint32_t kGeoSlotStride = 65536 + 8192 + 1024 + 64 + 5 * 32 * kGeoNostoreWords + 448; // deflateBound's arithmetic (deflate.c:513-515) for a chunk whose blocks may not be stored, + a header per block
constexpr uint32_t kMaxBlocks = 6;
// ChunkMeta::nostore bit 31: deflate_slow tallies the last byte's literal behind its loop without looking at "buffer full" (deflate.c:1660-1665), so when that
// literal is the token that fills a block, the block is flushed as the chunk's LAST one -- there is no empty block behind it
constexpr uint32_t kFullFinalBlock = 1u << 31;
// ChunkMeta::ntok of a chunk the lane-per-chunk loop has handed on (see lz_serial_kernel): no tokens yet, the wave-per-chunk kernel takes it
constexpr uint32_t kHandedOn = 0xFFFFFFFFu;             // ceil(65536/16383) + a possible empty final block
constexpr int kLCodes = 286, kDCodes = 30, kBLCodes = 19, kHeapSize = 2 * kLCodes + 1, kMaxBits = 15, kMaxBLBits = 7;
constexpr int kEndBlock = 256;
constexpr uint32_t kSlotStride = 65536 + 256;  // per-chunk output slot (worst case: 5 stored blocks + marker = +30)

struct LevelCfg { uint32_t good, lazy, nice, chain, slow, strategy; uint32_t w_bits, hash_bits; }; // w_bits, hash_bits: 0 = the default geometry (15, 15); else deflateInit2's windowBits and memLevel + 7 // strategy: Z_DEFAULT_STRATEGY 0 .. Z_FIXED 4 (h/zlib.h:176-181)
constexpr uint32_t kFiltered = 1, kHuffmanOnly = 2, kRle = 3, kFixed = 4;
inline LevelCfg level_cfg(int level)
{
    static const LevelCfg t[10] = {{0, 0, 0, 0, 0, 0},      {4, 4, 8, 4, 0, 0},      {4, 5, 16, 8, 0, 0},      {4, 6, 32, 32, 0, 0},
                                   {4, 4, 16, 16, 1, 0},    {8, 16, 32, 32, 1, 0},   {8, 16, 128, 128, 1, 0},  {8, 32, 128, 256, 1, 0},
                                   {32, 128, 258, 1024, 1, 0}, {32, 258, 258, 4096, 1, 0}};
    return t[level];
}

// Per-chunk record passed between the LZ77 stage, the Huffman stage and the stitcher.
struct ChunkMeta {
    uint32_t ntok;        // tokens produced by the LZ77 stage
    uint32_t nostore;     // bit b set: block b may not be emitted stored (reference: buf == NULL after the slide); kFullFinalBlock: see there
    uint32_t out_bytes;   // compressed bytes in the chunk's slot
    uint32_t data_type;   // Z_BINARY 0 / Z_TEXT 1 / Z_UNKNOWN 2 from the first non-empty block
    uint32_t adler_a, adler_b; // Adler-32 halves of the chunk bytes, as if started from 1
    uint32_t in_bytes;
    uint32_t crc;         // CRC-32 of the chunk bytes (crc_kernel; gzip wrapper only)
};

// Where the chunks of one launch live.  Either uniform (chunk k = bytes [k*chunk_size, ...)) or an explicit
// segment table (chunk k = bytes [seg_off[k], seg_off[k+1])), each segment at most 65536 bytes.
struct ChunkGeom {
    const uint8_t *in;
    uint64_t in_bytes;
    const uint64_t *seg_off; // nullptr: uniform chunking
    uint64_t chunk0;         // global index of the first chunk of this batch
    uint64_t final_chunk;    // global index of the chunk that carries BFINAL (~0: none)
    uint32_t chunk_size;
    uint32_t nchunks;        // chunks in this batch
    uint32_t all_final;      // every chunk is a complete stream of its own
    uint32_t pos0_mode;      // 0 none, 1 chunks other than global chunk 0, 2 all chunks are position-0 matchable
    uint32_t skip0;          // preset dictionary: global chunk 0 starts with this many bytes that are window content, not data
    uint32_t prime;          // deflatePrime: (nbits << 16) | value, the bits global chunk 0 starts behind
    // deflateInit2's geometry (deflate.c:222-297) when it is not the default windowBits 15 / memLevel 8 (the lane-per-chunk loop serves it):
    uint32_t block_tokens;   // lit_bufsize - 1: a block is cut after this many tokens (kBlockTokens by default)
    uint32_t slot_stride;    // bytes between the chunks' output slots (kSlotStride by default)
    const uint32_t *nostore_bits; // nullptr: ChunkMeta::nostore has a bit per block; else kGeoNostoreWords words per chunk (a chunk may have 517 blocks)
    // A launch over a LIST of the batch's chunks (levels 1-3: the chunks the lane-per-chunk loop handed on, zgpu_engine.hip): workgroup c works on
    // chunk chunk_map[c] of the batch -- the input, tokens and meta of that chunk; its scratch (sorted buckets) is slot c.  nullptr: chunk c.
    const uint32_t *chunk_map;
    // Continuous stream (round 4, zgpu_cont.hip): tile_stride != 0 makes "chunk" c of the launch the TILE chunk0 + c -- the 64 KiB of the buffer that start at
    // tile_w0 + (chunk0 + c) * tile_stride, clipped at in_bytes.  Tiles overlap: each brings the 32512 bytes in front of its own positions along as history.
    uint64_t tile_w0;
    uint32_t tile_stride;
    // positions of the buffer (ascending buffer offsets) that are NOT in the hash chains although they have three bytes: the two positions in front of every
    // earlier Z_SYNC_FLUSH / Z_PARTIAL_FLUSH point (lookahead < MIN_MATCH when the loop stood there, deflate.c:1470,1576; zlib 1.2.3 never inserts them later)
    const uint64_t *excl; uint32_t nexcl;
};
__device__ inline uint32_t chunk_of(const ChunkGeom &g, uint32_t c) { return g.chunk_map ? g.chunk_map[c] : c; }
__device__ inline void chunk_span(const ChunkGeom &g, uint32_t c, uint64_t &lo, uint32_t &n)
{
    const uint64_t gc = g.chunk0 + chunk_of(g, c);
    if (g.tile_stride) { lo = g.tile_w0 + gc * g.tile_stride; const uint64_t rem = g.in_bytes - lo; n = (uint32_t)(rem < kChunkMax ? rem : kChunkMax); return; }
    if (g.seg_off) { lo = g.seg_off[gc]; n = (uint32_t)(g.seg_off[gc + 1] - lo); }
    else { lo = gc * g.chunk_size; uint64_t rem = g.in_bytes - lo; n = (uint32_t)(rem < g.chunk_size ? rem : g.chunk_size); }
}
__device__ inline bool chunk_is_final(const ChunkGeom &g, uint32_t c) { return g.all_final || g.chunk0 + chunk_of(g, c) == g.final_chunk; }
__device__ inline uint32_t chunk_skip(const ChunkGeom &g, uint32_t c) { return g.chunk0 + chunk_of(g, c) == 0 ? g.skip0 : 0u; }
__device__ inline uint32_t chunk_prime(const ChunkGeom &g, uint32_t c) { return g.chunk0 + chunk_of(g, c) == 0 ? g.prime : 0u; }
__device__ inline uint32_t chunk_base(const ChunkGeom &g, uint32_t c) { return (g.pos0_mode == 2 || (g.pos0_mode == 1 && g.chunk0 + chunk_of(g, c) != 0)) ? 3u : 0u; }

// ---- continuous stream: tiles (see zgpu_cont.hip) ----
constexpr uint32_t kTileStride = 32512;            // positions a tile parses = bytes of history it brings along (MAX_DIST = 32506 fits)
constexpr uint32_t kTileH1 = 2 * kTileStride;      // a tile parses the local positions [h0, h1) = [32512, 65024) (the first tile of a feed: from its entry)
constexpr uint32_t kTileSlack = kChunkMax - kTileH1; // 512: a game that starts below h1 searches up to 254 positions further (each lazy step needs a longer match) and needs MAX_MATCH of lookahead there
constexpr uint32_t kTileEntries = kTileSlack + 1;  // where a tile can be entered: h0 + 0 .. 512 (the last game of the tile before ends at most 512 behind its h1)
constexpr uint32_t kTileExitStride = 520;          // u16 per tile: exit (relative to h1) as a function of the entry (relative to h0)
struct TileGeom {
    uint64_t abs0;     // stream position of the buffer's first byte
    uint64_t e0;       // buffer offset of the neutral position the first tile of the feed is entered at (>= tile_w0, at most kTileStride above it)
    uint64_t end;      // buffer offset the parse runs up to: in_bytes when the segment ends there (flush / finish), else in_bytes - kTileSlack
    uint64_t nil_pos;  // buffer offset of the one position whose first candidate, exactly MAX_DIST back, is NIL (slid out at that very loop top), ~0: none
    uint32_t abs0_nil; // stream position 0 is NIL (a stream without dictionary: window index 0 is never a candidate, deflate.c:1479)
    uint32_t pad;
    uint16_t *exits;   // [tile of the launch][kTileExitStride]
    uint16_t *entry;   // [tile of the feed]: where the parse enters the tile, relative to its h0
};
// is buffer offset `at` one of the positions that are not in the hash chains (ChunkGeom::excl, ascending)?  lower bound of `at` in the list
__device__ inline uint32_t excl_lower(const ChunkGeom &g, uint64_t at)
{
    uint32_t lo = 0, hi = g.nexcl;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (g.excl[mid] < at) lo = mid + 1; else hi = mid; }
    return lo;
}
__device__ inline bool excl_has(const ChunkGeom &g, uint64_t at) { const uint32_t i = excl_lower(g, at); return i < g.nexcl && g.excl[i] == at; }
__device__ inline void tile_span(const ChunkGeom &g, const TileGeom &tg, uint32_t c, uint64_t &wb, uint32_t &nloc, uint32_t &h0, uint32_t &h1, uint32_t &nent)
{
    const uint64_t ti = g.chunk0 + c;
    wb = g.tile_w0 + ti * g.tile_stride;
    const uint64_t rem = g.in_bytes - wb, lim = tg.end > wb ? tg.end - wb : 0;
    nloc = (uint32_t)(rem < kChunkMax ? rem : kChunkMax);
    h0 = ti == 0 ? (uint32_t)(tg.e0 - g.tile_w0) : kTileStride;
    h1 = (uint32_t)(lim < kTileH1 ? lim : kTileH1);
    if (h1 < h0) h1 = h0;
    nent = ti == 0 ? 1u : kTileEntries;
}

// levels 1-3 of a continuous stream (fastwin_tile_kernel, zgpu_lz_fastwin.hip): what the rounds of a batch of tiles work on
constexpr uint32_t kInsWords = (kChunkMax - kTileStride) / 32; // 1032 words: one bit per local position 32512 .. 65535, "in the hash chains"
struct FastTiles {
    const uint16_t *exit_cur; uint16_t *exit_new;  // [tile of the batch]: where the tile's parse ends, relative to its h1
    const uint32_t *ins0, *ins1; uint32_t *ins0w, *ins1w; // [tile][kInsWords], two buffers
    const uint8_t *cur;                            // [tile]: the buffer its current bits are in; a tile that is parsed writes the other one
    const uint8_t *active; uint8_t *changed;       // [tile]
    const uint32_t *prev_ins;                      // the bits in front of the batch's first tile (its local positions 0 .. 33023): the feed's history, or the batch before
    uint32_t round;
    uint32_t warm_mode;                            // 1: the launch's tiles (but the batch's first) start from nothing and parse their history as well; 0: from where the tile in front ended, with its bits
    const uint32_t *list;                          // the tiles to parse in this round (round 0: nullptr = all of the batch): workgroup w works on tile list[w]
    uint32_t *low_out;                             // the feed's first tile also says which of its local positions 0 .. 32511 are in the chains (history and, at a stream's start, its own)
    uint32_t *dbg;                                 // (ZGPU_FAST_TRACE: eight words per tile about what changed)
    // what a tile's current results were made from: the history's bits it used [tile][kInsWords] and where it entered; kept[tile]: this round's parse stopped
    // early because nothing it could reach had changed -- its results stand, its buffers are not flipped (nullptr: every active tile is parsed to its end)
    uint32_t *used_ins; uint16_t *entry_used; uint8_t *kept;
    uint32_t *stat;                                // (ZGPU_FAST_TRACE) [0] active tiles with the entry they had, [1] parses stopped early, [2] ... that met a different token first, [3] ... with changes out of reach of nothing
};

// One block of a continuous stream (blocks are cut every 16383 tokens counted from the start of the STREAM, h/deflate.h:313): filled in by cont_table_kernel,
// coded by huffman_kernel<true> into a slot of its own from bit 0, put in its place in the stream (a bit position) by cont_stitch_kernel.
struct ContBlk {
    uint64_t end_pos;    // buffer offset behind the block's last byte
    uint64_t start_pos;  // buffer offset of its first byte (stored blocks are copied from there)
    uint32_t last_len;   // bytes its last token covers: the loop stood at end_pos - last_len + 1 when the block was flushed
    uint32_t tok0, nt;   // its tokens in the batch's compact token array
    uint32_t eof;        // the stream's final block
    uint32_t nostore;    // its first byte has left the reference's window by the time it is flushed (buf == NULL, deflate.c:1364-1367)
    uint32_t first;      // the stream's first block: decides strm->data_type (trees.c:934-935)
    uint32_t nbits;      // out: bits of the coded block (btype 1, 2)
    uint32_t btype;      // out: 0 stored, 1 static, 2 dynamic
    uint32_t stored_len; // out: bytes of input the block covers
    uint32_t eob_len;    // out: last_eob_len behind it (trees.c:1117,1206)
};
// carried from batch to batch on the device and from feed to feed through the host
struct ContState {
    uint64_t out_bits;    // bit position in the output where the next block starts
    uint64_t block_start; // buffer offset of the first byte of the block that is being filled
    uint64_t ntokens;     // (diagnostic)
    uint64_t e_next;      // buffer offset of the neutral position the parse has reached
    uint32_t carry_n;     // tokens of the block being filled that are waiting in the carry buffer
    uint32_t data_type;   // Z_UNKNOWN 2 until the first block has decided
    uint32_t last_eob;
    uint32_t overflow;
    uint32_t nblk, total; // the current batch: blocks to emit, tokens in the compact array (carry included)
    uint32_t first_block; // 1 until the stream's first block has been emitted
    uint32_t entry_k;     // entry of the next batch's first tile relative to its h0 (the chain's hand-over)
};

// token: bits 0..7 = literal byte or (match length - 3); bits 8..23 = match distance (0 for a literal)
__host__ __device__ inline uint32_t tok_lit(uint32_t c) { return c; }
__host__ __device__ inline uint32_t tok_match(uint32_t dist, uint32_t lenm3) { return lenm3 | (dist << 8); }

// ---------------------------------------------------------------------------------------------------
// RFC 1951 code tables, generated at compile time (trees.c:238-316 builds the same ones at run time).
struct CodeTables {
    uint8_t len_code[256];   // match length-3 -> length code 0..28
    uint8_t dist_code[512];  // d_code() lookup, h/deflate.h:290-291
    uint16_t base_len[29];
    uint16_t base_dist[30];
    uint8_t xl[29];
    uint8_t xd[30];
    uint16_t sl_code[288];   // static literal/length codes, bit-reversed
    uint8_t sl_len[288];
    uint16_t sd_code[30];
    uint8_t bl_order[19];
    uint8_t xbl[19];
};

constexpr unsigned bit_reverse(unsigned v, int len)
{
    unsigned r = 0;
    for (int i = 0; i < len; i++) { r = (r << 1) | (v & 1); v >>= 1; }
    return r;
}

constexpr CodeTables make_code_tables()
{
    CodeTables t{};
    const uint8_t xl[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    const uint8_t xd[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    for (int i = 0; i < 29; i++) t.xl[i] = xl[i];
    for (int i = 0; i < 30; i++) t.xd[i] = xd[i];
    for (int i = 0; i < 19; i++) { t.bl_order[i] = order[i]; t.xbl[i] = 0; }
    t.xbl[16] = 2; t.xbl[17] = 3; t.xbl[18] = 7;
    int length = 0;
    for (int code = 0; code < 28; code++) {
        t.base_len[code] = (uint16_t)length;
        for (int n = 0; n < (1 << xl[code]); n++) t.len_code[length++] = (uint8_t)code;
    }
    t.len_code[255] = 28; t.base_len[28] = 0;
    int dist = 0, code = 0;
    for (code = 0; code < 16; code++) {
        t.base_dist[code] = (uint16_t)dist;
        for (int n = 0; n < (1 << xd[code]); n++) t.dist_code[dist++] = (uint8_t)code;
    }
    dist >>= 7;
    for (; code < 30; code++) {
        t.base_dist[code] = (uint16_t)(dist << 7);
        for (int n = 0; n < (1 << (xd[code] - 7)); n++) t.dist_code[256 + dist++] = (uint8_t)code;
    }
    unsigned blc[16] = {}, next[16] = {}, c = 0;
    for (int n = 0; n < 288; n++) { t.sl_len[n] = (uint8_t)(n < 144 ? 8 : n < 256 ? 9 : n < 280 ? 7 : 8); blc[t.sl_len[n]]++; }
    for (int n = 1; n <= 15; n++) { c = (c + blc[n - 1]) << 1; next[n] = c; }
    for (int n = 0; n < 288; n++) t.sl_code[n] = (uint16_t)bit_reverse(next[t.sl_len[n]]++, t.sl_len[n]);
    for (int n = 0; n < 30; n++) t.sd_code[n] = (uint16_t)bit_reverse((unsigned)n, 5);
    return t;
}

// One copy in constant memory per translation unit that includes this header.
__constant__ const CodeTables kTables = make_code_tables();

__device__ inline uint32_t dist_code_of(uint32_t dist_minus1)
{
    return dist_minus1 < 256 ? kTables.dist_code[dist_minus1] : kTables.dist_code[256 + (dist_minus1 >> 7)];
}

// 3-byte hash the reference's rolling UPDATE_HASH converges to (deflate.c:170,189-192)
__host__ __device__ inline uint32_t hash3(uint32_t b0, uint32_t b1, uint32_t b2) { return ((b0 << 10) ^ (b1 << 5) ^ b2) & kHashMask; }

#define ZGPU_HIP_CHECK(expr)                                                                     \
    do {                                                                                         \
        hipError_t err__ = (expr);                                                               \
        if (err__ != hipSuccess) return zgpu::fail_hip(e, err__, #expr, __FILE__, __LINE__);     \
    } while (0)

} // namespace zgpu
