/* zamd_gpu.h -- C ABI of the MI355X-native DEFLATE engine (libzamd_gpu.so).
 *
 * This is the thin HIP shim SURVEY.md section 8(b) asks for: plain pointers, sizes and int status codes,
 * no C++ or torch types.  It sits *below* the unchanged zlib API: the zlib-compatible host library
 * (include/zamd_zlib.h, libzamd_z.so) implements deflate()/inflate()/compress2()/uncompress() on top
 * of the *_host entry points below, at the place where the reference dispatches to its per-level
 * compress function and to inflate_fast:
 *
 *   zgpu_deflate_*   replaces  configuration_table[level].func(s, flush)   /root/reference/qcsrc/deflate.c:790
 *                    i.e. deflate_fast / deflate_slow + longest_match      qcsrc/deflate.c:1027-1168,1448-1674
 *                    and _tr_flush_block / compress_block                  qcsrc/trees.c:921-1016,1072-1118
 *                    and the adler32 update inside read_buf                qcsrc/deflate.c:968-970
 *   zgpu_inflate_*   replaces  the TYPE..LEN..CHECK states of inflate()    qcsrc/inflate.c:773-1098
 *                    with inflate_table and inflate_fast                   qcsrc/inftrees.c:32, qcsrc/inffast.c:67
 *
 * Unit of work: a "chunk" = up to 65536 input bytes compressed as an independent raw-deflate segment,
 * exactly what the reference emits for a fresh stream deflateInit2(level, 8, -15, 8, 0) fed the chunk
 * and finished with Z_FULL_FLUSH (or Z_FINISH for the last chunk of a stream) -- "mode B" of
 * SURVEY.md section 8(c).  Output is bit-exact with that.
 *
 * Status codes are zlib's (h/zlib.h:170-178) so the host library can return them unchanged.
 */
#ifndef ZAMD_GPU_H
#define ZAMD_GPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZGPU_OK 0
#define ZGPU_ERRNO (-1)        /* HIP runtime failure; text in zgpu_engine_error() */
#define ZGPU_STREAM_ERROR (-2) /* bad parameters */
#define ZGPU_DATA_ERROR (-3)   /* malformed deflate data (inflate) */
#define ZGPU_MEM_ERROR (-4)    /* device allocation failed */
#define ZGPU_BUF_ERROR (-5)    /* output capacity too small */

#define ZGPU_CHUNK_MAX 65536u

/* flags for zgpu_deflate_params.flags */
#define ZGPU_F_FINAL 1u     /* the last chunk of this call ends the stream (BFINAL + byte align) */
#define ZGPU_F_ZLIB_WRAP 2u /* prepend the 2-byte zlib header, append the big-endian Adler-32 (needs FINAL) */
#define ZGPU_F_POS0 4u      /* chunks other than the first are "position-0 matchable" (mode A) */
#define ZGPU_F_POS0_ALL 8u  /* every chunk, including the first, is position-0 matchable */
#define ZGPU_F_GZIP_WRAP 16u /* gzip member: the 10-byte header deflate() writes without a gz_header (qcsrc/deflate.c:578-596),
                               the body, CRC-32 and length little-endian (deflate.c:833-843); needs FINAL */
#define ZGPU_F_CRC32 32u     /* also compute the CRC-32 of the input (result.crc32) without writing a wrapper */

#define ZGPU_F_CONTINUOUS 64u /* zgpu_deflate_device / zgpu_deflate_host: the input as ONE continuous stream, byte for byte what the reference's un-flushed
                                deflate() -- plain compress2(), qcsrc/compress.c:22-58 -- emits: the 32 KiB window slides through the whole input
                                (qcsrc/deflate.c:1266-1358), matches cross every 64 KiB boundary, blocks are cut every 16383 tokens counted from the
                                stream's start (h/deflate.h:313).  Needs FINAL; chunk_size, POS0*, prime do not apply; no chunk table. */

/* LZ77 match-finder implementation selector (debug / A-B measurements) */
#define ZGPU_LZ_AUTO 0
#define ZGPU_LZ_SERIAL 1   /* one lane per chunk, tables in HBM: any level 1..9 */
#define ZGPU_LZ_PARALLEL 2 /* static-chain search over all positions (link ring in LDS) + scan parse: levels 4..9 */
#define ZGPU_LZ_SORTED 3   /* the same search over counting-sorted hash buckets: levels 4..9 */
#define ZGPU_LZ_WALK 4     /* parse-driven search over the same buckets (only the positions deflate_slow searches): levels 4..9, the default */
#define ZGPU_LZ_FAST 5     /* levels 1-3: deflate_fast over the sorted buckets, one lane per chunk, a flag byte per position for "in the chains" */
#define ZGPU_LZ_FASTWIN 6  /* levels 1-3: deflate_fast, one wave per chunk, 64 positions a step, window + chain bits in LDS: the default there */

typedef struct zgpu_engine zgpu_engine;

typedef struct {
    int32_t level;       /* 1..9 */
    uint32_t chunk_size; /* 1..65536; 0 selects 65536 */
    uint32_t flags;      /* ZGPU_F_* */
    int32_t lz_impl;     /* ZGPU_LZ_* */
    int32_t strategy;    /* 0 Z_DEFAULT_STRATEGY, 1 Z_FILTERED, 2 Z_HUFFMAN_ONLY, 3 Z_RLE, 4 Z_FIXED (qcsrc/deflate.c:1485-1497,
                            1594-1611; trees.c:986).  Not served by ZGPU_LZ_PARALLEL. */
    uint32_t prime;      /* deflatePrime (qcsrc/deflate.c:404-413): (nbits << 16) | value, nbits 0..16.  The first chunk of the call starts behind these
                            bits (its blocks are shifted, the padding of its stored blocks and of its end moves with them).  Not with a
                            wrapper flag.  0: none. */
} zgpu_deflate_params;

typedef struct {
    uint64_t out_bytes;  /* bytes written to the output buffer */
    uint64_t nchunks;
    uint32_t adler32;    /* Adler-32 of the whole input (1 for empty input) */
    uint32_t data_type;  /* Z_BINARY 0 / Z_TEXT 1 / Z_UNKNOWN 2, as strm->data_type after the first block */
    uint64_t ntokens;    /* literal/match tokens over all chunks (diagnostic) */
    uint32_t crc32;      /* CRC-32 of the whole input when ZGPU_F_GZIP_WRAP or ZGPU_F_CRC32 was given, else 0 */
    uint32_t reserved;
} zgpu_deflate_result;

typedef struct {
    uint64_t out_bytes;   /* bytes produced */
    uint32_t adler32;     /* Adler-32 of the produced bytes */
    int32_t first_bad_chunk; /* -1, or the first chunk that failed */
    int32_t error_code;   /* ZGPU_OK or the failure of first_bad_chunk */
    uint32_t error_msg;   /* index into zgpu_inflate_message() */
    uint32_t crc32;       /* CRC-32 of the produced bytes */
    uint32_t in_used_bits; /* with `incomplete`: bits of the byte at in + in_used that are consumed as well (0..7): the stream goes on at that bit
                              (zgpu_inflate_stream_host3's start_bit) */
    /* zgpu_inflate_stream_host2 / 3 with ZGPU_INF_STREAM (otherwise in_used = in_bytes, the flags 0): */
    uint64_t in_used;     /* input bytes consumed: up to the end of the final block, or of the last segment / piece that decoded */
    uint32_t stream_end;  /* the final block was reached */
    uint32_t incomplete;  /* the input stops inside a block: out_bytes / in_used cover the segments in front of it (possibly none) */
} zgpu_inflate_result;

/* ---- multi-GPU: the per-rank raw bodies gathered into ONE RFC 1950 stream on rank 0 over RCCL / xGMI (zlib_amd/csrc/zgpu_comm.hip) ----
 * One process per GPU; rank r compresses its contiguous chunk range with zgpu_deflate_device (flags = ZGPU_F_FINAL on the last rank only, no
 * wrapper) and the only exchange is this gather.  The reference has no distributed path.  RCCL is dlopen()ed by the first call here.
 *   rank 0: zgpu_comm_unique_id(id); the 128 bytes reach the other ranks by any channel; every rank: zgpu_comm_create(device, world, rank, id, &c);
 *   per stream, every rank: zgpu_deflate_gather_sizes (an all-gather of three u64 per rank: rank 0 learns the exact size of the stream),
 *   then zgpu_deflate_gather (one send per peer; on rank 0 one receive per peer at its offset, all in one group).  Both calls return when the
 *   stream has drained on the calling rank: d_body may be reused then.  The gathered stream's header is the default strategy's (FLEVEL from the level). */
typedef struct zgpu_comm zgpu_comm;
#define ZGPU_COMM_ID_BYTES 128
int zgpu_comm_unique_id(void *id128);
int zgpu_comm_create(int device, int world, int rank, const void *id128, zgpu_comm **out);
void zgpu_comm_destroy(zgpu_comm *c);
const char *zgpu_comm_error(void); /* the calling thread's last failure */
/* table: world x {body bytes, Adler-32, input bytes}; offsets[r] = where rank r's body starts in the stream (offsets[0] = 2),
 * offsets[world] = where the 4-byte trailer goes, *total = the stream's length */
void zgpu_gather_layout(int world, const uint64_t *table, uint64_t *offsets, uint64_t *total);
int zgpu_deflate_gather_sizes(zgpu_comm *c, uint64_t body_bytes, uint32_t adler32, uint64_t in_bytes, uint64_t *table, uint64_t *total, void *hip_stream);
int zgpu_deflate_gather(zgpu_comm *c, const void *d_body, const uint64_t *table, int level, void *d_out, uint64_t out_cap, uint32_t *adler_out,
                        void *hip_stream);

/* ---- engine lifetime ---- */
int zgpu_device_count(void);
int zgpu_engine_create(int device, zgpu_engine **out);
void zgpu_engine_destroy(zgpu_engine *e);
const char *zgpu_engine_error(const zgpu_engine *e);
const char *zgpu_version(void);
uint64_t zgpu_debug_handed_on(const zgpu_engine *e); /* diagnostic: chunks of level 1-3 calls that went from the lane-per-chunk loop to the wave-per-chunk kernel */

/* deflateInit2's geometry (qcsrc/deflate.c:222-297) for the calls that follow: windowBits 9..15 (the window is 2^windowBits bytes, matches reach
 * 2^windowBits - 262 back, the window slides every 2^windowBits bytes) and memLevel 1..9 (hash of memLevel + 7 bits, a block is cut after
 * 2^(memLevel+6) - 1 tokens).  15 / 8 is the default and what every kernel is built for; any other geometry is served by the lane-per-chunk loop
 * (ZGPU_LZ_SERIAL) and the same block-construction kernel, bit-exact with the reference's chunk function under that deflateInit2.  Output
 * capacity: zgpu_deflate_bound_geometry. */
int zgpu_deflate_set_geometry(zgpu_engine *e, int window_bits, int mem_level);
uint64_t zgpu_deflate_bound_geometry(uint64_t in_bytes, uint32_t chunk_size, int window_bits, int mem_level);

/* deflateTune (qcsrc/deflate.c:453-470): while `on`, the deflate calls of this engine use these four parameters of the match
 * search instead of the level's row of configuration_table (deflate.c:137-149); the level keeps its compress function
 * (deflate_fast for 1..3, deflate_slow for 4..9). */
int zgpu_deflate_set_tuning(zgpu_engine *e, int on, uint32_t good_length, uint32_t max_lazy, uint32_t nice_length, uint32_t max_chain);

/* worst-case output bytes for in_bytes of input cut into chunk_size pieces (framing included) */
uint64_t zgpu_deflate_bound(uint64_t in_bytes, uint32_t chunk_size);

/* ---- deflate ---- */
/* Device-resident: d_in/d_out are device pointers on the engine's device; d_chunk_offsets (optional)
 * receives nchunks+1 byte offsets of each chunk's segment inside d_out.  hip_stream may be NULL
 * (engine's own stream).  Blocks until the result is known. */
int zgpu_deflate_device(zgpu_engine *e, const void *d_in, uint64_t in_bytes, const zgpu_deflate_params *p,
                        void *d_out, uint64_t out_cap, uint64_t *d_chunk_offsets, zgpu_deflate_result *res,
                        void *hip_stream);
/* Host buffers: stages through device memory (H2D, kernels, D2H). chunk_offsets (optional, host)
 * receives nchunks+1 offsets. */
int zgpu_deflate_host(zgpu_engine *e, const void *in, uint64_t in_bytes, const zgpu_deflate_params *p, void *out,
                      uint64_t out_cap, uint64_t *chunk_offsets, zgpu_deflate_result *res);

/* ---- ONE continuous stream, feed by feed (zlib_amd/csrc/zgpu_cont.hip): what deflate() of the zlib API is built on ----
 * The stream's state between feeds lives with the CALLER (zgpu_cont_state + the tokens of the block that is still filling); the engine keeps nothing.
 * All positions are positions in the stream (a preset dictionary's bytes count: the first one is position 0).
 *   hist, in   buf = hist followed by in (two host buffers, so that a caller's large input need not be copied behind the little history the stream keeps):
 *              the bytes the parse can still reach followed by the bytes it has not parsed.  buf[0] is stream position cs->abs0, which must be at most
 *              cs->entry - 32512 (or 0) -- and at most cs->block_start when that lies within 65536 + 512 of cs->entry (a block that may still be stored
 *              is copied from there)
 *   check_from offset into buf of the first byte whose checksum is wanted: res->adler32 / crc32 cover buf[check_from ..) alone (hist_bytes + in_bytes: nothing)
 *   mode       ZGPU_CONT_MORE: more input follows; the parse stops 512 bytes (or a little less) in front of the end of buf and cs->entry says where --
 *              the caller keeps the bytes from cs->entry - 32512 on for the next feed.  ZGPU_CONT_FLUSH: the segment ends here as at Z_SYNC_FLUSH /
 *              Z_PARTIAL_FLUSH / Z_FULL_FLUSH (lookahead runs out at the end of buf, the block that is filling is closed; the marker behind it is the
 *              caller's to write -- it knows cs->bit_count / bit_value and cs->last_eob).  ZGPU_CONT_FINISH: the same with the final bit, padded to a byte.
 *   hist_bits  levels 1-3 (deflate_fast leaves the strings inside a long match out of its hash chains, qcsrc/deflate.c:1510-1534, so WHICH positions of the
 *              history are in the chains is part of the stream's state): ZGPU_CONT_HIST_WORDS words, bit j = the position max(cs->entry - 32512, the
 *              stream position of buf's first byte if that is later) + j is in the chains; in and out; all zero for a fresh stream, ones for a preset
 *              dictionary's positions but its last two.  Levels 4-9 do not look at it (may be NULL).
 *   excl       (levels 4-9) stream positions inside buf's history that are NOT in the hash chains: the two in front of every earlier flush point (zlib 1.2.3 never
 *              inserts them, qcsrc/deflate.c:1576 with lookahead < MIN_MATCH)
 *   out        whole bytes of the stream from the byte the last feed left unfinished (cs->bit_count bits of it are in cs->bit_value); res->out_bytes */
#define ZGPU_CONT_MORE 0
#define ZGPU_CONT_FLUSH 1
#define ZGPU_CONT_FINISH 2
#define ZGPU_CONT_CARRY_TOKENS 16384
typedef struct {
    uint64_t abs0;        /* in: stream position of buf[0] */
    uint64_t entry;       /* in/out: stream position the parse stands at with nothing in hand (deflate_slow between two matches) */
    uint64_t block_start; /* in/out: stream position of the first byte of the block that is filling */
    uint32_t carry_ntok;  /* in/out: tokens of that block so far (in carry_tok) */
    uint32_t bit_count, bit_value; /* in/out: bits of the stream's next byte that are decided already (0..7 of them) */
    uint32_t data_type;   /* in/out: 2 (Z_UNKNOWN) until the first block has decided */
    uint32_t first_block; /* in/out: 1 until the stream's first block is out */
    uint32_t last_eob;    /* in/out: last_eob_len (qcsrc/trees.c:1117) -- what _tr_align looks at; 8 for a fresh stream */
} zgpu_cont_state;
uint64_t zgpu_deflate_cont_bound(uint64_t buf_bytes); /* output capacity that is enough for one feed */
#define ZGPU_CONT_HIST_WORDS 1032
int zgpu_deflate_cont_host(zgpu_engine *e, const void *hist, uint64_t hist_bytes, const void *in, uint64_t in_bytes, uint64_t check_from, const zgpu_deflate_params *p,
                           int mode, zgpu_cont_state *cs, uint32_t *carry_tok, uint32_t *hist_bits, const uint64_t *excl, uint32_t nexcl, void *out, uint64_t out_cap,
                           zgpu_deflate_result *res);

/* Batch of independent small buffers: segment k = in[seg_offsets[k] .. seg_offsets[k+1]), each at most
 * 65536 bytes, becomes one chunk.  With ZGPU_F_FINAL every segment is a complete raw-deflate stream of its
 * own (Z_FINISH); without it every segment ends with the full-flush marker.  ZGPU_F_POS0_ALL applies to all
 * segments; ZGPU_F_ZLIB_WRAP is not allowed.  out_offsets (optional) receives nseg+1 output offsets. */
int zgpu_deflate_segments_device(zgpu_engine *e, const void *d_in, uint64_t in_bytes, const uint64_t *d_seg_offsets,
                                 uint64_t nseg, const zgpu_deflate_params *p, void *d_out, uint64_t out_cap,
                                 uint64_t *d_out_offsets, zgpu_deflate_result *res, void *hip_stream);
int zgpu_deflate_segments_host(zgpu_engine *e, const void *in, const uint64_t *seg_offsets, uint64_t nseg,
                               const zgpu_deflate_params *p, void *out, uint64_t out_cap, uint64_t *out_offsets,
                               zgpu_deflate_result *res);

/* One chunk with a preset dictionary (deflateSetDictionary, qcsrc/deflate.c:315-354).  `window` holds the dictionary bytes the
 * reference copies into its window -- the last min(length, 32506) bytes of the dictionary, at least 3 -- followed by the data;
 * window_bytes <= 65536.  Output: the raw deflate stream of the data alone, exactly what the reference's fresh stream emits
 * after deflateSetDictionary when it is finished (ZGPU_F_FINAL) or full-flushed; ZGPU_F_CRC32 as usual.  result.adler32 /
 * crc32 / data_type describe the data alone.  (One lane works on one chunk here: meant for the first chunk of a stream.) */
int zgpu_deflate_dict_chunk_host(zgpu_engine *e, const void *window, uint32_t window_bytes, uint32_t dict_bytes,
                                 const zgpu_deflate_params *p, void *out, uint64_t out_cap, zgpu_deflate_result *res);

/* ---- inflate ---- */
/* Segment k = d_in[offsets[k] .. offsets[k+1]) is a raw-deflate segment that decodes to at most
 * chunk_size bytes, written at d_out + k*chunk_size (every segment but the last must decode to exactly
 * chunk_size bytes).  chunk_size == 0 selects "compact" mode: segments of any size up to 65536 bytes are
 * decoded and their outputs concatenated (streams that were flushed in the middle of a chunk).  The last
 * segment must end with a final block; the others end with a stored empty block (flush marker).
 * chunk_size == ZGPU_WHOLE_STREAM with nchunks == 1: the one segment is a complete raw-deflate stream of any size
 * (< 512 MiB compressed, < 4 GiB decoded) that was not produced in chunks; one workgroup decodes it from end to end
 * (inflate.c:773-1076 + inffast.c:67-302 as they run for any stream).  When out_cap is too small the call returns
 * ZGPU_BUF_ERROR with res->out_bytes = the size the stream decodes to. */
#define ZGPU_WHOLE_STREAM 0xFFFFFFFFu
int zgpu_inflate_device(zgpu_engine *e, const void *d_in, uint64_t in_bytes, const uint64_t *d_chunk_offsets,
                        uint64_t nchunks, uint32_t chunk_size, void *d_out, uint64_t out_cap,
                        zgpu_inflate_result *res, void *hip_stream);
int zgpu_inflate_host(zgpu_engine *e, const void *in, uint64_t in_bytes, const uint64_t *chunk_offsets,
                      uint64_t nchunks, uint32_t chunk_size, void *out, uint64_t out_cap, zgpu_inflate_result *res);
/* Find the chunk boundaries of a mode-B stream that carries no side table: scans for the
 * 00 00 FF FF flush markers on the device and validates the split by decoding.
 * offsets (host, capacity max_chunks+1) receives the boundaries relative to in (raw body, no zlib
 * header).  Returns ZGPU_OK and *nchunks, or ZGPU_DATA_ERROR when no consistent split exists. */
int zgpu_inflate_find_chunks_host(zgpu_engine *e, const void *in, uint64_t in_bytes, uint32_t chunk_size,
                                  uint64_t *offsets, uint64_t max_chunks, uint64_t *nchunks);
/* The same search, delivering the decoded bytes: raw deflate body in (no zlib header / trailer), bytes out.  A body
 * that does not split into independent segments (any other producer's stream) is decoded in pieces found by search (see
 * zgpu_inflate_spec_count below) or, failing that, as ZGPU_WHOLE_STREAM. */
int zgpu_inflate_stream_host(zgpu_engine *e, const void *in, uint64_t in_bytes, void *out, uint64_t out_cap,
                             zgpu_inflate_result *res);
/* The same for the REST OF A STREAM (flags = ZGPU_INF_STREAM): `in` starts at a block boundary and may reach beyond the end of the
 * deflate data -- trailers, further members, anything.  The stream ends with its final block (res->stream_end, res->in_used =
 * offset of the first byte behind it); input that stops inside a block is not an error: the call delivers the full-flush
 * segments in front of the incomplete one (res->incomplete, res->in_used, res->out_bytes; all three can be 0) and is repeated from
 * in + in_used when more has arrived.  This is what inflate() of the zlib API needs (qcsrc/inflate.c:1114: DONE leaves the rest of
 * the input with the caller).  flags = 0 is zgpu_inflate_stream_host. */
#define ZGPU_INF_STREAM 1u
int zgpu_inflate_stream_host2(zgpu_engine *e, const void *in, uint64_t in_bytes, uint32_t flags, void *out, uint64_t out_cap,
                              zgpu_inflate_result *res);
/* The same for a stream that is taken up again INSIDE a byte: a stream of another producer that carries no flush markers is delivered piece by
 * piece as it arrives (qcsrc/inflate.c:323-371: the reference's inflate() hands output on with 32 KiB of window kept; here the caller keeps the
 * last 32 KiB it received and gives them to zgpu_inflate_set_dictionary before the next call).  A call whose input stops inside a block returns
 * the whole pieces in front of it (res->incomplete, res->out_bytes) and where they end: byte res->in_used, bit res->in_used_bits -- the next call
 * passes in + in_used and start_bit = in_used_bits.  start_bit 0 is zgpu_inflate_stream_host2. */
int zgpu_inflate_stream_host3(zgpu_engine *e, const void *in, uint64_t in_bytes, uint32_t start_bit, uint32_t flags, void *out, uint64_t out_cap,
                              zgpu_inflate_result *res);
/* A body that does not split at flush markers and holds at least 128 KiB (ZGPU_SPEC_MIN_BYTES) is decoded in pieces all the same: block starts
 * are searched behind every 1/4096 of the input (at least 32 KiB apart), every piece is decoded with the 32 KiB in front of it unknown,
 * the chain of pieces is checked and the unknowns filled in afterwards (zgpu_inflate.hip, spec_*).  A stream whose pieces do
 * not chain -- damaged, cut short, or one false block start -- goes through the one-workgroup decoder and gets its verdict.
 * Diagnostics: how many streams this process decoded in pieces / sent to the one-workgroup decoder. */
/* Test hook: the next launch of the fast position sort reports that its self-check failed (the LDS did not serve an atomic's lanes in
 * lane order), so that the engine's fallback -- redo the call with the ballot-ranked sort, and keep to it -- can be exercised. */
void zgpu_debug_inject_sort_fault(void);
uint64_t zgpu_inflate_spec_count(int which); /* 0: decoded in pieces, 1: one-workgroup decodes */
const char *zgpu_inflate_message(uint32_t index);
/* Preset dictionary of the inflate calls that follow (inflateSetDictionary, qcsrc/inflate.c:1200-1236): the first segment of a
 * call may reach back into its last min(len, 32768) bytes.  Stays set until replaced; len 0 clears it. */
int zgpu_inflate_set_dictionary(zgpu_engine *e, const void *dict, uint32_t len);
/* Which checks of the decoded bytes the inflate calls that follow compute into zgpu_inflate_result: ZGPU_CHECK_ADLER32 (what a zlib
 * stream's trailer holds, qcsrc/inflate.c:1083-1094), ZGPU_CHECK_CRC32 (a gzip member's), both (the default) or none (raw deflate:
 * inflate() keeps no check there, inflate.c:862-866).  A check that is not computed reads adler32 = 1 / crc32 = 0.  Each one is a pass
 * over the output (0.7 and 2.5 ms per 4 GiB).  Stays set until replaced. */
#define ZGPU_CHECK_ADLER32 1u
#define ZGPU_CHECK_CRC32 2u
int zgpu_inflate_set_checks(zgpu_engine *e, uint32_t mask);

/* ---- checksums (qcsrc/adler32.c:57-149) ---- */
int zgpu_adler32_device(zgpu_engine *e, const void *d_in, uint64_t in_bytes, uint32_t *adler_out, void *hip_stream);
/* CRC-32 as crc32() computes it (qcsrc/crc32.c:219-266), chunk CRCs joined like crc32_combine (crc32.c:370-423) */
int zgpu_crc32_device(zgpu_engine *e, const void *d_in, uint64_t in_bytes, uint32_t *crc_out, void *hip_stream);

/* ---- measurement support ---- */
/* Per-stage device time (HIP events on the launch stream), accumulated while profiling is on. */
enum {
    ZGPU_STAGE_CHAIN = 0, /* hash + static chain links */
    ZGPU_STAGE_MATCH,     /* longest-match search */
    ZGPU_STAGE_PARSE,     /* lazy/greedy parse -> tokens */
    ZGPU_STAGE_LZ_SERIAL, /* serial LZ77 (levels 1-3, or forced) */
    ZGPU_STAGE_HUFFMAN,   /* histogram + tree build + bit emit */
    ZGPU_STAGE_STITCH,    /* size scan + concatenation + Adler-32 */
    ZGPU_STAGE_INFLATE,
    ZGPU_STAGE_COUNT
};
void zgpu_profile_enable(zgpu_engine *e, int on);
void zgpu_profile_reset(zgpu_engine *e);
/* total milliseconds and launch count of one stage since the last reset */
int zgpu_profile_get(zgpu_engine *e, int stage, double *ms, uint64_t *launches);
const char *zgpu_stage_name(int stage);

/* Seeded synthetic corpora of SURVEY.md 8(d), generated in HBM (zlib_amd/csrc/corpus.h).
 * kind 0 silesia-mix, 1 log-text; d_out receives nchunks * 65536 bytes. */
int zgpu_corpus_fill_device(zgpu_engine *e, uint32_t kind, uint64_t seed, uint64_t first_chunk, uint64_t nchunks,
                            void *d_out, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif
