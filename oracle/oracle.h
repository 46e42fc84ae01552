/* oracle.h -- CPU restatement of the zlib-1.2.3 deflate/inflate hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load liboracle.so; the product (zlib_amd/) never links, imports or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_reference.py checks every function here byte-for-byte
 * against the real reference compiled from /root/reference (oracle/_ref/libzref.so) and against the
 * committed golden vectors in tests/golden/ (generated from that same compiled reference by
 * oracle/gen_golden.py).
 *
 * All citations are file:line under /root/reference.
 */
#ifndef ZAMD_ORACLE_H
#define ZAMD_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORA_CHUNK_MAX 65536u

/* Return codes follow h/zlib.h:170-178. */
#define ORA_OK 0
#define ORA_STREAM_END 1
#define ORA_NEED_DICT 2
#define ORA_STREAM_ERROR (-2)
#define ORA_DATA_ERROR (-3)
#define ORA_BUF_ERROR (-5)

/* One LZ77 token, as tallied by _tr_tally (qcsrc/trees.c:1022-1067): dist==0 -> literal byte lc,
 * otherwise a match of length lc+3 at distance dist. */
typedef struct {
    uint16_t dist;
    uint8_t lc;
    uint8_t pad;
} ora_token;

/* Per-chunk summary, for stage-wise comparison with the device pipeline. */
typedef struct {
    uint32_t ntokens; /* total tokens over all blocks of the chunk */
    uint32_t nblocks; /* deflate blocks emitted, not counting the flush marker */
    uint32_t btype[8]; /* 0 stored / 1 static / 2 dynamic, first 8 blocks */
    uint32_t data_type; /* Z_BINARY 0 / Z_TEXT 1 / Z_UNKNOWN 2 after the first non-empty block */
} ora_chunk_info;

/* F(bytes, level, pos0_matchable, is_last) of SURVEY.md section 8c: a fresh raw-deflate stream
 * (windowBits -15, memLevel 8, Z_DEFAULT_STRATEGY) over in[0..n), n <= 65536, terminated by Z_FINISH
 * (is_last) or Z_FULL_FLUSH (otherwise).  Returns bytes written, or 0 when cap is too small.
 * tokens (optional, capacity >= n) receives the token stream; info is optional. */
/* strategies, h/zlib.h:176-181 */
enum { ORA_DEFAULT_STRATEGY = 0, ORA_FILTERED = 1, ORA_HUFFMAN_ONLY = 2, ORA_RLE = 3, ORA_FIXED = 4 };

size_t ora_deflate_chunk_s(const uint8_t *in, size_t n, int level, int strategy, int pos0_matchable, int is_last,
                           uint8_t *out, size_t cap, ora_token *tokens, ora_chunk_info *info);
size_t ora_deflate_stream_s(const uint8_t *in, size_t n, int level, int strategy, size_t chunk_size, uint8_t *out, size_t cap);
size_t ora_deflate_chunk_d(const uint8_t *in, size_t n, size_t dict_len, int level, int strategy, int pos0_matchable, int is_last,
                           uint8_t *out, size_t cap, ora_token *tokens, ora_chunk_info *info);

size_t ora_deflate_chunk(const uint8_t *in, size_t n, int level, int pos0_matchable, int is_last,
                         uint8_t *out, size_t cap, ora_token *tokens, ora_chunk_info *info);

/* "Mode B" stream: 2-byte zlib header + F(chunk k) for every chunk + big-endian Adler-32.
 * chunk_size <= 65536.  Returns bytes written or 0 when cap is too small. */
size_t ora_deflate_stream(const uint8_t *in, size_t n, int level, size_t chunk_size, uint8_t *out,
                          size_t cap);

/* worst-case output of ora_deflate_stream (compress.c:75-79 bound per chunk, plus framing). */
size_t ora_deflate_bound(size_t n, size_t chunk_size);

/* The CONTINUOUS stream (deflate_oracle.c, last section): one raw deflate stream over dict + data, driven call by call -- call k hands over the data
 * up to offset cuts[k] with flush kinds[k] (0 none, 1 partial, 2 sync, 3 full), a last call the rest with Z_FINISH.  in[0..dict_len) = what
 * deflateSetDictionary puts into the window (<= 32506 bytes); n counts both.  What plain compress2() emits is ncuts == 0. */
size_t ora_deflate_cont(const uint8_t *in, size_t n, size_t dict_len, int level, int strategy, const uint32_t *cuts, const int32_t *kinds, size_t ncuts,
                        uint8_t *out, size_t cap);

size_t ora_deflate_cont_p(const uint8_t *in, size_t n, size_t dict_len, int level, int strategy, const uint32_t *cuts, const int32_t *kinds,
                          const int32_t *plevel, const int32_t *pstrategy, size_t ncuts, uint8_t *out, size_t cap);

/* adler32.c:57-125 and :128-149 */
uint32_t ora_adler32(uint32_t adler, const uint8_t *buf, size_t len);
uint32_t ora_adler32_combine(uint32_t adler1, uint32_t adler2, uint64_t len2);

/* crc32.c:219-251 (bitwise restatement) and :370-423 */
uint32_t ora_crc32(uint32_t crc, const uint8_t *buf, size_t len);
uint32_t ora_crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2);

/* Raw inflate of a complete deflate stream (inflate.c:554-1153 with wrap==0, inffast.c, inftrees.c).
 * Decodes until the final block ends.  *used / *produced report progress.  Returns ORA_STREAM_END on
 * success, ORA_DATA_ERROR with *msg set to the reference's message text on a malformed stream,
 * ORA_BUF_ERROR when input or output space runs out first. */
int ora_inflate_raw(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *used,
                    size_t *produced, const char **msg);

/* zlib-wrapped inflate: header check (inflate.c:589-632), raw body, Adler-32 trailer (:1077-1098). */
int ora_inflate_zlib(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *used,
                     size_t *produced, const char **msg);

#ifdef __cplusplus
}
#endif
#endif
