/* zamd_zlib.h -- the zlib 1.2.3 stream API as served by libzamd_z.so (host C library over the HIP engine).
 *
 * This header is NOT a copy of the reference's h/zlib.h; it re-declares, in this project's own words, exactly the
 * binary interface that header defines, so that a program compiled against the reference's h/zlib.h
 * (/root/reference/h/zlib.h:82-101 z_stream, :162-205 constants, :1317-1342 entry points) links and runs against
 * libzamd_z.so unchanged.  tests/test_host_abi.py compiles a client against the reference header and runs it
 * against this library to prove that.
 *
 * What the library does differently from the reference, by design (BASELINE.json north_star):
 *   - deflate() output is the "mode B" stream: the input is cut into independent 64 KiB chunks, each compressed
 *     exactly as the reference compresses a fresh raw stream of that chunk, separated by full-flush markers.
 *     It is a valid RFC 1950 stream that any inflate() reads; it is not byte-identical to the reference's
 *     unchunked output, whose matches cross 64 KiB boundaries.
 *   - Served parameters: method Z_DEFLATED, windowBits 15 (zlib wrapper), -15 (raw) or 31 (gzip wrapper with the
 *     default header; inflate also 47 = zlib or gzip, detected), memLevel 8, all five strategies, levels 0..9 and
 *     Z_DEFAULT_COMPRESSION, deflateParams, preset dictionaries (set before the first input byte).  Anything else
 *     returns Z_STREAM_ERROR (deflateSetHeader / inflateGetHeader, deflateTune/Prime/Copy are "next" rows of
 *     SURVEY.md 8f).
 *   - There is no CPU codec behind this API: without a usable GPU, the Init functions return Z_MEM_ERROR with
 *     strm->msg explaining why.
 */
#ifndef ZAMD_ZLIB_H
#define ZAMD_ZLIB_H
#ifdef __cplusplus
extern "C" {
#endif

#define ZLIB_VERSION "1.2.3"
#define ZLIB_VERNUM 0x1230

typedef unsigned char Byte;
typedef unsigned int uInt;   /* 32 bits */
typedef unsigned long uLong; /* 64 bits on LP64 */
typedef Byte Bytef;
typedef uLong uLongf;
typedef void *voidpf;
typedef void *voidp;
typedef long z_off_t;

typedef voidpf (*alloc_func)(voidpf opaque, uInt items, uInt size);
typedef void (*free_func)(voidpf opaque, voidpf address);

struct internal_state;

typedef struct z_stream_s {
    Bytef *next_in;   /* next input byte */
    uInt avail_in;    /* bytes available at next_in */
    uLong total_in;   /* input bytes read so far */
    Bytef *next_out;  /* where the next output byte goes */
    uInt avail_out;   /* free space at next_out */
    uLong total_out;  /* bytes output so far */
    char *msg;        /* last error text, or NULL */
    struct internal_state *state;
    alloc_func zalloc;
    free_func zfree;
    voidpf opaque;
    int data_type;    /* Z_BINARY / Z_TEXT guess after deflate */
    uLong adler;      /* Adler-32 of the uncompressed data */
    uLong reserved;
} z_stream;
typedef z_stream *z_streamp;

/* flush values */
#define Z_NO_FLUSH 0
#define Z_PARTIAL_FLUSH 1
#define Z_SYNC_FLUSH 2
#define Z_FULL_FLUSH 3
#define Z_FINISH 4
#define Z_BLOCK 5
/* return codes */
#define Z_OK 0
#define Z_STREAM_END 1
#define Z_NEED_DICT 2
#define Z_ERRNO (-1)
#define Z_STREAM_ERROR (-2)
#define Z_DATA_ERROR (-3)
#define Z_MEM_ERROR (-4)
#define Z_BUF_ERROR (-5)
#define Z_VERSION_ERROR (-6)
/* levels, strategies, data types */
#define Z_NO_COMPRESSION 0
#define Z_BEST_SPEED 1
#define Z_BEST_COMPRESSION 9
#define Z_DEFAULT_COMPRESSION (-1)
#define Z_FILTERED 1
#define Z_HUFFMAN_ONLY 2
#define Z_RLE 3
#define Z_FIXED 4
#define Z_DEFAULT_STRATEGY 0
#define Z_BINARY 0
#define Z_TEXT 1
#define Z_ASCII Z_TEXT
#define Z_UNKNOWN 2
#define Z_DEFLATED 8
#define Z_NULL 0

const char *zlibVersion(void);
uLong zlibCompileFlags(void);
const char *zError(int err);

int deflateInit_(z_streamp strm, int level, const char *version, int stream_size);
int deflateInit2_(z_streamp strm, int level, int method, int windowBits, int memLevel, int strategy, const char *version,
                  int stream_size);
int deflate(z_streamp strm, int flush);
int deflateEnd(z_streamp strm);
int deflateReset(z_streamp strm);
uLong deflateBound(z_streamp strm, uLong sourceLen);
int deflateSetDictionary(z_streamp strm, const Bytef *dictionary, uInt dictLength); /* before the first input byte */
int deflateParams(z_streamp strm, int level, int strategy);

int inflateInit_(z_streamp strm, const char *version, int stream_size);
int inflateInit2_(z_streamp strm, int windowBits, const char *version, int stream_size);
int inflate(z_streamp strm, int flush);
int inflateEnd(z_streamp strm);
int inflateReset(z_streamp strm);
int inflateSetDictionary(z_streamp strm, const Bytef *dictionary, uInt dictLength);

int compress(Bytef *dest, uLongf *destLen, const Bytef *source, uLong sourceLen);
int compress2(Bytef *dest, uLongf *destLen, const Bytef *source, uLong sourceLen, int level);
uLong compressBound(uLong sourceLen);
int uncompress(Bytef *dest, uLongf *destLen, const Bytef *source, uLong sourceLen);

uLong adler32(uLong adler, const Bytef *buf, uInt len);
uLong adler32_combine(uLong adler1, uLong adler2, z_off_t len2);
uLong crc32(uLong crc, const Bytef *buf, uInt len);
uLong crc32_combine(uLong crc1, uLong crc2, z_off_t len2);

#define deflateInit(strm, level) deflateInit_((strm), (level), ZLIB_VERSION, (int)sizeof(z_stream))
#define inflateInit(strm) inflateInit_((strm), ZLIB_VERSION, (int)sizeof(z_stream))
#define deflateInit2(strm, level, method, windowBits, memLevel, strategy) \
    deflateInit2_((strm), (level), (method), (windowBits), (memLevel), (strategy), ZLIB_VERSION, (int)sizeof(z_stream))
#define inflateInit2(strm, windowBits) inflateInit2_((strm), (windowBits), ZLIB_VERSION, (int)sizeof(z_stream))

#ifdef __cplusplus
}
#endif
#endif
