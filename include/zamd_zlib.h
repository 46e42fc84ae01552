/* zamd_zlib.h -- the zlib 1.2.3 stream API as served by libzamd_z.so (host C library over the HIP engine).
 *
 * This header is NOT a copy of the reference's h/zlib.h; it re-declares, in this project's own words, exactly the
 * binary interface that header defines, so that a program compiled against the reference's h/zlib.h
 * (/root/reference/h/zlib.h:82-101 z_stream, :162-205 constants, :1317-1342 entry points) links and runs against
 * libzamd_z.so unchanged.  tests/test_abi.py links the reference's own example.c (compiled against the reference's header)
 * with this library, tests/test_gpu_example.py runs it.
 *
 * What the library does differently from the reference, by design (BASELINE.json north_star):
 *   - deflate() output is the "mode B" stream: the input is cut into independent 64 KiB chunks, each compressed
 *     exactly as the reference compresses a fresh raw stream of that chunk, separated by full-flush markers.
 *     It is a valid RFC 1950 stream that any inflate() reads; it is not byte-identical to the reference's
 *     unchunked output, whose matches cross 64 KiB boundaries.
 *   - Served parameters: method Z_DEFLATED, windowBits 8..15 (zlib wrapper), -8..-15 (raw) or 24..31 (gzip wrapper; inflate also 40..47 =
 *     zlib or gzip, detected, and any window size 8..15 a header may declare), memLevel 1..9, all five strategies, levels 0..9 and
 *     Z_DEFAULT_COMPRESSION, deflateParams, deflateTune, deflateSetHeader / inflateGetHeader, deflateCopy / inflateCopy, preset
 *     dictionaries (set before the first input byte), inflateSync / inflateSyncPoint, inflatePrime (raw streams), the gz* file
 *     functions and inflateBack*.  A deflateInit2 geometry other than windowBits 15 / memLevel 8 changes the window the matches live in,
 *     the hash and the block cut of the reference's output: such streams are bit-exact too (every chunk = the reference's chunk function
 *     under that deflateInit2) and are served by the engine's general lane-per-chunk kernel, not by the kernels built for the default
 *     geometry (tests/golden/geometry_kat.json: all 63 geometries).  deflatePrime: up to 16 bits in front of the next chunk that is
 *     emitted (tests/golden/prime_kat.json); refused while input waits for its chunk to fill.
 *   - Z_SYNC_FLUSH and Z_PARTIAL_FLUSH end the pending chunk like Z_FULL_FLUSH: the marker is the same 00 00 FF FF, the chunk
 *     behind it simply does not refer back across it (a decoder cannot tell; the reference would keep its window).
 *   - inflate() hands out data per full-flush segment: output appears when a segment (or the stream) is complete, and a stream
 *     that was not produced in independent segments is decoded when its last byte has arrived.  Input behind the end of the
 *     stream is handed back (next_in / avail_in / total_in) as far as it came with the call that reached the end.
 *   - Exported but absent on purpose: nothing of the API.  The reference's internal globals (_tr_*, _dist_code, _length_code,
 *     inflate_fast, inflate_table, deflate_copyright, inflate_copyright) are implementation, not interface, and have no
 *     counterpart here; z_errmsg, zcalloc and zcfree are provided.
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

/* gzip header fields handed to deflateSetHeader / filled in by inflateGetHeader (RFC 1952; h/zlib.h:109-124) */
typedef struct gz_header_s {
    int text;        /* the data is believed to be text */
    uLong time;      /* modification time */
    int xflags;      /* extra flags (read only) */
    int os;          /* operating system */
    Bytef *extra;    /* extra field, or Z_NULL */
    uInt extra_len;  /* its length */
    uInt extra_max;  /* room at extra (reading) */
    Bytef *name;     /* zero-terminated file name, or Z_NULL */
    uInt name_max;   /* room at name (reading) */
    Bytef *comment;  /* zero-terminated comment, or Z_NULL */
    uInt comm_max;   /* room at comment (reading) */
    int hcrc;        /* a header CRC is / will be present */
    int done;        /* reading: 1 when the header has been read, -1 when the stream is not gzip */
} gz_header;
typedef gz_header *gz_headerp;
typedef voidp gzFile;
typedef unsigned (*in_func)(void *, unsigned char **);
typedef int (*out_func)(void *, unsigned char *, unsigned);

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
int deflateCopy(z_streamp dest, z_streamp source);
int deflateTune(z_streamp strm, int good_length, int max_lazy, int nice_length, int max_chain);
int deflatePrime(z_streamp strm, int bits, int value);
int deflateSetHeader(z_streamp strm, gz_headerp head);

int inflateInit_(z_streamp strm, const char *version, int stream_size);
int inflateInit2_(z_streamp strm, int windowBits, const char *version, int stream_size);
int inflate(z_streamp strm, int flush);
int inflateEnd(z_streamp strm);
int inflateReset(z_streamp strm);
int inflateSetDictionary(z_streamp strm, const Bytef *dictionary, uInt dictLength);
int inflateSync(z_streamp strm);
int inflateSyncPoint(z_streamp strm);
int inflateCopy(z_streamp dest, z_streamp source);
int inflatePrime(z_streamp strm, int bits, int value); /* raw streams, before the first input byte */
int inflateGetHeader(z_streamp strm, gz_headerp head);
int inflateBackInit_(z_streamp strm, int windowBits, unsigned char *window, const char *version, int stream_size);
int inflateBack(z_streamp strm, in_func in, void *in_desc, out_func out, void *out_desc);
int inflateBackEnd(z_streamp strm);

int compress(Bytef *dest, uLongf *destLen, const Bytef *source, uLong sourceLen);
int compress2(Bytef *dest, uLongf *destLen, const Bytef *source, uLong sourceLen, int level);
uLong compressBound(uLong sourceLen);
int uncompress(Bytef *dest, uLongf *destLen, const Bytef *source, uLong sourceLen);

uLong adler32(uLong adler, const Bytef *buf, uInt len);
uLong adler32_combine(uLong adler1, uLong adler2, z_off_t len2);
uLong crc32(uLong crc, const Bytef *buf, uInt len);
uLong crc32_combine(uLong crc1, uLong crc2, z_off_t len2);
const uLongf *get_crc_table(void);

/* gz* : .gz files through deflate() / inflate() of this library (qcsrc/gzio.c) */
gzFile gzopen(const char *path, const char *mode);
gzFile gzdopen(int fd, const char *mode);
int gzsetparams(gzFile file, int level, int strategy);
int gzread(gzFile file, voidp buf, unsigned len);
int gzwrite(gzFile file, const void *buf, unsigned len);
int gzprintf(gzFile file, const char *format, ...);
int gzputs(gzFile file, const char *s);
char *gzgets(gzFile file, char *buf, int len);
int gzputc(gzFile file, int c);
int gzgetc(gzFile file);
int gzungetc(int c, gzFile file);
int gzflush(gzFile file, int flush);
z_off_t gzseek(gzFile file, z_off_t offset, int whence);
int gzrewind(gzFile file);
z_off_t gztell(gzFile file);
int gzeof(gzFile file);
int gzdirect(gzFile file);
int gzclose(gzFile file);
const char *gzerror(gzFile file, int *errnum);
void gzclearerr(gzFile file);

#define deflateInit(strm, level) deflateInit_((strm), (level), ZLIB_VERSION, (int)sizeof(z_stream))
#define inflateInit(strm) inflateInit_((strm), ZLIB_VERSION, (int)sizeof(z_stream))
#define deflateInit2(strm, level, method, windowBits, memLevel, strategy) \
    deflateInit2_((strm), (level), (method), (windowBits), (memLevel), (strategy), ZLIB_VERSION, (int)sizeof(z_stream))
#define inflateInit2(strm, windowBits) inflateInit2_((strm), (windowBits), ZLIB_VERSION, (int)sizeof(z_stream))
#define inflateBackInit(strm, windowBits, window) inflateBackInit_((strm), (windowBits), (window), ZLIB_VERSION, (int)sizeof(z_stream))

#ifdef __cplusplus
}
#endif
#endif
