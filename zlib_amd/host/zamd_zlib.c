/* zamd_zlib.c -- zlib 1.2.3 stream API (include/zamd_zlib.h) on top of the HIP engine (include/zamd_gpu.h).
 *
 * Host side of the drop-in boundary, in C like the reference.  It restates the *API state machines* of
 *   deflate()   /root/reference/qcsrc/deflate.c:552-856   (argument checks, header, pending output, flush/finish rules, trailer)
 *   inflate()   /root/reference/qcsrc/inflate.c:554-1153  (header check, trailer check, return-code rules)
 *   compress2 / uncompress / compressBound   qcsrc/compress.c:22-79, qcsrc/uncompr.c:26-61
 *   adler32 / adler32_combine                qcsrc/adler32.c:57-149
 *   zError / zlibVersion / zcalloc / zcfree  qcsrc/zutil.c:14-30,133-137,300-316
 * and hands the codec work -- everything the reference does inside configuration_table[level].func() and inside
 * inflate_fast()/inflate_table() -- to the GPU through zgpu_deflate_host() / zgpu_inflate_stream_host().
 * There is no CPU codec in this file.
 *
 * deflate(): input is collected until at least one 64 KiB chunk is complete and has input behind it (or a flush / finish asks for
 * everything), those chunks go to the GPU in one call, the compressed bytes are handed out through next_out as space
 * allows.  inflate(): compressed input is collected; whenever the caller signals the end (Z_FINISH) or stops
 * supplying input, everything collected so far is decoded on the GPU and delivered through next_out.
 */
#include "../../include/zamd_zlib.h"
#include "../../include/zamd_gpu.h"
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHUNK 65536u
#define EXPORT __attribute__((visibility("default")))

/* ---- one engine per process (and per device), created on first use ---- */
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER; /* also serialises engine calls: the engine owns one stream + workspace */
static zgpu_engine *g_engine;
static char g_engine_err[256];

static zgpu_engine *engine_get(void)
{
    pthread_mutex_lock(&g_lock);
    if (!g_engine) {
        const char *dev = getenv("ZAMD_DEVICE");
        int rc = zgpu_engine_create(dev ? atoi(dev) : 0, &g_engine);
        if (rc != ZGPU_OK) {
            snprintf(g_engine_err, sizeof g_engine_err, "zamd: no usable MI355X (zgpu_engine_create rc=%d, %d device(s) visible); there is no CPU fallback",
                     rc, zgpu_device_count());
            g_engine = NULL;
        }
    }
    zgpu_engine *e = g_engine;
    pthread_mutex_unlock(&g_lock);
    return e;
}

/* ---- several streams at once: every engine call needs an engine to itself (one stream, one workspace).  ZAMD_ENGINES=n (1..8, default 1) lets up to n
 * calls of different threads run side by side on engines of their own (created when first needed, on the device of engine_get()); with the default
 * the calls take turns, as before. ---- */
#define ZAMD_MAX_ENGINES 8
static zgpu_engine *g_pool[ZAMD_MAX_ENGINES];
static int g_pool_busy[ZAMD_MAX_ENGINES], g_pool_made, g_pool_max;
static pthread_cond_t g_pool_cv = PTHREAD_COND_INITIALIZER;

static zgpu_engine *engine_checkout(void)
{
    zgpu_engine *first = engine_get();
    if (!first) return NULL;
    pthread_mutex_lock(&g_lock);
    if (g_pool_max == 0) {
        const char *v = getenv("ZAMD_ENGINES");
        long n = v ? strtol(v, NULL, 10) : 1;
        g_pool_max = n < 1 ? 1 : n > ZAMD_MAX_ENGINES ? ZAMD_MAX_ENGINES : (int)n;
        g_pool[0] = first; g_pool_made = 1;
    }
    for (;;) {
        for (int i = 0; i < g_pool_made; i++) if (!g_pool_busy[i]) { g_pool_busy[i] = 1; zgpu_engine *e = g_pool[i]; pthread_mutex_unlock(&g_lock); return e; }
        if (g_pool_made < g_pool_max) {
            const char *dev = getenv("ZAMD_DEVICE");
            zgpu_engine *e = NULL;
            if (zgpu_engine_create(dev ? atoi(dev) : 0, &e) == ZGPU_OK) { g_pool[g_pool_made] = e; g_pool_busy[g_pool_made++] = 1; pthread_mutex_unlock(&g_lock); return e; }
            g_pool_max = g_pool_made; /* (no room for another one: the ones there are take turns) */
        }
        pthread_cond_wait(&g_pool_cv, &g_lock);
    }
}
static void engine_checkin(zgpu_engine *e)
{
    pthread_mutex_lock(&g_lock);
    for (int i = 0; i < g_pool_made; i++) if (g_pool[i] == e) g_pool_busy[i] = 0;
    pthread_cond_signal(&g_pool_cv);
    pthread_mutex_unlock(&g_lock);
}

/* ---- more than one GPU: ZAMD_DEVICES="0,1,2,3" names the devices one deflate() / compress2() call may use.  Chunks are
 * independent, so a large input is cut into contiguous chunk ranges, one per device, each compressed by that device's engine from a thread
 * of its own (no exchange between the devices: the ranges' streams are laid end to end on the host, only the last one carries the final
 * block; Adler-32 / CRC-32 of the ranges are combined).  The first name is also the device of every other call. ---- */
#define ZAMD_MAX_DEVICES 16
static zgpu_engine *g_multi[ZAMD_MAX_DEVICES];
static pthread_mutex_t g_multi_lock[ZAMD_MAX_DEVICES];
static int g_multi_n = -1; /* -1: ZAMD_DEVICES not read yet */

static int multi_devices(void) /* engines ready for a fan-out (0: one device only) */
{
    pthread_mutex_lock(&g_lock);
    if (g_multi_n < 0) {
        g_multi_n = 0;
        const char *v = getenv("ZAMD_DEVICES");
        int ids[ZAMD_MAX_DEVICES], n = 0;
        while (v && *v && n < ZAMD_MAX_DEVICES) {
            char *end;
            long d = strtol(v, &end, 10);
            if (end == v) break;
            ids[n++] = (int)d;
            v = *end == ',' ? end + 1 : end;
        }
        if (n > 1) {
            int ok = 1;
            for (int i = 0; i < n && ok; i++) { ok = zgpu_engine_create(ids[i], &g_multi[i]) == ZGPU_OK; pthread_mutex_init(&g_multi_lock[i], NULL); }
            if (ok) g_multi_n = n;
            else for (int i = 0; i < n; i++) if (g_multi[i]) { zgpu_engine_destroy(g_multi[i]); g_multi[i] = NULL; }
        }
    }
    const int n = g_multi_n;
    pthread_mutex_unlock(&g_lock);
    return n;
}

struct multi_job { zgpu_engine *e; pthread_mutex_t *lock; const uint8_t *src; size_t n; zgpu_deflate_params p; uint8_t *dst; uint64_t cap; zgpu_deflate_result r; int rc;
                   int tuned; uint32_t tune[4]; int w_bits, mem_level; };
static void *multi_worker(void *arg)
{
    struct multi_job *j = (struct multi_job *)arg;
    pthread_mutex_lock(j->lock);
    zgpu_deflate_set_tuning(j->e, j->tuned, j->tune[0], j->tune[1], j->tune[2], j->tune[3]);
    zgpu_deflate_set_geometry(j->e, j->w_bits, j->mem_level);
    j->rc = zgpu_deflate_host(j->e, j->src, j->n, &j->p, j->dst, j->cap, NULL, &j->r);
    zgpu_deflate_set_geometry(j->e, 15, 8);
    zgpu_deflate_set_tuning(j->e, 0, 0, 0, 0, 0);
    pthread_mutex_unlock(j->lock);
    return NULL;
}

/* ---- small utilities ---- */
EXPORT const char *const z_errmsg[10] = {"need dictionary", "stream end", "", "file error", "stream error", "data error",
                                         "insufficient memory", "buffer error", "incompatible version", ""}; /* zutil.c:14-24 */
#define k_errmsg z_errmsg
#define ERR_MSG(code) ((char *)k_errmsg[2 - (code)])

EXPORT const char *zlibVersion(void) { return ZLIB_VERSION; }
EXPORT const char *zError(int err) { return k_errmsg[2 - err]; }
EXPORT uLong zlibCompileFlags(void) { return 0xa9; /* sizes of uInt/uLong/voidpf/z_off_t as the reference reports on LP64 */ }

EXPORT voidpf zcalloc(voidpf opaque, unsigned items, unsigned size) { (void)opaque; return malloc((size_t)items * size); } /* zutil.c:300-308 */
EXPORT void zcfree(voidpf opaque, voidpf p) { (void)opaque; free(p); }                                                   /* zutil.c:310-316 */
#define default_alloc zcalloc
#define default_free zcfree

typedef struct { uint8_t *p; size_t len, cap; } bytebuf;
static int buf_reserve(bytebuf *b, size_t extra)
{
    if (b->len + extra <= b->cap) return 1;
    size_t ncap = b->cap ? b->cap : 4096;
    while (ncap < b->len + extra) ncap += ncap / 2 + 4096;
    uint8_t *np = (uint8_t *)realloc(b->p, ncap);
    if (!np) return 0;
    b->p = np; b->cap = ncap;
    return 1;
}
static int buf_put(bytebuf *b, const void *src, size_t n) { if (!buf_reserve(b, n)) return 0; memcpy(b->p + b->len, src, n); b->len += n; return 1; }

/* adler32.c:57-125 */
#define ADLER_BASE 65521u
EXPORT uLong adler32(uLong adler, const Bytef *buf, uInt len)
{
    if (buf == Z_NULL) return 1L;
    uint32_t a = (uint32_t)adler & 0xffff, b = ((uint32_t)adler >> 16) & 0xffff;
    while (len) {
        uInt run = len < 5552 ? len : 5552;
        len -= run;
        while (run--) { a += *buf++; b += a; }
        a %= ADLER_BASE; b %= ADLER_BASE;
    }
    return a | ((uLong)b << 16);
}
/* adler32.c:128-149, including its quirk of comparing with > rather than >= */
EXPORT uLong adler32_combine(uLong adler1, uLong adler2, z_off_t len2)
{
    unsigned long sum1, sum2; unsigned rem = (unsigned)(len2 % ADLER_BASE);
    sum1 = adler1 & 0xffff; sum2 = (rem * sum1) % ADLER_BASE;
    sum1 += (adler2 & 0xffff) + ADLER_BASE - 1;
    sum2 += ((adler1 >> 16) & 0xffff) + ((adler2 >> 16) & 0xffff) + ADLER_BASE - rem;
    if (sum1 > ADLER_BASE) sum1 -= ADLER_BASE;
    if (sum1 > ADLER_BASE) sum1 -= ADLER_BASE;
    if (sum2 > (ADLER_BASE << 1)) sum2 -= (ADLER_BASE << 1);
    if (sum2 > ADLER_BASE) sum2 -= ADLER_BASE;
    return sum1 | (sum2 << 16);
}

/* Adler-32 of X||Y from the checksums of X and Y (the arithmetic adler32_combine is meant to do, without its quirk) */
static uint32_t adler_join(uint32_t x, uint32_t y, uint64_t leny)
{
    uint64_t rem = leny % ADLER_BASE, ax = x & 0xffff, bx = x >> 16, ay = y & 0xffff, by = y >> 16;
    uint64_t a = (ax + ay + ADLER_BASE - 1) % ADLER_BASE, b = (bx + by + rem * ((ax + ADLER_BASE - 1) % ADLER_BASE)) % ADLER_BASE;
    return (uint32_t)(a | (b << 16));
}

/* ---- CRC-32 (crc32.c:219-266 bytewise with one table; crc32_combine :370-423 as polynomial arithmetic mod P) ---- */
static uint32_t g_crc_table[256];
static uLong g_crc_table_ul[256]; /* the same values in the type get_crc_table() hands out */
static pthread_once_t g_crc_once = PTHREAD_ONCE_INIT;
static void crc_table_init(void)
{
    for (uint32_t i = 0; i < 256; i++) { uint32_t r = i; for (int k = 0; k < 8; k++) r = (r & 1u) ? (r >> 1) ^ 0xedb88320u : r >> 1; g_crc_table[i] = r; g_crc_table_ul[i] = r; }
}
EXPORT const uLongf *get_crc_table(void) { pthread_once(&g_crc_once, crc_table_init); return g_crc_table_ul; } /* crc32.c:205-213: the bytewise table */
EXPORT uLong crc32(uLong crc, const Bytef *buf, uInt len)
{
    if (buf == Z_NULL) return 0;
    pthread_once(&g_crc_once, crc_table_init);
    uint32_t c = (uint32_t)crc ^ 0xffffffffu;
    while (len--) c = g_crc_table[(c ^ *buf++) & 255u] ^ (c >> 8);
    return c ^ 0xffffffffu;
}
static uint32_t crc_mulmod(uint32_t a, uint32_t b) /* a(x) b(x) mod P(x), x^0 in bit 31 */
{
    uint32_t p = 0;
    for (uint32_t m = 0x80000000u; m; m >>= 1) { if (a & m) p ^= b; b = (b & 1u) ? (b >> 1) ^ 0xedb88320u : b >> 1; }
    return p;
}
static uint32_t crc_join(uint32_t cx, uint32_t cy, uint64_t leny) /* CRC of X||Y */
{
    uint32_t r = 0x80000000u, sq = 0x00800000u; /* x^0, x^8 */
    for (uint64_t n = leny; n; n >>= 1) { if (n & 1) r = crc_mulmod(sq, r); sq = crc_mulmod(sq, sq); }
    return crc_mulmod(r, cx) ^ cy;
}
EXPORT uLong crc32_combine(uLong crc1, uLong crc2, z_off_t len2)
{
    if (len2 == 0) return crc1; /* crc32.c:383-384 */
    return crc_join((uint32_t)crc1, (uint32_t)crc2, (uint64_t)len2);
}

/* ---- stream state ---- */
enum { KIND_DEFLATE = 0x5a44, KIND_INFLATE = 0x5a49 };
enum { ST_INIT = 1, ST_BUSY = 2, ST_FINISH = 3, ST_DONE = 4, ST_BAD = 5 };

enum { IN_HEAD = 1, IN_BODY = 2, IN_TRAIL = 3, IN_DONE = 4 }; /* inflate: where the stream stands */

struct internal_state {
    int kind, status, wrap, level, strategy, last_flush;
    bytebuf in;       /* deflate: input not yet compressed;  inflate: received bytes, consumed up to in_pos */
    bytebuf out;      /* produced bytes not yet handed to the caller */
    size_t out_pos;   /* first undelivered byte of out */
    int trailer_done; /* deflate: Adler trailer already appended */
    int any_block;    /* deflate: at least one chunk has been emitted */
    int w_bits, mem_level; /* deflateInit2's windowBits (9..15) and memLevel (1..9) */
    uint32_t dprime;  /* deflatePrime: (nbits << 16) | value, waiting for the next chunk that is emitted */
    uint32_t adler;   /* Adler-32 of the uncompressed data that went through the GPU (deflate) / was produced (inflate) */
    uint32_t crc;     /* the same for CRC-32 (gzip wrapper, wrap == 2) */
    bytebuf dict;     /* preset dictionary: deflate, the bytes the window receives until the first chunk is out; inflate, as set */
    int dict_pending; /* deflate: the first chunk has not been compressed yet and starts behind the dictionary */
    int need_dict, have_dict; /* inflate: the header asked for one / one has been set */
    uint32_t dictid;  /* Adler-32 of the dictionary (header field, deflate.c:646-649, inflate.c:623-627) */
    gz_headerp gzhead; /* deflateSetHeader / inflateGetHeader */
    int tuned; uint32_t tune[4]; /* deflateTune: good_length, max_lazy, nice_length, max_chain */
    /* deflate, ONE CONTINUOUS STREAM (the default at windowBits 15 / memLevel 8): what the reference's deflate() emits for the same calls, byte for byte --
     * include/zamd_gpu.h zgpu_deflate_cont_host.  The stream's state between two feeds of the engine lives here. */
    int cont;
    bytebuf win;          /* stream bytes from position win_abs0 on: what the parse can still reach (32512 bytes in front of cs.entry, the block that may
                             still be stored) and everything that has not been parsed yet */
    uint64_t win_abs0;
    zgpu_cont_state cs;
    uint32_t *carry;      /* the tokens of the block that is filling (ZGPU_CONT_CARRY_TOKENS words), then ZGPU_CONT_HIST_WORDS words: which positions of the
                             history are in the hash chains (levels 1-3) */
    uint64_t *excl; uint32_t nexcl, excl_cap; /* stream positions that are in no hash chain: the two in front of every flush point (and of the dictionary's end) */
    uint64_t floor_pos;   /* nothing in front of this position is history any more (Z_FULL_FLUSH: CLEAR_HASH, deflate.c:817) */
    uint64_t fed;         /* stream position behind the last byte received (dictionary bytes count) */
    uint64_t checked;     /* s->adler / s->crc cover the data up to here; tail_adler / tail_crc the bytes from here to fed */
    uint32_t tail_adler, tail_crc;
    int flush_done;       /* the flush value of the last call that went through the engine */
    uint64_t st_str, st_blk, st_off; /* level 0: deflate_stored's strstart, block_start and the window's first position (deflate.c:1390-1439, fill_window) */
    /* inflate */
    int mode;         /* IN_* */
    size_t in_pos;    /* first byte of `in` that has not been consumed */
    int gz;           /* the stream is a gzip member (wrap & 2 and the magic was there) */
    uint64_t produced; /* bytes decoded so far (ISIZE check) */
    size_t next_try;  /* do not try to decode again before this many unconsumed bytes have been collected */
    int pending_err;  /* a data error found behind output that has not been delivered yet: reported when it has */
    size_t last_piece; /* inflate: the largest piece of input a call has brought so far */
    unsigned skip_bits; /* inflate: bits of the byte at in_pos that belong to what has been decoded already (a stream without flush points is taken up
                           again where the last whole piece ended, which is a bit position) */
    const char *pending_msg;
    int no_partial;   /* keep everything from in_pos until the stream ends in one decode (inflatePrime) */
    int prime_bits; uint32_t prime_val; /* inflatePrime */
    int sync_have;    /* inflateSync: bytes of 00 00 FF FF matched so far */
    int at_marker;    /* the last decode consumed up to a flush marker (inflateSyncPoint) */
};

static uLong bound_for(uLong n)
{
    /* compress.c:75-79 gives n + n/4096 + n/16384 + 11; independent 64 KiB chunks of incompressible data cost up to 30 bytes
     * each (5 stored-block headers + the flush marker), more than that formula allows from the second chunk on */
    uLong ref = n + (n >> 12) + (n >> 14) + 11, chunks = n ? (n + CHUNK - 1) / CHUNK : 1, ours = n + 30 * chunks + 6 + 6;
    return ref > ours ? ref : ours;
}
static int chunks_mode(void) /* ZAMD_DEFLATE_CHUNKS=1: every stream is independent 64 KiB chunks (mode B of SURVEY.md 8c, rounds 1-3's behaviour) */
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("ZAMD_DEFLATE_CHUNKS"); v = e && *e && *e != '0'; }
    return v;
}
/* compress.c:75-79 */
EXPORT uLong compressBound(uLong sourceLen) { return chunks_mode() ? bound_for(sourceLen) : sourceLen + (sourceLen >> 12) + (sourceLen >> 14) + 11; }
EXPORT uLong deflateBound(z_streamp strm, uLong sourceLen)
{
    const int ours = strm != Z_NULL && strm->state != Z_NULL && strm->state->kind == KIND_DEFLATE, gz = ours && strm->state->wrap == 2;
    if (!ours || strm->state->cont) { /* deflate.c:489-511 */
        const uLong destLen = sourceLen + ((sourceLen + 7) >> 3) + ((sourceLen + 63) >> 6) + 11;
        if (!ours || strm->state->w_bits != 15) return destLen; /* (hash_bits is 15 at memLevel 8) */
        return sourceLen + (sourceLen >> 12) + (sourceLen >> 14) + 11; /* compressBound() */
    }
    if (ours && (strm->state->w_bits != 15 || strm->state->mem_level != 8)) /* short blocks, and blocks a small window cannot store (the arithmetic of deflate.c:513-515 per chunk) */
        return (uLong)zgpu_deflate_bound_geometry(sourceLen, CHUNK, strm->state->w_bits, strm->state->mem_level) + 18;
    return bound_for(sourceLen) + (gz ? 12 : 0); /* 18 bytes of gzip framing instead of the 6 of zlib (deflate.c:520-534) */
}

static struct internal_state *state_new(z_streamp strm, int kind)
{
    if (strm->zalloc == (alloc_func)0) { strm->zalloc = default_alloc; strm->opaque = (voidpf)0; }
    if (strm->zfree == (free_func)0) strm->zfree = default_free;
    struct internal_state *s = (struct internal_state *)strm->zalloc(strm->opaque, 1, (uInt)sizeof *s);
    if (!s) return NULL;
    memset(s, 0, sizeof *s);
    s->kind = kind;
    return s;
}
static void state_free(z_streamp strm)
{
    struct internal_state *s = strm->state;
    free(s->in.p); free(s->out.p); free(s->dict.p); free(s->win.p); free(s->carry); free(s->excl);
    strm->zfree(strm->opaque, s);
    strm->state = Z_NULL;
}

/* hand pending output to the caller (flush_pending, deflate.c:532-549) */
static void deliver(z_streamp strm)
{
    struct internal_state *s = strm->state;
    size_t n = s->out.len - s->out_pos;
    if (n > strm->avail_out) n = strm->avail_out;
    if (n == 0) return;
    memcpy(strm->next_out, s->out.p + s->out_pos, n);
    strm->next_out += n; strm->avail_out -= (uInt)n; strm->total_out += n; s->out_pos += n;
    if (s->out_pos == s->out.len) { s->out_pos = 0; s->out.len = 0; }
}

/* ======================================================================== deflate */
EXPORT int deflateInit2_(z_streamp strm, int level, int method, int windowBits, int memLevel, int strategy, const char *version, int stream_size)
{
    if (version == Z_NULL || version[0] != ZLIB_VERSION[0] || stream_size != (int)sizeof(z_stream)) return Z_VERSION_ERROR;
    if (strm == Z_NULL) return Z_STREAM_ERROR;
    strm->msg = Z_NULL;
    if (level == Z_DEFAULT_COMPRESSION) level = 6;
    int wrap = 1;
    if (windowBits < 0) { wrap = 0; windowBits = -windowBits; }
    else if (windowBits > 15) { wrap = 2; windowBits -= 16; } /* gzip wrapper, deflate.c:251-254 */
    if (method != Z_DEFLATED || windowBits < 8 || windowBits > 15 || memLevel < 1 || memLevel > 9 || strategy < 0 || strategy > Z_FIXED || level < 0 || level > 9) return Z_STREAM_ERROR;
    if (windowBits == 8) windowBits = 9; /* deflate.c:262: a 256-byte window is not served, the stream says 512 */
    if (!engine_get()) { strm->msg = g_engine_err; return Z_MEM_ERROR; }
    struct internal_state *s = state_new(strm, KIND_DEFLATE);
    if (!s) return Z_MEM_ERROR;
    strm->state = s;
    s->wrap = wrap; s->level = level; s->strategy = strategy; s->w_bits = windowBits; s->mem_level = memLevel;
    /* one continuous stream at the default geometry; another windowBits / memLevel is served in independent chunks.
     * (levels 1-3: the continuous deflate_fast is served since round 4 as well) */
    s->cont = !chunks_mode() && windowBits == 15 && memLevel == 8 && !(strategy == Z_RLE && level >= 1 && level <= 3); /* (deflate_fast with longest_match_fast: served in chunks) */
    if (s->cont) { s->carry = (uint32_t *)calloc((size_t)ZGPU_CONT_CARRY_TOKENS + ZGPU_CONT_HIST_WORDS + 8, 4); if (!s->carry) { state_free(strm); return Z_MEM_ERROR; } }
    return deflateReset(strm);
}
EXPORT int deflateInit_(z_streamp strm, int level, const char *version, int stream_size)
{
    return deflateInit2_(strm, level, Z_DEFLATED, 15, 8, Z_DEFAULT_STRATEGY, version, stream_size);
}
EXPORT int deflateReset(z_streamp strm)
{
    if (strm == Z_NULL || strm->state == Z_NULL || strm->state->kind != KIND_DEFLATE) return Z_STREAM_ERROR;
    struct internal_state *s = strm->state;
    strm->total_in = strm->total_out = 0; strm->msg = Z_NULL; strm->data_type = Z_UNKNOWN;
    s->in.len = 0; s->out.len = 0; s->out_pos = 0; s->trailer_done = 0; s->any_block = 0; s->dict.len = 0; s->dict_pending = 0; s->dprime = 0; /* (_tr_init, trees.c:401-402) */
    s->tuned = 0; /* lm_init: the level's own parameters again (deflate.c:380, 1009-1012) */
    s->status = s->wrap ? ST_INIT : ST_BUSY; s->last_flush = Z_NO_FLUSH;
    s->adler = 1; s->crc = 0; strm->adler = s->wrap == 2 ? 0 : 1; /* deflate.c:374-378 */
    s->win.len = 0; s->win_abs0 = 0; s->nexcl = 0; s->floor_pos = 0; s->fed = 0; s->checked = 0; s->tail_adler = 1; s->tail_crc = 0;
    s->st_str = s->st_blk = s->st_off = 0;
    if (s->carry) memset(s->carry + ZGPU_CONT_CARRY_TOKENS, 0, ((size_t)ZGPU_CONT_HIST_WORDS + 8) * 4);
    memset(&s->cs, 0, sizeof s->cs); s->cs.data_type = 2; s->cs.first_block = 1; s->cs.last_eob = 8; /* _tr_init, trees.c:382-406 */
    return Z_OK;
}
EXPORT int deflateEnd(z_streamp strm)
{
    if (strm == Z_NULL || strm->state == Z_NULL || strm->state->kind != KIND_DEFLATE) return Z_STREAM_ERROR;
    const int busy = strm->state->status == ST_BUSY;
    state_free(strm);
    return busy ? Z_DATA_ERROR : Z_OK; /* deflate.c:886 */
}
static int excl_add(struct internal_state *s, uint64_t pos);
static int cont_rewindow(struct internal_state *s, uint64_t lo, const uint8_t *in, size_t n);
static int cont_feed(z_streamp strm, const uint8_t *in, size_t n, int mode);
static int tail_put(struct internal_state *s, uint32_t value, int nbits);
/* deflate.c:315-354.  Served before the first byte of input (the reference also lets a raw stream replace its window later). */
EXPORT int deflateSetDictionary(z_streamp strm, const Bytef *d, uInt n)
{
    if (strm == Z_NULL || strm->state == Z_NULL || strm->state->kind != KIND_DEFLATE || d == Z_NULL) return Z_STREAM_ERROR;
    struct internal_state *s = strm->state;
    if (s->wrap == 2 || (s->wrap == 1 && s->status != ST_INIT) || s->any_block || s->in.len != 0 || strm->total_in != 0 || (s->cont && s->fed != 0)) return Z_STREAM_ERROR;
    if (s->wrap) strm->adler = adler32(strm->adler, d, n); /* becomes the DICTID of the header */
    if (n < 3) return Z_OK;                                /* shorter than MIN_MATCH: nothing to match against */
    const uInt max_dist = (1u << s->w_bits) - 262u, keep = n > max_dist ? max_dist : n; /* MAX_DIST: the tail of the dictionary (deflate.c:337-340) */
    if (s->cont) { /* the dictionary is the stream's first `keep` positions: window content, all of it in the hash chains but its last two bytes (deflate.c:341-351) */
        s->win.len = 0; s->win_abs0 = 0;
        if (!buf_put(&s->win, d + (n - keep), keep)) return Z_MEM_ERROR;
        s->fed = s->checked = keep; s->cs.entry = s->cs.block_start = keep; s->st_str = s->st_blk = keep; s->st_off = 0; s->nexcl = 0;
        if (!excl_add(s, keep - 2) || !excl_add(s, keep - 1)) return Z_MEM_ERROR;
        for (uInt j = 0; j + 2 < keep; j++) s->carry[ZGPU_CONT_CARRY_TOKENS + (j >> 5)] |= 1u << (j & 31u); /* levels 1-3: the same as bits */
        s->dict_pending = s->wrap ? 1 : 0; /* (only the header's PRESET_DICT flag and DICTID look at it) */
        return Z_OK;
    }
    s->dict.len = 0;
    if (!buf_put(&s->dict, d + (n - keep), keep)) return Z_MEM_ERROR;
    s->dict_pending = 1;
    return Z_OK;
}
static int run_chunks(z_streamp strm, const uint8_t *src, size_t n, int final);
/* deflate.c:416-451.  What was handed to deflate() so far is compressed with the old parameters (as one more run of chunks,
 * ending in a flush marker), what follows with the new ones. */
EXPORT int deflateParams(z_streamp strm, int level, int strategy)
{
    if (strm == Z_NULL || strm->state == Z_NULL || strm->state->kind != KIND_DEFLATE) return Z_STREAM_ERROR;
    struct internal_state *s = strm->state;
    if (level == Z_DEFAULT_COMPRESSION) level = 6;
    if (level < 0 || level > 9 || strategy < 0 || strategy > Z_FIXED) return Z_STREAM_ERROR;
    int rc = Z_OK;
    if (s->cont) {
        /* deflate.c:436-448: a change of the compress FUNCTION (stored / fast / slow) flushes what has been read with Z_PARTIAL_FLUSH first; other changes
         * take effect where the loop stands -- here: where the parse of the next feed begins, after what is waiting has been parsed as far as it goes */
        const int f_old = s->level == 0 ? 0 : s->level <= 3 ? 1 : 2, f_new = level == 0 ? 0 : level <= 3 ? 1 : 2;
        if (f_old != f_new && strm->total_in != 0) rc = deflate(strm, Z_PARTIAL_FLUSH);
        else if ((level != s->level || strategy != s->strategy) && s->level != 0 && s->fed - s->cs.entry > 1024 && s->status != ST_FINISH) rc = cont_feed(strm, NULL, 0, ZGPU_CONT_MORE);
        if (level >= 1 && level <= 3 && strategy == Z_RLE) return Z_STREAM_ERROR; /* (not served on a continuous stream) */
        if (rc == Z_OK && f_old == 1 && f_new == 2) {
            /* deflate_slow inherits deflate_fast's chains: the positions of the history that are NOT in them, as a list */
            const uint64_t w0 = s->cs.entry > 32512 ? s->cs.entry - 32512 : 0, base0 = w0 > s->win_abs0 ? w0 : s->win_abs0;
            s->nexcl = 0;
            for (uint64_t q = base0 > s->floor_pos ? base0 : s->floor_pos; q < s->cs.entry; q++) {
                const uint64_t j = q - base0;
                if (!((s->carry[ZGPU_CONT_CARRY_TOKENS + (j >> 5)] >> (j & 31u)) & 1u) && !excl_add(s, q)) return Z_MEM_ERROR;
            }
        } else if (rc == Z_OK && f_old == 2 && f_new == 1) {
            /* ... and deflate_fast deflate_slow's: every position of the history but the listed ones */
            const uint64_t w0 = s->cs.entry > 32512 ? s->cs.entry - 32512 : 0, base0 = w0 > s->win_abs0 ? w0 : s->win_abs0;
            memset(s->carry + ZGPU_CONT_CARRY_TOKENS, 0, (size_t)ZGPU_CONT_HIST_WORDS * 4);
            for (uint64_t q = base0 > s->floor_pos ? base0 : s->floor_pos; q < s->cs.entry; q++) { const uint64_t j = q - base0; s->carry[ZGPU_CONT_CARRY_TOKENS + (j >> 5)] |= 1u << (j & 31u); }
            for (uint32_t i = 0; i < s->nexcl; i++) if (s->excl[i] >= base0 && s->excl[i] < s->cs.entry) { const uint64_t j = s->excl[i] - base0; s->carry[ZGPU_CONT_CARRY_TOKENS + (j >> 5)] &= ~(1u << (j & 31u)); }
        }
        if (f_old != f_new) {
            if (f_new == 0) { s->st_str = s->st_blk = s->fed; s->st_off = s->fed >= 65275u ? ((s->fed - 65275u) / 32768u + 1u) * 32768u : 0; s->cs.entry = s->fed; }
            else if (f_old == 0) { s->cs.entry = s->cs.block_start = s->fed; s->floor_pos = s->fed; memset(s->carry + ZGPU_CONT_CARRY_TOKENS, 0, (size_t)ZGPU_CONT_HIST_WORDS * 4); if (!cont_rewindow(s, s->fed, NULL, 0)) return Z_MEM_ERROR; }
        }
        if (s->level != level) s->tuned = 0;
        s->level = level; s->strategy = strategy;
        return rc;
    }
    if ((level != s->level || strategy != s->strategy) && s->in.len != 0 && s->status != ST_FINISH) {
        rc = run_chunks(strm, s->in.p, s->in.len, 0);
        s->in.len = 0;
    }
    if (s->level != level) s->tuned = 0; /* deflate.c:443-448 */
    s->level = level; s->strategy = strategy;
    return rc;
}

/* deflate.c:453-470 */
EXPORT int deflateTune(z_streamp strm, int good_length, int max_lazy, int nice_length, int max_chain)
{
    if (strm == Z_NULL || strm->state == Z_NULL || strm->state->kind != KIND_DEFLATE) return Z_STREAM_ERROR;
    struct internal_state *s = strm->state;
    s->tuned = 1; s->tune[0] = (uint32_t)good_length; s->tune[1] = (uint32_t)max_lazy; s->tune[2] = (uint32_t)nice_length; s->tune[3] = (uint32_t)max_chain;
    /* deflate_fast with parameters of the caller's is served by the lane-per-chunk loop, i.e. in independent chunks: a stream that has not begun goes there
     * (one that has keeps its parameters' level row: the engine refuses the feed) */
    if (s->cont && s->level >= 1 && s->level <= 3 && s->fed == 0 && !s->any_block && s->out.len == s->out_pos) { s->cont = 0; }
    return Z_OK;
}
/* deflate.c:404-413: bi_valid = bits, bi_buf = value's low bits -- the next block starts behind them.  The chunk streams end at byte boundaries
 * (full flush), so the bits go in front of the next chunk that is emitted; the engine starts that chunk's first block inside the byte and the
 * padding of its stored blocks and of its end moves with it, as in the reference.  Input that is still waiting for its chunk to fill would
 * come out BEHIND the bits here but has, in the reference, partly been emitted already: that case is refused. */
EXPORT int deflatePrime(z_streamp strm, int bits, int value)
{
    if (strm == Z_NULL || strm->state == Z_NULL || strm->state->kind != KIND_DEFLATE) return Z_STREAM_ERROR;
    struct internal_state *s = strm->state;
    if (bits < 0 || bits > 16 || s->in.len != 0 || s->status == ST_FINISH) return Z_STREAM_ERROR;
    if (s->cont) { /* bi_valid = bits, bi_buf = value: whole bytes of them go out with the next output, the rest waits in the unfinished byte */
        if (s->fed != s->cs.entry) return Z_STREAM_ERROR; /* (input that is waiting would have been partly emitted in front of the bits by the reference) */
        /* (the bits wait for the next deflate() call: a stream's header, which that call may still have to write, goes in front of them) */
        s->dprime = ((uint32_t)bits << 16) | (bits ? (uint32_t)value & ((1u << bits) - 1u) : 0u) | 0x80000000u;
        return Z_OK;
    }
    s->dprime = bits ? ((uint32_t)bits << 16) | ((uint32_t)value & ((1u << bits) - 1u)) : 0u;
    return Z_OK;
}
/* deflate.c:393-401 */
EXPORT int deflateSetHeader(z_streamp strm, gz_headerp head)
{
    if (strm == Z_NULL || strm->state == Z_NULL || strm->state->kind != KIND_DEFLATE || strm->state->wrap != 2) return Z_STREAM_ERROR;
    strm->state->gzhead = head;
    return Z_OK;
}
static int buf_dup(bytebuf *d, const bytebuf *src) { d->p = NULL; d->len = d->cap = 0; return src->len == 0 || buf_put(d, src->p, src->len); }
/* deflate.c:894-947 / inflate.c:1323-1368: an independent copy of the stream with everything it holds */
static int state_copy(z_streamp dest, z_streamp source, int kind)
{
    if (source == Z_NULL || dest == Z_NULL || source->state == Z_NULL || source->state->kind != kind) return Z_STREAM_ERROR;
    const struct internal_state *ss = source->state;
    *dest = *source;
    struct internal_state *ds = (struct internal_state *)dest->zalloc(dest->opaque, 1, (uInt)sizeof *ds);
    if (!ds) return Z_MEM_ERROR;
    *ds = *ss;
    dest->state = ds;
    ds->carry = NULL; ds->excl = NULL; ds->excl_cap = 0;
    int ok = buf_dup(&ds->in, &ss->in) && buf_dup(&ds->out, &ss->out) && buf_dup(&ds->dict, &ss->dict) && buf_dup(&ds->win, &ss->win);
    if (ok && ss->carry) { const size_t cb = ((size_t)ZGPU_CONT_CARRY_TOKENS + ZGPU_CONT_HIST_WORDS + 8) * 4; ds->carry = (uint32_t *)malloc(cb); ok = ds->carry != NULL; if (ok) memcpy(ds->carry, ss->carry, cb); }
    if (ok && ss->nexcl) { ds->excl = (uint64_t *)malloc((size_t)ss->nexcl * sizeof(uint64_t)); ok = ds->excl != NULL; if (ok) { memcpy(ds->excl, ss->excl, (size_t)ss->nexcl * sizeof(uint64_t)); ds->excl_cap = ss->nexcl; } }
    if (!ok) {
        free(ds->in.p); free(ds->out.p); free(ds->dict.p); free(ds->win.p); free(ds->carry); free(ds->excl); dest->zfree(dest->opaque, ds); dest->state = Z_NULL;
        return Z_MEM_ERROR;
    }
    return Z_OK;
}
EXPORT int deflateCopy(z_streamp dest, z_streamp source) { return state_copy(dest, source, KIND_DEFLATE); }

/* level 0 needs no match finder or entropy coder: stored blocks are framing.  One chunk = the bytes the reference's
 * deflate_stored emits for a fresh stream of that chunk (deflate.c:1390-1439): blocks of at most 65531 bytes, the rest,
 * then the flush marker or, on the last chunk, the final bit. */
static int stored_block(bytebuf *out, const uint8_t *src, size_t len, int last, uint32_t *prime)
{
    /* the 3 header bits behind whatever deflatePrime left in the bit buffer, then bi_windup (trees.c:869-877, 1184-1204) */
    const uint32_t pb = *prime >> 16, bits = (*prime & 0xffffu) | ((uint32_t)(last ? 1 : 0) << pb), nh = (pb + 3 + 7) / 8;
    const uint8_t h[7] = {(uint8_t)bits, (uint8_t)(bits >> 8), (uint8_t)(bits >> 16)};
    const uint8_t l[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)~len, (uint8_t)(~len >> 8)};
    *prime = 0;
    return buf_put(out, h, nh) && buf_put(out, l, 4) && buf_put(out, src, len);
}
/* One chunk as deflate_stored cuts it (deflate.c:1390-1439) when the whole chunk is at hand and the output has room: the window of 2 << w_bits bytes is
 * filled (fill_window, :1265-1350: sliding by a window when strstart has reached wsize + MAX_DIST), everything buffered joins the block, a block is closed
 * when it reaches max_block_size = min(0xffff, pending_buf_size - 5) and again when it reaches MAX_DIST, and the flush at the end closes whatever is left --
 * an EMPTY block after a cut that took everything.  `dict`: the window starts with that many dictionary bytes (deflateSetDictionary, :347-348). */
static int stored_one_chunk(bytebuf *out, const uint8_t *src, size_t len, int last, uint32_t *prime, int w_bits, int mem_level, size_t dict)
{
    const long W = 1L << w_bits, MAXD = W - 262, pend = 4L << (mem_level + 6), maxblk = pend - 5 < 0xffff ? pend - 5 : 0xffff;
    long str = (long)dict, blk = (long)dict, look = 0, avail = (long)len, slid = 0; /* window indices; data offset of index i = i - dict + slid */
    for (;;) {
        if (look <= 1) {
            do {
                long more = 2 * W - look - str;
                if (str >= W + MAXD) { str -= W; blk -= W; slid += W; more += W; }
                if (avail == 0) break;
                const long take = avail < more ? avail : more;
                avail -= take; look += take;
            } while (look < 262 && avail != 0);
            if (look == 0) break;
        }
        str += look; look = 0;
        const long max_start = blk + maxblk;
        if (str == 0 || str >= max_start) {
            look = str - max_start; str = max_start;
            if (!stored_block(out, src + (blk - (long)dict + slid), (size_t)(str - blk), 0, prime)) return 0;
            blk = str;
        }
        if (str - blk >= MAXD) {
            if (!stored_block(out, src + (blk - (long)dict + slid), (size_t)(str - blk), 0, prime)) return 0;
            blk = str;
        }
    }
    return stored_block(out, src + (blk - (long)dict + slid), (size_t)(str - blk), last, prime);
}
static int stored_chunks(bytebuf *out, const uint8_t *src, size_t n, int final, uint32_t *prime, int w_bits, int mem_level, size_t dict)
{
    size_t nchunks = n ? (n + CHUNK - 1) / CHUNK : 1;
    for (size_t k = 0; k < nchunks; k++) {
        const size_t lo = k * CHUNK, len = n - lo < CHUNK ? n - lo : CHUNK;
        const int last = final && k + 1 == nchunks;
        if (!stored_one_chunk(out, src + lo, len, last, prime, w_bits, mem_level, k == 0 ? dict : 0)) return 0;
        if (!last) { static const uint8_t marker[5] = {0, 0, 0, 0xff, 0xff}; if (!buf_put(out, marker, 5)) return 0; }
    }
    return 1;
}


/* ======================================================================== deflate: ONE CONTINUOUS STREAM
 * What the reference's deflate() emits for the same sequence of calls (deflate.c:552-856 with deflate_stored / deflate_fast / deflate_slow behind it),
 * byte for byte, whatever the size of the input: the engine parses the stream in feeds (zgpu_deflate_cont_host, include/zamd_gpu.h), this file keeps
 * what lies between two feeds -- the bytes the parse can still reach, the tokens of the block that is filling, the bits of the unfinished byte -- and
 * writes what is framing: flush markers (trees.c:867-879, 892-915), stored blocks of level 0 (deflate.c:1390-1439), header and trailer.
 * Where the reference parses while the input trickles in, this library collects: a Z_NO_FLUSH call hands its bytes to the engine once ZAMD_FEED_BYTES
 * (default 16 MiB) of unparsed input have gathered; flushes and Z_FINISH hand over everything.  The stream's bytes do not depend on that.
 * Not modelled: deflateParams() between two levels of the same compress function while unflushed input is waiting (the reference switches the
 * parameters at the position its loop happens to stand at, deflate.c:436-448; here they change where the parse of the next feed begins), and a
 * Z_NO_FLUSH slice that ends 5 .. 261 bytes behind a position 32768 k + 65274 whose first chain candidate lies exactly 32506 bytes back (the
 * reference's window slides one loop iteration earlier there). */
static int tail_put(struct internal_state *s, uint32_t value, int nbits) /* send_bits + bi_flush (trees.c:217-229, 1161-1173): whole bytes go out, fewer than 8 bits wait */
{
    uint64_t v = s->cs.bit_value | ((uint64_t)value << s->cs.bit_count);
    int n = (int)s->cs.bit_count + nbits;
    while (n >= 8) { const uint8_t b = (uint8_t)v; if (!buf_put(&s->out, &b, 1)) return 0; v >>= 8; n -= 8; }
    s->cs.bit_count = (uint32_t)n; s->cs.bit_value = (uint32_t)v;
    return 1;
}
static int tail_align(struct internal_state *s) /* bi_windup, trees.c:1178-1191 */
{
    if (s->cs.bit_count) { const uint8_t b = (uint8_t)s->cs.bit_value; if (!buf_put(&s->out, &b, 1)) return 0; }
    s->cs.bit_count = 0; s->cs.bit_value = 0;
    return 1;
}
static int cont_stored_block(struct internal_state *s, const uint8_t *src, size_t len, int last) /* _tr_stored_block, trees.c:867-879 */
{
    const uint8_t l[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)~len, (uint8_t)(~len >> 8)};
    s->cs.last_eob = 8;
    return tail_put(s, last ? 1u : 0u, 3) && tail_align(s) && buf_put(&s->out, l, 4) && buf_put(&s->out, src, len);
}
static int cont_marker(struct internal_state *s, int flush) /* deflate.c:808-819 */
{
    if (flush == Z_PARTIAL_FLUSH) { /* _tr_align, trees.c:892-915: one empty static block, two when the decoder's lookahead could fall short */
        if (!tail_put(s, 2, 3) || !tail_put(s, 0, 7)) return 0;
        if (1 + (int)s->cs.last_eob + 10 - (int)s->cs.bit_count < 9 && (!tail_put(s, 2, 3) || !tail_put(s, 0, 7))) return 0;
        s->cs.last_eob = 7;
        return 1;
    }
    return cont_stored_block(s, NULL, 0, 0);
}
static int excl_add(struct internal_state *s, uint64_t pos)
{
    if (s->nexcl && s->excl[s->nexcl - 1] >= pos) return 1; /* (ascending, no duplicates: flush points only move forward) */
    if (s->nexcl == s->excl_cap) {
        const uint32_t nc = s->excl_cap ? s->excl_cap * 2 : 64;
        uint64_t *q = (uint64_t *)realloc(s->excl, (size_t)nc * sizeof *q);
        if (!q) return 0;
        s->excl = q; s->excl_cap = nc;
    }
    s->excl[s->nexcl++] = pos;
    return 1;
}
/* the window after a feed (or an append): bytes [lo, end) of (win followed by in) */
static int cont_rewindow(struct internal_state *s, uint64_t lo, const uint8_t *in, size_t n)
{
    const uint64_t old_end = s->win_abs0 + s->win.len;
    if (lo < s->win_abs0) lo = s->win_abs0;
    if (lo >= old_end) { /* everything that stays comes from `in` */
        const size_t skip = (size_t)(lo - old_end);
        s->win.len = 0; s->win_abs0 = lo;
        return skip >= n || buf_put(&s->win, in + skip, n - skip);
    }
    const size_t drop = (size_t)(lo - s->win_abs0);
    if (drop) { memmove(s->win.p, s->win.p + drop, s->win.len - drop); s->win.len -= drop; s->win_abs0 = lo; }
    return n == 0 || buf_put(&s->win, in, n);
}
static void cont_checks_done(z_streamp strm) /* strm->adler covers every byte received (deflate.c:968-970 applies it slice by slice) */
{
    struct internal_state *s = strm->state;
    const uint64_t tail = s->fed - s->checked;
    if (s->wrap == 2) strm->adler = tail ? crc_join(s->crc, s->tail_crc, tail) : s->crc;
    else if (s->wrap) strm->adler = tail ? adler_join(s->adler, s->tail_adler, tail) : s->adler;
}
/* one feed of the engine: the window so far + `in` (n bytes that have not been appended to the window) */
static int cont_feed(z_streamp strm, const uint8_t *in, size_t n, int mode)
{
    struct internal_state *s = strm->state;
    if (!engine_get()) { strm->msg = g_engine_err; return Z_MEM_ERROR; }
    const uint64_t cap = zgpu_deflate_cont_bound(s->win.len + n) + 64;
    if (!buf_reserve(&s->out, cap)) return Z_MEM_ERROR;
    zgpu_deflate_params p = {s->level, 0, s->wrap == 2 ? ZGPU_F_CRC32 : 0u, ZGPU_LZ_AUTO, s->strategy, 0};
    zgpu_deflate_result r;
    s->cs.abs0 = s->win_abs0;
    zgpu_engine *e = engine_checkout();
    zgpu_deflate_set_tuning(e, s->tuned, s->tune[0], s->tune[1], s->tune[2], s->tune[3]);
    const int rc = zgpu_deflate_cont_host(e, s->win.p, s->win.len, in, n, s->checked - s->win_abs0, &p, mode, &s->cs, s->carry, s->carry + ZGPU_CONT_CARRY_TOKENS, s->excl, s->nexcl, s->out.p + s->out.len, cap, &r);
    zgpu_deflate_set_tuning(e, 0, 0, 0, 0, 0);
    engine_checkin(e);
    if (rc != ZGPU_OK) { strm->msg = (char *)zgpu_engine_error(e); return rc == ZGPU_MEM_ERROR ? Z_MEM_ERROR : Z_STREAM_ERROR; }
    s->out.len += r.out_bytes;
    const uint64_t new_fed = s->fed + n, nck = new_fed - s->checked;
    if (nck) { s->adler = adler_join(s->adler, r.adler32, nck); if (s->wrap == 2) s->crc = crc_join(s->crc, r.crc32, nck); }
    s->checked = new_fed; s->tail_adler = 1; s->tail_crc = 0;
    if (s->cs.data_type != 2) strm->data_type = (int)s->cs.data_type; /* the first block has decided (trees.c:934-935) */
    s->any_block = 1;
    /* what the next feed can still reach: 32512 bytes in front of the parse (MAX_DIST and the engine's tiles), and the block that is filling while it may
     * still be stored (its first byte must not have left the reference's window: at most 64 KiB + a game in front of where the parse stands) */
    uint64_t lo = s->cs.entry > 32512 ? s->cs.entry - 32512 : 0;
    if (lo < s->floor_pos) lo = s->floor_pos;
    if (s->cs.entry - s->cs.block_start <= 65536 + 512 && s->cs.block_start < lo) lo = s->cs.block_start;
    if (!cont_rewindow(s, lo, in, n)) return Z_MEM_ERROR;
    s->fed = new_fed;
    uint32_t k = 0;
    for (uint32_t i = 0; i < s->nexcl; i++) if (s->excl[i] >= s->win_abs0) s->excl[k++] = s->excl[i];
    s->nexcl = k;
    return Z_OK;
}
/* level 0: deflate_stored (deflate.c:1390-1439) over the window, with fill_window's arithmetic (deflate.c:1266-1358) -- which bytes end up in which block
 * depends on how the input arrives, so it runs in every call.  The window holds everything from st_blk on. */
static int cont_stored(z_streamp strm, int flush)
{
    struct internal_state *s = strm->state;
    const uint64_t W = 32768, MAXD = W - 262, max_block = 65536 - 5; /* min(0xffff, pending_buf_size - 5) at memLevel 8 */
    uint64_t filled = s->cs.entry; /* strstart + lookahead (kept in cs.entry between calls: a cut at max_block leaves lookahead behind) */
    for (;;) {
        if (filled - s->st_str <= 1) {
            do { /* fill_window */
                if (s->st_str - s->st_off >= W + MAXD) s->st_off += W;
                const uint64_t more = 2 * W - (filled - s->st_str) - (s->st_str - s->st_off);
                if (filled == s->fed) break;
                filled += s->fed - filled < more ? s->fed - filled : more;
            } while (filled - s->st_str < 262 && filled != s->fed);
            if (filled == s->st_str && flush == Z_NO_FLUSH) break;
            if (filled == s->st_str) goto closing;
        }
        s->st_str = filled;
        const uint64_t max_start = s->st_blk + max_block;
        if (s->st_str >= max_start) {
            s->st_str = max_start;
            if (!cont_stored_block(s, s->win.p + (s->st_blk - s->win_abs0), (size_t)(s->st_str - s->st_blk), 0)) return Z_MEM_ERROR;
            s->st_blk = s->st_str;
        }
        if (s->st_str - s->st_blk >= MAXD) {
            if (!cont_stored_block(s, s->win.p + (s->st_blk - s->win_abs0), (size_t)(s->st_str - s->st_blk), 0)) return Z_MEM_ERROR;
            s->st_blk = s->st_str;
        }
    }
    s->cs.entry = filled;
    return Z_OK;
closing:
    if (!cont_stored_block(s, s->win.p + (s->st_blk - s->win_abs0), (size_t)(s->st_str - s->st_blk), flush == Z_FINISH)) return Z_MEM_ERROR;
    s->st_blk = s->st_str; s->cs.entry = filled;
    return Z_OK;
}
/* the body of deflate() for a continuous stream: the caller's input is taken (read_buf, deflate.c:956-981), whatever the call makes final is appended to s->out */
static int cont_deflate(z_streamp strm, int flush)
{
    struct internal_state *s = strm->state;
    const uint8_t *src = strm->next_in; const size_t n = strm->avail_in;
    strm->next_in += n; strm->total_in += n; strm->avail_in = 0;
    if (s->dprime) { /* deflatePrime: bi_valid = bits, bi_buf = value (deflate.c:411-412) */
        const uint32_t pb = (s->dprime >> 16) & 0x1fu, pv = s->dprime & 0xffffu;
        s->dprime = 0; s->cs.bit_count = 0; s->cs.bit_value = 0;
        if (pb && !tail_put(s, pv, (int)pb)) return Z_MEM_ERROR;
    }
    long feed_min = 16l << 20;
    { const char *v = getenv("ZAMD_FEED_BYTES"); if (v && *v) feed_min = strtol(v, NULL, 10); if (feed_min < 1024) feed_min = 1024; }
    if (s->level == 0) {
        if (n) {
            if (!buf_put(&s->win, src, n)) return Z_MEM_ERROR;
            if (s->wrap == 2) s->crc = (uint32_t)crc32(s->crc, src, (uInt)n); else s->adler = (uint32_t)adler32(s->adler, src, (uInt)n);
            s->fed += n; s->checked = s->fed;
        }
        int rc = cont_stored(strm, flush);
        if (rc != Z_OK) return rc;
        s->any_block = 1;
        if (flush != Z_NO_FLUSH && flush != Z_FINISH && !cont_marker(s, flush)) return Z_MEM_ERROR;
        if (flush == Z_FINISH && !tail_align(s)) return Z_MEM_ERROR;
        if (flush == Z_FULL_FLUSH) s->floor_pos = s->fed;
        if (!cont_rewindow(s, s->st_blk, NULL, 0)) return Z_MEM_ERROR; /* (the block that is filling is all that is needed) */
        s->cs.block_start = s->st_blk;
        cont_checks_done(strm);
        return Z_OK;
    }
    const uint64_t unparsed = s->fed + n - s->cs.entry;
    if (flush == Z_NO_FLUSH) {
        if (unparsed < (uint64_t)feed_min) { /* collect */
            if (n) {
                if (!buf_put(&s->win, src, n)) return Z_MEM_ERROR;
                if (s->wrap == 2) s->tail_crc = (uint32_t)crc32(s->tail_crc, src, (uInt)n); else s->tail_adler = (uint32_t)adler32(s->tail_adler, src, (uInt)n);
                s->fed += n;
            }
            cont_checks_done(strm);
            return Z_OK;
        }
        const int rc = cont_feed(strm, src, n, ZGPU_CONT_MORE);
        cont_checks_done(strm);
        return rc;
    }
    int rc = cont_feed(strm, src, n, flush == Z_FINISH ? ZGPU_CONT_FINISH : ZGPU_CONT_FLUSH);
    if (rc != Z_OK) return rc;
    if (flush != Z_FINISH) {
        if (!cont_marker(s, flush)) return Z_MEM_ERROR;
        if (flush == Z_FULL_FLUSH) { s->floor_pos = s->fed; s->nexcl = 0; memset(s->carry + ZGPU_CONT_CARRY_TOKENS, 0, (size_t)ZGPU_CONT_HIST_WORDS * 4); if (!cont_rewindow(s, s->fed, NULL, 0)) return Z_MEM_ERROR; }
        else if ((s->fed >= 2 && !excl_add(s, s->fed - 2)) || (s->fed >= 1 && !excl_add(s, s->fed - 1))) return Z_MEM_ERROR; /* never inserted: lookahead < MIN_MATCH there (deflate.c:1576) */
    }
    cont_checks_done(strm);
    return Z_OK;
}

/* compress `n` bytes (whole chunks, or everything when a flush / finish asks for it) and append the result to s->out */
static int run_chunks(z_streamp strm, const uint8_t *src, size_t n, int final)
{
    struct internal_state *s = strm->state;
    if (s->dict_pending) {
        /* The first chunk shares the 64 KiB window with the dictionary: it takes 65536 - |dictionary| bytes (or all there is) and
         * goes through the engine's dictionary entry point; whatever follows is ordinary chunks.  Stored blocks (level 0) come out
         * the same with or without a dictionary. */
        const size_t room = CHUNK - s->dict.len, take = n < room ? n : room;
        const int first_final = final && take == n;
        s->dict_pending = 0;
        if (s->level == 0) {
            if (!stored_chunks(&s->out, src, take, first_final, &s->dprime, s->w_bits, s->mem_level, s->dict.len)) return Z_MEM_ERROR;
            s->adler = (uint32_t)adler32(s->adler, src, (uInt)take);
        } else {
            if (!engine_get()) { strm->msg = g_engine_err; return Z_MEM_ERROR; }
            bytebuf w = {0};
            if (!buf_put(&w, s->dict.p, s->dict.len) || !buf_put(&w, src, take)) { free(w.p); return Z_MEM_ERROR; }
            const uint64_t cap = zgpu_deflate_bound_geometry(w.len, CHUNK, s->w_bits, s->mem_level);
            if (!buf_reserve(&s->out, cap)) { free(w.p); return Z_MEM_ERROR; }
            zgpu_deflate_params p = {s->level, CHUNK, first_final ? ZGPU_F_FINAL : 0u, ZGPU_LZ_AUTO, s->strategy, s->dprime};
            s->dprime = 0;
            zgpu_deflate_result r;
            zgpu_engine *e = engine_checkout();
            zgpu_deflate_set_tuning(e, s->tuned, s->tune[0], s->tune[1], s->tune[2], s->tune[3]);
            zgpu_deflate_set_geometry(e, s->w_bits, s->mem_level);
            int rc = zgpu_deflate_dict_chunk_host(e, w.p, (uint32_t)w.len, (uint32_t)s->dict.len, &p, s->out.p + s->out.len, cap, &r);
            zgpu_deflate_set_geometry(e, 15, 8);
            zgpu_deflate_set_tuning(e, 0, 0, 0, 0, 0);
            engine_checkin(e);
            free(w.p);
            if (rc != ZGPU_OK) { strm->msg = (char *)zgpu_engine_error(e); return rc == ZGPU_MEM_ERROR ? Z_MEM_ERROR : Z_STREAM_ERROR; }
            s->out.len += r.out_bytes;
            s->adler = adler_join(s->adler, r.adler32, take);
            if (take > 0) strm->data_type = (int)r.data_type;
        }
        s->any_block = 1;
        s->dict.len = 0;
        if (take == n) return Z_OK;
        src += take; n -= take;
    }
    if (s->level == 0) {
        if (!stored_chunks(&s->out, src, n, final, &s->dprime, s->w_bits, s->mem_level, 0)) return Z_MEM_ERROR;
        for (size_t o = 0; o < n; o += 0x40000000u) {
            size_t m = n - o < 0x40000000u ? n - o : 0x40000000u;
            if (s->wrap == 2) s->crc = (uint32_t)crc32(s->crc, src + o, (uInt)m); else s->adler = (uint32_t)adler32(s->adler, src + o, (uInt)m);
        }
        s->any_block = 1; return Z_OK;
    }
    zgpu_engine *e = engine_get();
    if (!e) { strm->msg = g_engine_err; return Z_MEM_ERROR; }
    uint64_t cap = zgpu_deflate_bound_geometry(n, CHUNK, s->w_bits, s->mem_level);
    const int ndev = n >= ((size_t)64 << 20) ? multi_devices() : 0; /* (a fan-out pays from 32 MiB per device on) */
    if (ndev > 1) {
        struct multi_job job[ZAMD_MAX_DEVICES];
        pthread_t th[ZAMD_MAX_DEVICES];
        const size_t nchunks = (n + CHUNK - 1) / CHUNK;
        int used = ndev;
        while (used > 1 && nchunks / (size_t)used < 512) used--;
        if (!buf_reserve(&s->out, cap + (uint64_t)used * 64)) return Z_MEM_ERROR;
        uint64_t at = 0;
        for (int d = 0; d < used; d++) {
            const size_t c0 = nchunks * (size_t)d / (size_t)used, c1 = nchunks * (size_t)(d + 1) / (size_t)used;
            const size_t b0 = c0 * CHUNK, b1 = d + 1 == used ? n : c1 * CHUNK;
            struct multi_job *j = &job[d];
            j->e = g_multi[d]; j->lock = &g_multi_lock[d]; j->src = src + b0; j->n = b1 - b0;
            j->p = (zgpu_deflate_params){s->level, CHUNK, ((final && d + 1 == used) ? ZGPU_F_FINAL : 0u) | (s->wrap == 2 ? ZGPU_F_CRC32 : 0u), ZGPU_LZ_AUTO, s->strategy, d == 0 ? s->dprime : 0u};
            j->cap = zgpu_deflate_bound_geometry(j->n, CHUNK, s->w_bits, s->mem_level); j->dst = s->out.p + s->out.len + at; at += j->cap;
            j->tuned = s->tuned; memcpy(j->tune, s->tune, sizeof j->tune); j->w_bits = s->w_bits; j->mem_level = s->mem_level; j->rc = ZGPU_ERRNO;
        }
        s->dprime = 0;
        int started = 0;
        for (; started < used; started++) if (pthread_create(&th[started], NULL, multi_worker, &job[started]) != 0) break;
        for (int d = 0; d < started; d++) pthread_join(th[d], NULL);
        for (int d = started; d < used; d++) multi_worker(&job[d]); /* (no thread to be had: the range is compressed here) */
        uint64_t w = 0;
        for (int d = 0; d < used; d++) {
            struct multi_job *j = &job[d];
            if (j->rc != ZGPU_OK) { strm->msg = (char *)zgpu_engine_error(j->e); return j->rc == ZGPU_MEM_ERROR ? Z_MEM_ERROR : Z_STREAM_ERROR; }
            memmove(s->out.p + s->out.len + w, j->dst, j->r.out_bytes); /* the ranges' streams end to end */
            w += j->r.out_bytes;
            s->adler = adler_join(s->adler, j->r.adler32, j->n);
            if (s->wrap == 2) s->crc = crc_join(s->crc, j->r.crc32, j->n);
        }
        s->out.len += w;
        if (!s->any_block && n > 0) strm->data_type = (int)job[0].r.data_type;
        s->any_block = 1;
        return Z_OK;
    }
    if (!buf_reserve(&s->out, cap)) return Z_MEM_ERROR;
    zgpu_deflate_params p = {s->level, CHUNK, (final ? ZGPU_F_FINAL : 0u) | (s->wrap == 2 ? ZGPU_F_CRC32 : 0u), ZGPU_LZ_AUTO, s->strategy, s->dprime};
    s->dprime = 0;
    zgpu_deflate_result r;
    e = engine_checkout();
    zgpu_deflate_set_tuning(e, s->tuned, s->tune[0], s->tune[1], s->tune[2], s->tune[3]);
    zgpu_deflate_set_geometry(e, s->w_bits, s->mem_level);
    int rc = zgpu_deflate_host(e, src, n, &p, s->out.p + s->out.len, cap, NULL, &r);
    zgpu_deflate_set_geometry(e, 15, 8);
    zgpu_deflate_set_tuning(e, 0, 0, 0, 0, 0);
    engine_checkin(e);
    if (rc != ZGPU_OK) { strm->msg = (char *)zgpu_engine_error(e); return rc == ZGPU_MEM_ERROR ? Z_MEM_ERROR : Z_STREAM_ERROR; }
    s->out.len += r.out_bytes;
    s->adler = adler_join(s->adler, r.adler32, n); /* computed on the GPU with the chunks (deflate.c:968-970) */
    if (s->wrap == 2) s->crc = crc_join(s->crc, r.crc32, n);
    if (!s->any_block && n > 0) strm->data_type = (int)r.data_type; /* the first block decides (trees.c:934-935) */
    s->any_block = 1;
    return Z_OK;
}

EXPORT int deflate(z_streamp strm, int flush)
{
    if (strm == Z_NULL || strm->state == Z_NULL || strm->state->kind != KIND_DEFLATE || flush > Z_FINISH || flush < 0) return Z_STREAM_ERROR;
    struct internal_state *s = strm->state;
    if (strm->next_out == Z_NULL || (strm->next_in == Z_NULL && strm->avail_in != 0) || (s->status == ST_FINISH && flush != Z_FINISH)) {
        strm->msg = ERR_MSG(Z_STREAM_ERROR); return Z_STREAM_ERROR;
    }
    if (strm->avail_out == 0) { strm->msg = ERR_MSG(Z_BUF_ERROR); return Z_BUF_ERROR; }
    int old_flush = s->last_flush;
    s->last_flush = flush;

    if (s->status == ST_INIT && s->wrap == 2) { /* gzip header, deflate.c:578-621 and the EXTRA/NAME/COMMENT/HCRC states :660-754; OS_CODE 3 as the reference builds on this host */
        const uint8_t xfl = (uint8_t)(s->level == 9 ? 2 : (s->strategy >= Z_HUFFMAN_ONLY || s->level < 2) ? 4 : 0);
        const size_t h0 = s->out.len;
        const gz_headerp g = s->gzhead;
        if (g == Z_NULL) {
            const uint8_t h[10] = {31, 139, 8, 0, 0, 0, 0, 0, xfl, 3};
            if (!buf_put(&s->out, h, 10)) return Z_MEM_ERROR;
        } else {
            const uint8_t h[10] = {31, 139, 8, (uint8_t)((g->text ? 1 : 0) + (g->hcrc ? 2 : 0) + (g->extra == Z_NULL ? 0 : 4) + (g->name == Z_NULL ? 0 : 8) + (g->comment == Z_NULL ? 0 : 16)),
                                   (uint8_t)g->time, (uint8_t)(g->time >> 8), (uint8_t)(g->time >> 16), (uint8_t)(g->time >> 24), xfl, (uint8_t)g->os};
            int ok = buf_put(&s->out, h, 10);
            if (ok && g->extra != Z_NULL) { const uint8_t xl[2] = {(uint8_t)g->extra_len, (uint8_t)(g->extra_len >> 8)}; ok = buf_put(&s->out, xl, 2) && buf_put(&s->out, g->extra, g->extra_len & 0xffffu); }
            if (ok && g->name != Z_NULL) ok = buf_put(&s->out, g->name, strlen((const char *)g->name) + 1);
            if (ok && g->comment != Z_NULL) ok = buf_put(&s->out, g->comment, strlen((const char *)g->comment) + 1);
            if (ok && g->hcrc) { const uint32_t hc = (uint32_t)crc32(0, s->out.p + h0, (uInt)(s->out.len - h0)); const uint8_t c2[2] = {(uint8_t)hc, (uint8_t)(hc >> 8)}; ok = buf_put(&s->out, c2, 2); }
            if (!ok) return Z_MEM_ERROR;
        }
        s->status = ST_BUSY;
    }
    if (s->status == ST_INIT) { /* zlib header, deflate.c:625-649 */
        unsigned hdr = (Z_DEFLATED + ((unsigned)(s->w_bits - 8) << 4)) << 8, lf = (s->strategy >= Z_HUFFMAN_ONLY || s->level < 2) ? 0 : s->level < 6 ? 1 : s->level == 6 ? 2 : 3;
        hdr |= lf << 6;
        if (s->dict_pending) hdr |= 0x20; /* PRESET_DICT, deflate.c:641 */
        hdr += 31 - hdr % 31;
        uint8_t h[6] = {(uint8_t)(hdr >> 8), (uint8_t)hdr, (uint8_t)(strm->adler >> 24), (uint8_t)(strm->adler >> 16), (uint8_t)(strm->adler >> 8), (uint8_t)strm->adler};
        if (!buf_put(&s->out, h, s->dict_pending ? 6 : 2)) return Z_MEM_ERROR;
        strm->adler = 1; /* deflate.c:650 */
        if (s->cont) s->dict_pending = 0;
        s->status = ST_BUSY;
    }
    if (s->out.len - s->out_pos != 0) { /* deflate.c:757-768 */
        deliver(strm);
        if (strm->avail_out == 0) { s->last_flush = -1; return Z_OK; }
    } else if (strm->avail_in == 0 && flush <= old_flush && flush != Z_FINISH) { /* deflate.c:774-777 */
        strm->msg = ERR_MSG(Z_BUF_ERROR); return Z_BUF_ERROR;
    }
    if (s->status == ST_FINISH && strm->avail_in != 0) { strm->msg = ERR_MSG(Z_BUF_ERROR); return Z_BUF_ERROR; }

    /* (a call that only comes back for the rest of a flush's output -- the call before ran out of avail_out -- does not flush again.  The reference
     * writes one more empty block and marker in that case, deflate.c:795-806: its bytes depend on the caller's output space, these do not.) */
    if (s->cont && (strm->avail_in != 0 || (flush != Z_NO_FLUSH && s->status != ST_FINISH && !(old_flush == -1 && s->flush_done == flush)))) {
        const int rc = cont_deflate(strm, flush);
        if (rc != Z_OK) return rc;
        s->flush_done = flush;
        if (flush == Z_FINISH) s->status = ST_FINISH;
        deliver(strm);
        if (strm->avail_out == 0 && (s->out.len - s->out_pos != 0 || flush != Z_FINISH)) { s->last_flush = -1; return Z_OK; }
    } else if (!s->cont && (strm->avail_in != 0 || s->in.len != 0 || (flush != Z_NO_FLUSH && s->status != ST_FINISH))) {
        /* take the caller's input (read_buf, deflate.c:956-981) */
        const uint8_t *src = strm->next_in; size_t n = strm->avail_in;
        strm->next_in += n; strm->total_in += n; strm->avail_in = 0;
        int rc = Z_OK;
        if (flush == Z_NO_FLUSH && s->dict_pending) {
            /* the first chunk behind a dictionary is 65536 - |dictionary| bytes: collect until it is complete */
            const size_t room = CHUNK - s->dict.len;
            if (!buf_put(&s->in, src, n)) rc = Z_MEM_ERROR;
            else if (s->in.len > room) {
                rc = run_chunks(strm, s->in.p, room, 0);
                memmove(s->in.p, s->in.p + room, s->in.len - room); s->in.len -= room;
                if (rc == Z_OK && s->in.len > CHUNK) {
                    size_t whole = (s->in.len - 1) / CHUNK * CHUNK;
                    rc = run_chunks(strm, s->in.p, whole, 0);
                    memmove(s->in.p, s->in.p + whole, s->in.len - whole); s->in.len -= whole;
                }
            }
        } else if (flush == Z_NO_FLUSH) {
            /* only chunks with input behind them are compressed now: the tail waits for more input, and so does a chunk that is complete
             * but may turn out to be the last one (it then carries the final bit instead of a flush marker and an empty final block --
             * what the reference driven chunk by chunk writes, however the caller slices its input) */
            if (s->in.len == 0 && n > CHUNK) { size_t whole = (n - 1) / CHUNK * CHUNK; rc = run_chunks(strm, src, whole, 0); src += whole; n -= whole; }
            if (rc == Z_OK && n) { if (!buf_put(&s->in, src, n)) rc = Z_MEM_ERROR; }
            if (rc == Z_OK && s->in.len > CHUNK) {
                size_t whole = (s->in.len - 1) / CHUNK * CHUNK;
                rc = run_chunks(strm, s->in.p, whole, 0);
                memmove(s->in.p, s->in.p + whole, s->in.len - whole); s->in.len -= whole;
            }
        } else {
            /* a flush or the end: everything seen so far becomes decodable output */
            const int final = flush == Z_FINISH;
            if (s->in.len == 0) rc = run_chunks(strm, src, n, final);
            else { if (!buf_put(&s->in, src, n)) rc = Z_MEM_ERROR; else { rc = run_chunks(strm, s->in.p, s->in.len, final); s->in.len = 0; } }
            if (final) s->status = ST_FINISH;
        }
        if (rc != Z_OK) return rc;
        /* strm->adler covers every byte read so far: what the GPU has checksummed plus the (< 64 KiB) buffered tail */
        if (s->wrap == 2) strm->adler = s->in.len ? crc_join(s->crc, (uint32_t)crc32(0, s->in.p, (uInt)s->in.len), s->in.len) : s->crc;
        else if (s->wrap) strm->adler = s->in.len ? adler_join(s->adler, (uint32_t)adler32(1, s->in.p, (uInt)s->in.len), s->in.len) : s->adler;
        deliver(strm);
        if (strm->avail_out == 0 && (s->out.len - s->out_pos != 0 || flush != Z_FINISH)) { s->last_flush = -1; return Z_OK; }
    }
    if (flush != Z_FINISH) return Z_OK;
    if (s->out.len - s->out_pos != 0) { s->last_flush = -1; return Z_OK; }
    if (!s->wrap || s->trailer_done) return Z_STREAM_END;
    if (s->wrap == 2) { /* trailer, deflate.c:833-843: CRC-32 and total_in, least significant byte first */
        const uint32_t c = s->crc, l = (uint32_t)strm->total_in;
        uint8_t t8[8] = {(uint8_t)c, (uint8_t)(c >> 8), (uint8_t)(c >> 16), (uint8_t)(c >> 24), (uint8_t)l, (uint8_t)(l >> 8), (uint8_t)(l >> 16), (uint8_t)(l >> 24)};
        if (!buf_put(&s->out, t8, 8)) return Z_MEM_ERROR;
        s->trailer_done = 1;
        deliver(strm);
        return s->out.len - s->out_pos != 0 ? Z_OK : Z_STREAM_END;
    }
    /* trailer, deflate.c:847-855 */
    uint8_t t[4] = {(uint8_t)(s->adler >> 24), (uint8_t)(s->adler >> 16), (uint8_t)(s->adler >> 8), (uint8_t)s->adler};
    if (!buf_put(&s->out, t, 4)) return Z_MEM_ERROR;
    s->trailer_done = 1;
    deliver(strm);
    return s->out.len - s->out_pos != 0 ? Z_OK : Z_STREAM_END;
}

/* ======================================================================== inflate */
EXPORT int inflateInit2_(z_streamp strm, int windowBits, const char *version, int stream_size)
{
    if (version == Z_NULL || version[0] != ZLIB_VERSION[0] || stream_size != (int)sizeof(z_stream)) return Z_VERSION_ERROR;
    if (strm == Z_NULL) return Z_STREAM_ERROR;
    strm->msg = Z_NULL;
    int wrap = 1;
    if (windowBits < 0) { wrap = 0; windowBits = -windowBits; }
    else if (windowBits > 15) { wrap = (windowBits >> 4) + 1; windowBits &= 15; } /* inflate.c:158-164: 2 gzip only, 3 zlib or gzip */
    if (windowBits < 8 || windowBits > 15 || wrap > 3) return Z_STREAM_ERROR;
#ifndef ZAMD_SAN_NO_ENGINE /* (the sanitizer build of oracle/Makefile runs the header parsers where there is no GPU: there the engine is asked for when the body starts) */
    if (!engine_get()) { strm->msg = g_engine_err; return Z_MEM_ERROR; }
#endif
    struct internal_state *s = state_new(strm, KIND_INFLATE);
    if (!s) return Z_MEM_ERROR;
    strm->state = s;
    s->wrap = wrap; s->level = windowBits; /* level field reused: the window size the header may declare */
    return inflateReset(strm);
}
EXPORT int inflateInit_(z_streamp strm, const char *version, int stream_size) { return inflateInit2_(strm, 15, version, stream_size); }
EXPORT int inflateReset(z_streamp strm)
{
    if (strm == Z_NULL || strm->state == Z_NULL || strm->state->kind != KIND_INFLATE) return Z_STREAM_ERROR;
    struct internal_state *s = strm->state;
    strm->total_in = strm->total_out = 0; strm->msg = Z_NULL; strm->adler = 1;
    s->in.len = 0; s->in_pos = 0; s->out.len = 0; s->out_pos = 0; s->status = ST_BUSY; s->adler = 1; s->crc = 0; s->next_try = 0; s->dict.len = 0;
    s->need_dict = 0; s->have_dict = 0; s->mode = s->wrap ? IN_HEAD : IN_BODY; s->gz = 0; s->produced = 0; s->pending_err = 0; s->pending_msg = NULL;
    s->no_partial = 0; s->prime_bits = 0; s->prime_val = 0; s->sync_have = 0; s->at_marker = 0; s->gzhead = Z_NULL; s->last_piece = 0; s->skip_bits = 0;
    return Z_OK;
}
EXPORT int inflateEnd(z_streamp strm)
{
    if (strm == Z_NULL || strm->state == Z_NULL || strm->state->kind != KIND_INFLATE) return Z_STREAM_ERROR;
    state_free(strm);
    return Z_OK;
}
/* inflate.c:1200-1236 */
EXPORT int inflateSetDictionary(z_streamp strm, const Bytef *d, uInt n)
{
    if (strm == Z_NULL || strm->state == Z_NULL || strm->state->kind != KIND_INFLATE || d == Z_NULL) return Z_STREAM_ERROR;
    struct internal_state *s = strm->state;
    if (s->wrap != 0 && !s->need_dict) return Z_STREAM_ERROR;
    if (s->need_dict && (uint32_t)adler32(1, d, n) != s->dictid) return Z_DATA_ERROR;
    const uInt keep = n > 32768u ? 32768u : n; /* the window keeps the tail */
    s->dict.len = 0;
    if (keep && !buf_put(&s->dict, d + (n - keep), keep)) return Z_MEM_ERROR;
    s->have_dict = 1; s->need_dict = 0;
    return Z_OK;
}
/* inflate.c:1305-1321 */
EXPORT int inflateGetHeader(z_streamp strm, gz_headerp head)
{
    if (strm == Z_NULL || strm->state == Z_NULL || strm->state->kind != KIND_INFLATE || (strm->state->wrap & 2) == 0) return Z_STREAM_ERROR;
    strm->state->gzhead = head;
    head->done = 0;
    return Z_OK;
}
EXPORT int inflateCopy(z_streamp dest, z_streamp source) { return state_copy(dest, source, KIND_INFLATE); }
/* inflate.c:128-142: bits in front of the first input byte.  Served for raw streams before any input (what it exists for: a
 * deflate stream that starts inside a byte); such a stream is decoded in one piece when its end has arrived. */
EXPORT int inflatePrime(z_streamp strm, int bits, int value)
{
    if (strm == Z_NULL || strm->state == Z_NULL || strm->state->kind != KIND_INFLATE) return Z_STREAM_ERROR;
    struct internal_state *s = strm->state;
    if (bits < 0 || bits > 16 || s->prime_bits + bits > 32) return Z_STREAM_ERROR;
    if (bits == 0) return Z_OK;
    if (s->wrap != 0 || s->in.len != 0 || s->produced != 0 || s->prime_bits + bits > 24) return Z_STREAM_ERROR;
    s->prime_val |= ((uint32_t)value & ((1u << bits) - 1)) << s->prime_bits;
    s->prime_bits += bits; s->no_partial = 1;
    return Z_OK;
}

#define IN_AVAIL(s) ((s)->in.len - (s)->in_pos)
#define IN_PTR(s) ((s)->in.p + (s)->in_pos)
static int in_bad(z_streamp strm, const char *msg) { strm->msg = (char *)msg; strm->state->status = ST_BAD; return Z_DATA_ERROR; }

/* the wrapper in front of the deflate data (inflate.c:589-632 zlib, :596-602 and :634-759 gzip).  Z_OK: consumed, the body follows;
 * Z_BUF_ERROR: not all there yet (nothing consumed); Z_NEED_DICT; Z_DATA_ERROR. */
static int parse_header(z_streamp strm)
{
    struct internal_state *s = strm->state;
    const uint8_t *p = IN_PTR(s); const size_t n = IN_AVAIL(s);
    size_t skip;
    if (n < 2) return Z_BUF_ERROR;
    if ((s->wrap & 2) && p[0] == 31 && p[1] == 139) {
        if (n < 10) return Z_BUF_ERROR;
        if (p[2] != Z_DEFLATED) return in_bad(strm, "unknown compression method");
        if (p[3] & 0xe0) return in_bad(strm, "unknown header flags set");
        const unsigned flg = p[3];
        size_t xoff = 0, xlen = 0, noff = 0, coff = 0;
        skip = 10;
        if (flg & 4) { if (n < skip + 2) return Z_BUF_ERROR; xlen = (size_t)p[skip] | ((size_t)p[skip + 1] << 8); xoff = skip + 2; skip += 2 + xlen; if (n < skip) return Z_BUF_ERROR; }
        if (flg & 8) { noff = skip; while (skip < n && p[skip]) skip++; if (skip >= n) return Z_BUF_ERROR; skip++; }
        if (flg & 16) { coff = skip; while (skip < n && p[skip]) skip++; if (skip >= n) return Z_BUF_ERROR; skip++; }
        if (flg & 2) {
            if (n < skip + 2) return Z_BUF_ERROR;
            const uint32_t hc = (uint32_t)crc32(0, p, (uInt)skip) & 0xffffu;
            if (hc != ((uint32_t)p[skip] | ((uint32_t)p[skip + 1] << 8))) return in_bad(strm, "header crc mismatch");
            skip += 2;
        }
        gz_headerp g = s->gzhead;
        if (g != Z_NULL) { /* inflate.c:651-751 */
            g->text = (int)(flg & 1); g->time = (uLong)p[4] | ((uLong)p[5] << 8) | ((uLong)p[6] << 16) | ((uLong)p[7] << 24);
            g->xflags = p[8]; g->os = p[9]; g->hcrc = (int)((flg >> 1) & 1);
            if (flg & 4) { g->extra_len = (uInt)xlen; if (g->extra != Z_NULL) memcpy(g->extra, p + xoff, xlen < g->extra_max ? xlen : g->extra_max); }
            else g->extra = Z_NULL;
            if (flg & 8) { if (g->name != Z_NULL && g->name_max) { size_t l = strlen((const char *)p + noff) + 1; memcpy(g->name, p + noff, l < g->name_max ? l : g->name_max); } }
            else g->name = Z_NULL;
            if (flg & 16) { if (g->comment != Z_NULL && g->comm_max) { size_t l = strlen((const char *)p + coff) + 1; memcpy(g->comment, p + coff, l < g->comm_max ? l : g->comm_max); } }
            else g->comment = Z_NULL;
            g->done = 1;
        }
        s->gz = 1; strm->adler = 0; /* the running check of a gzip member is its CRC-32 (inflate.c:602) */
    } else {
        if (!(s->wrap & 1) || (((unsigned)p[0] << 8) + p[1]) % 31) return in_bad(strm, "incorrect header check");
        if ((p[0] & 15) != Z_DEFLATED) return in_bad(strm, "unknown compression method");
        if ((unsigned)(p[0] >> 4) + 8 > (unsigned)s->level) return in_bad(strm, "invalid window size");
        skip = 2;
        if (s->gzhead != Z_NULL) s->gzhead->done = -1; /* inflate.c:605 */
        if (p[1] & 0x20) { /* DICTID follows the header, inflate.c:617-627 */
            if (n < 6) return Z_BUF_ERROR;
            s->dictid = ((uint32_t)p[2] << 24) | ((uint32_t)p[3] << 16) | ((uint32_t)p[4] << 8) | p[5];
            if (!s->have_dict) { strm->adler = s->dictid; s->need_dict = 1; return Z_NEED_DICT; }
            skip = 6;
        }
        s->gz = 0;
    }
    s->in_pos += skip;
    s->mode = IN_BODY;
    return Z_OK;
}

/* Decode what has been collected behind in_pos as far as it goes.  Z_OK: progress or nothing to do yet; errors as usual. */
static int decode_some(z_streamp strm, size_t out_hint)
{
    struct internal_state *s = strm->state;
    const size_t n = IN_AVAIL(s);
    if (n == 0) return Z_OK;
    zgpu_engine *e = engine_get();
    if (!e) { strm->msg = g_engine_err; return Z_MEM_ERROR; }
    const uint8_t *src = IN_PTR(s); size_t srcn = n;
    uint8_t *shifted = NULL;
    if (s->prime_bits) { /* the stream as the decoder must see it: the primed bits, then the input */
        shifted = (uint8_t *)malloc(n + 5);
        if (!shifted) return Z_MEM_ERROR;
        uint64_t hold = s->prime_val; int nb = s->prime_bits; size_t o = 0;
        for (size_t i = 0; i < n; i++) { hold |= (uint64_t)src[i] << nb; nb += 8; while (nb >= 8) { shifted[o++] = (uint8_t)hold; hold >>= 8; nb -= 8; } }
        if (nb) shifted[o++] = (uint8_t)hold;
        src = shifted; srcn = o;
    }
    size_t cap = out_hint > srcn * 4 ? out_hint : srcn * 4;
    if (cap < 65536) cap = 65536;
    zgpu_inflate_result r;
    for (;;) {
        if (!buf_reserve(&s->out, cap)) { free(shifted); return Z_MEM_ERROR; }
        memset(&r, 0, sizeof r);
        /* what the window holds in front of this call's first byte (inflate.c:323-371 updatewindow): the preset dictionary, then the last 32 KiB
         * of what earlier calls produced -- a stream whose blocks reach back across a sync flush decodes in pieces as it arrives */
        const int first = s->dict.len != 0 && (s->have_dict || s->produced != 0);
        e = engine_checkout();
        int rc = zgpu_inflate_set_dictionary(e, first ? s->dict.p : NULL, first ? (uint32_t)s->dict.len : 0u);
        zgpu_inflate_set_checks(e, s->gz ? ZGPU_CHECK_CRC32 : ZGPU_CHECK_ADLER32); /* (the one check this stream's trailer holds: each is a pass over the output) */
        if (rc == ZGPU_OK) rc = zgpu_inflate_stream_host3(e, src, srcn, s->prime_bits ? 0u : s->skip_bits, ZGPU_INF_STREAM, s->out.p + s->out.len, cap, &r);
        if (first) zgpu_inflate_set_dictionary(e, NULL, 0);
        zgpu_inflate_set_checks(e, ZGPU_CHECK_ADLER32 | ZGPU_CHECK_CRC32);
        engine_checkin(e);
        if (rc == ZGPU_BUF_ERROR) { cap = r.out_bytes > cap ? (size_t)r.out_bytes : cap * 4; if (cap > ((size_t)1 << 40)) { free(shifted); return Z_MEM_ERROR; } continue; }
        free(shifted);
        if (rc == ZGPU_DATA_ERROR) return in_bad(strm, zgpu_inflate_message(r.error_msg));
        if (rc != ZGPU_OK) { strm->msg = (char *)zgpu_engine_error(e); return rc == ZGPU_MEM_ERROR ? Z_MEM_ERROR : Z_STREAM_ERROR; }
        break;
    }
    if (s->no_partial && !r.stream_end) { s->next_try = n + n / 4 + 1; return Z_OK; } /* (all or nothing) */
    if (r.out_bytes) {
        { /* the window for the next call: the last 32 KiB of dictionary + output so far */
            const uint8_t *fresh = s->out.p + s->out.len; const size_t nf = (size_t)r.out_bytes;
            if (nf >= 32768u) { s->dict.len = 0; if (!buf_put(&s->dict, fresh + (nf - 32768u), 32768u)) return Z_MEM_ERROR; }
            else {
                const size_t keep = s->dict.len < 32768u - nf ? s->dict.len : 32768u - nf;
                if (keep != s->dict.len) memmove(s->dict.p, s->dict.p + (s->dict.len - keep), keep);
                s->dict.len = keep;
                if (!buf_put(&s->dict, fresh, nf)) return Z_MEM_ERROR;
            }
        }
        s->out.len += r.out_bytes;
        if (s->gz) s->crc = crc_join(s->crc, r.crc32, r.out_bytes); else s->adler = adler_join(s->adler, r.adler32, r.out_bytes);
        s->produced += r.out_bytes;
    }
    size_t used = (size_t)r.in_used;
    if (s->prime_bits) { used = used * 8 > (size_t)s->prime_bits ? (used * 8 - (size_t)s->prime_bits + 7) / 8 : 0; if (used > n) used = n; s->prime_bits = 0; }
    s->skip_bits = r.stream_end ? 0u : r.in_used_bits;
    if (used >= 4) { const uint8_t *q = IN_PTR(s) + used - 4; s->at_marker = !r.stream_end && q[0] == 0 && q[1] == 0 && q[2] == 0xff && q[3] == 0xff; }
    s->in_pos += used;
    if (r.stream_end) s->mode = s->wrap ? IN_TRAIL : IN_DONE;
    else s->next_try = IN_AVAIL(s) + IN_AVAIL(s) / 4 + 1;
    return Z_OK;
}

EXPORT int inflate(z_streamp strm, int flush)
{
    if (strm == Z_NULL || strm->state == Z_NULL || strm->state->kind != KIND_INFLATE || strm->next_out == Z_NULL ||
        (strm->next_in == Z_NULL && strm->avail_in != 0))
        return Z_STREAM_ERROR;
    struct internal_state *s = strm->state;
    if (s->status == ST_BAD) return Z_DATA_ERROR;
    if (s->status == ST_DONE) return Z_STREAM_END;
    const uInt in0 = strm->avail_in, out0 = strm->avail_out;
    size_t absorbed = 0; /* what of this call's input went into s->in: only that can be handed back through next_in */
    /* input is taken in only while the caller's buffer can take what is decoded already: a reader with a small buffer keeps its input (as it does
     * with the reference, which stops reading when the output is full) and the library's backlog stays bounded */
    const size_t backlog = s->out.len - s->out_pos;
    if (s->mode != IN_DONE && strm->avail_in && !(backlog != 0 && backlog >= strm->avail_out)) {
        absorbed = strm->avail_in;
        if (s->in_pos && s->in_pos >= s->in.len / 2) { memmove(s->in.p, s->in.p + s->in_pos, s->in.len - s->in_pos); s->in.len -= s->in_pos; s->in_pos = 0; }
        if (!buf_put(&s->in, strm->next_in, strm->avail_in)) return Z_MEM_ERROR;
        strm->next_in += strm->avail_in; strm->total_in += strm->avail_in; strm->avail_in = 0;
    }
    if (absorbed > s->last_piece) s->last_piece = absorbed;
    if (s->mode == IN_HEAD) {
        int rc = parse_header(strm);
        if (rc == Z_NEED_DICT || rc == Z_DATA_ERROR) return rc;
    }
    if (s->mode == IN_BODY && !s->pending_err) {
        /* Decode when the caller says this is everything (Z_FINISH), has stopped supplying input, or enough has arrived since the
         * last look.  What is complete is taken; a stream that stops inside a block simply waits for more. */
        const size_t undelivered = s->out.len - s->out_pos;
        if (undelivered != 0 && undelivered >= strm->avail_out) { /* the caller's buffer is served from what is decoded already: nothing more is decoded
                                                                     (and held) until that is gone -- memory stays O(one call) for small readers */
        } else if (flush == Z_FINISH || in0 == 0 || IN_AVAIL(s) < 4096 || IN_AVAIL(s) >= s->next_try || (absorbed && absorbed < s->last_piece)) {
            /* (a piece shorter than the one before it is how the last slice of a file looks: decoding it in the call that brings it lets the bytes behind
             * the end of the stream go back to the caller, inflate.c:1114) */
            int rc = decode_some(strm, strm->avail_out);
            if (rc == Z_DATA_ERROR && s->out.len - s->out_pos != 0) { /* what was decoded before comes out first (inflate.c delivers as it goes) */
                s->status = ST_BUSY; s->pending_err = 1; s->pending_msg = strm->msg; strm->msg = Z_NULL;
            } else if (rc != Z_OK) return rc;
        }
    }
    if (s->mode == IN_TRAIL && !s->pending_err) { /* inflate.c:1077-1112 */
        const size_t need = s->gz ? 8 : 4;
        if (IN_AVAIL(s) >= need) {
            const uint8_t *t = IN_PTR(s);
            const char *bad = NULL;
            if (s->gz) {
                const uint32_t want = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
                const uint32_t wlen = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
                if (want != s->crc) bad = "incorrect data check";
                else if (wlen != (uint32_t)s->produced) bad = "incorrect length check";
            } else {
                const uint32_t want = ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | t[3];
                if (want != s->adler) bad = "incorrect data check";
            }
            if (bad) { s->pending_err = 1; s->pending_msg = bad; }
            else { s->in_pos += need; s->mode = IN_DONE; }
        }
    }
    if (s->mode == IN_DONE && absorbed) {
        /* inflate.c:1114: the rest of the input stays with the caller -- as far as it came with THIS call, and in the call that finds the end (later
         * calls, which only drain what is decoded, take nothing in and their next_in is left alone) */
        size_t back = IN_AVAIL(s);
        if (back > absorbed) back = absorbed;
        strm->next_in -= back; strm->avail_in += (uInt)back; strm->total_in -= back; s->in.len -= back;
    }
    deliver(strm);
    strm->adler = s->gz ? s->crc : s->adler;
    const int drained = s->out.len - s->out_pos == 0;
    if (s->pending_err && drained) return in_bad(strm, s->pending_msg);
    if (s->mode == IN_DONE && drained) { s->status = ST_DONE; return Z_STREAM_END; }
    if (flush == Z_FINISH) return Z_BUF_ERROR; /* not all input, or not enough room (inflate.c:1150-1151) */
    if (in0 == strm->avail_in && out0 == strm->avail_out) return Z_BUF_ERROR; /* no progress was possible (inflate.c:1150-1151): nothing absorbed, nothing delivered -- a caller that loops on Z_OK must not spin */
    return Z_OK;
}

/* Library-internal (zamd_gzio.c, zamd_infback.c): after Z_STREAM_END, the input this stream took in earlier calls that lies behind its
 * end and could not be handed back through next_in (inflate() only gives back what came with the call that reached the end).  In
 * stream order these bytes come BEFORE whatever next_in / avail_in hold now.  Valid until the next call on the stream. */
__attribute__((visibility("hidden"))) void zamd_inflate_rest(z_streamp strm, const unsigned char **p, size_t *n)
{
    *p = Z_NULL; *n = 0;
    if (strm == Z_NULL || strm->state == Z_NULL || strm->state->kind != KIND_INFLATE || strm->state->mode != IN_DONE) return;
    *p = IN_PTR(strm->state); *n = IN_AVAIL(strm->state);
}

/* Library-internal (zamd_gzio.c): output decoded and not yet delivered -- a reader drains it before it feeds more of the file */
__attribute__((visibility("hidden"))) size_t zamd_inflate_pending(z_streamp strm)
{
    if (strm == Z_NULL || strm->state == Z_NULL || strm->state->kind != KIND_INFLATE) return 0;
    return strm->state->out.len - strm->state->out_pos;
}

/* one byte of the search for 00 00 FF FF (inflate.c:1245-1265): `have` bytes of the pattern matched so far */
static int sync_step(int have, uint8_t c)
{
    if (c == (have < 2 ? 0 : 0xff)) return have + 1;
    return c ? 0 : 4 - have; /* a zero where FF was due: the zeros seen so far may still open the pattern */
}
/* inflate.c:1239-1303: skip to the next full-flush point (00 00 FF FF) and go on from there as a fresh decoder */
EXPORT int inflateSync(z_streamp strm)
{
    if (strm == Z_NULL || strm->state == Z_NULL || strm->state->kind != KIND_INFLATE) return Z_STREAM_ERROR;
    struct internal_state *s = strm->state;
    if (strm->avail_in == 0 && IN_AVAIL(s) == 0) return Z_BUF_ERROR;
    int have = s->sync_have;
    while (have < 4 && IN_AVAIL(s)) have = sync_step(have, s->in.p[s->in_pos++]); /* what was handed over before comes first */
    while (have < 4 && strm->avail_in) { have = sync_step(have, *strm->next_in++); strm->avail_in--; strm->total_in++; }
    if (have < 4) { s->sync_have = have; return Z_DATA_ERROR; }
    /* total_in / total_out stay; everything else starts over in the middle of the deflate data (inflate.c:1296-1301) */
    s->sync_have = 0; s->status = ST_BUSY; s->mode = IN_BODY; s->adler = 1; s->crc = 0; s->produced = 0; s->pending_err = 0; s->next_try = 0;
    s->have_dict = 0; s->need_dict = 0; s->at_marker = 1; strm->msg = Z_NULL; s->dict.len = 0; s->skip_bits = 0; /* (the window starts empty, inflate.c:1296 inflateReset) */
    return Z_OK;
}
/* inflate.c:1313-1321: "at the end of a block generated by Z_SYNC_FLUSH or Z_FULL_FLUSH" -- here: the decoder has consumed the
 * input up to and including a flush marker and nothing behind it */
EXPORT int inflateSyncPoint(z_streamp strm)
{
    if (strm == Z_NULL || strm->state == Z_NULL || strm->state->kind != KIND_INFLATE) return Z_STREAM_ERROR;
    const struct internal_state *s = strm->state;
    return s->mode == IN_BODY && s->at_marker && IN_AVAIL(s) == 0;
}

/* ======================================================================== one-shot wrappers (compress.c, uncompr.c) */
EXPORT int compress2(Bytef *dest, uLongf *destLen, const Bytef *source, uLong sourceLen, int level)
{
    /* compress.c:22-58.  The reference casts sourceLen to uInt and silently truncates inputs >= 4 GiB (compress.c:33);
     * here large inputs are fed in 3 GiB slices (a multiple of the chunk size, so the chunking is unchanged). */
    z_stream st; memset(&st, 0, sizeof st);
    int err = deflateInit_(&st, level, ZLIB_VERSION, (int)sizeof st);
    if (err != Z_OK) return err;
    const uLong slice = 0xC0000000ul;
    uLong left_in = sourceLen, left_out = *destLen;
    const Bytef *src = source;
    st.next_out = dest;
    for (;;) {
        const int last = left_in <= slice;
        const uInt take = last ? (uInt)left_in : (uInt)slice;
        st.next_in = (Bytef *)src; st.avail_in = take; src += take; left_in -= take;
        for (;;) { /* drain: the output window is 32 bits wide as well */
            const uInt room = left_out > 0xFFFFFFFFul ? 0xFFFFFFFFu : (uInt)left_out;
            st.avail_out = room;
            if (room == 0) { deflateEnd(&st); return Z_BUF_ERROR; }
            err = deflate(&st, last ? Z_FINISH : Z_NO_FLUSH);
            left_out -= room - st.avail_out;
            if (err == Z_STREAM_END) break;
            if (err != Z_OK) { deflateEnd(&st); return err; }
            if (st.avail_out != 0) break; /* everything produced so far has been delivered */
        }
        if (last) break;
    }
    if (err != Z_STREAM_END) { deflateEnd(&st); return err == Z_OK ? Z_BUF_ERROR : err; }
    *destLen = st.total_out;
    return deflateEnd(&st);
}
EXPORT int compress(Bytef *dest, uLongf *destLen, const Bytef *source, uLong sourceLen)
{
    return compress2(dest, destLen, source, sourceLen, Z_DEFAULT_COMPRESSION);
}
EXPORT int uncompress(Bytef *dest, uLongf *destLen, const Bytef *source, uLong sourceLen)
{
    z_stream st; memset(&st, 0, sizeof st);
    st.next_in = (Bytef *)source; st.avail_in = (uInt)sourceLen;
    if ((uLong)st.avail_in != sourceLen) return Z_BUF_ERROR; /* uncompr.c:36-38 */
    st.next_out = dest; st.avail_out = (uInt)*destLen;
    if ((uLong)st.avail_out != *destLen) return Z_BUF_ERROR;
    int err = inflateInit_(&st, ZLIB_VERSION, (int)sizeof st);
    if (err != Z_OK) return err;
    err = inflate(&st, Z_FINISH);
    if (err != Z_STREAM_END) {
        /* uncompr.c:50-56: a stream that ends early is a data error, an output buffer that is too small a buffer error.
         * This inflate() always consumes all input, so the two are told apart by whether the body could be decoded. */
        const int decoded = st.state != Z_NULL && st.state->mode == IN_DONE; /* (the whole stream was there: the output buffer is what ran out) */
        inflateEnd(&st);
        if (err == Z_NEED_DICT || (err == Z_BUF_ERROR && !decoded)) return Z_DATA_ERROR;
        return err == Z_OK ? Z_BUF_ERROR : err;
    }
    *destLen = st.total_out;
    return inflateEnd(&st);
}
