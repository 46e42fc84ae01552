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
 * deflate(): input is collected until at least one 64 KiB chunk is complete (or a flush / finish asks for everything),
 * the complete chunks go to the GPU in one call, the compressed bytes are handed out through next_out as space
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

/* ---- small utilities ---- */
static const char *const k_errmsg[10] = {"need dictionary", "stream end", "", "file error", "stream error", "data error",
                                         "insufficient memory", "buffer error", "incompatible version", ""};
#define ERR_MSG(code) ((char *)k_errmsg[2 - (code)])

EXPORT const char *zlibVersion(void) { return ZLIB_VERSION; }
EXPORT const char *zError(int err) { return k_errmsg[2 - err]; }
EXPORT uLong zlibCompileFlags(void) { return 0xa9; /* sizes of uInt/uLong/voidpf/z_off_t as the reference reports on LP64 */ }

static voidpf default_alloc(voidpf opaque, uInt items, uInt size) { (void)opaque; return malloc((size_t)items * size); }
static void default_free(voidpf opaque, voidpf p) { (void)opaque; free(p); }

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
static pthread_once_t g_crc_once = PTHREAD_ONCE_INIT;
static void crc_table_init(void)
{
    for (uint32_t i = 0; i < 256; i++) { uint32_t r = i; for (int k = 0; k < 8; k++) r = (r & 1u) ? (r >> 1) ^ 0xedb88320u : r >> 1; g_crc_table[i] = r; }
}
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

struct internal_state {
    int kind, status, wrap, level, strategy, last_flush;
    bytebuf in;       /* deflate: input not yet compressed;  inflate: compressed bytes not yet decoded */
    bytebuf out;      /* produced bytes not yet handed to the caller */
    size_t out_pos;   /* first undelivered byte of out */
    int trailer_done; /* deflate: Adler trailer already appended */
    int any_block;    /* deflate: at least one chunk has been emitted */
    uint32_t adler;   /* Adler-32 of the uncompressed data that went through the GPU (deflate) / was produced (inflate) */
    uint32_t crc;     /* the same for CRC-32 (gzip wrapper, wrap == 2) */
    int decoded;      /* inflate: the body has been decoded */
    size_t next_try;  /* inflate: do not re-try a decode before this many bytes have been collected */
    bytebuf dict;     /* preset dictionary: deflate, the bytes the window receives until the first chunk is out; inflate, as set */
    int dict_pending; /* deflate: the first chunk has not been compressed yet and starts behind the dictionary */
    int need_dict, have_dict; /* inflate: the header asked for one / one has been set */
    uint32_t dictid;  /* Adler-32 of the dictionary (header field, deflate.c:646-649, inflate.c:623-627) */
};

static uLong bound_for(uLong n)
{
    /* compress.c:75-79 gives n + n/4096 + n/16384 + 11; independent 64 KiB chunks of incompressible data cost up to 30 bytes
     * each (5 stored-block headers + the flush marker), more than that formula allows from the second chunk on */
    uLong ref = n + (n >> 12) + (n >> 14) + 11, chunks = n ? (n + CHUNK - 1) / CHUNK : 1, ours = n + 30 * chunks + 6 + 6;
    return ref > ours ? ref : ours;
}
EXPORT uLong compressBound(uLong sourceLen) { return bound_for(sourceLen); }
EXPORT uLong deflateBound(z_streamp strm, uLong sourceLen)
{
    const int gz = strm != Z_NULL && strm->state != Z_NULL && strm->state->kind == KIND_DEFLATE && strm->state->wrap == 2;
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
    free(s->in.p); free(s->out.p); free(s->dict.p);
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
    /* served subset: see include/zamd_zlib.h */
    if (method != Z_DEFLATED || windowBits != 15 || memLevel != 8 || strategy < 0 || strategy > Z_FIXED || level < 0 || level > 9) return Z_STREAM_ERROR;
    if (!engine_get()) { strm->msg = g_engine_err; return Z_MEM_ERROR; }
    struct internal_state *s = state_new(strm, KIND_DEFLATE);
    if (!s) return Z_MEM_ERROR;
    strm->state = s;
    s->wrap = wrap; s->level = level; s->strategy = strategy;
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
    s->in.len = 0; s->out.len = 0; s->out_pos = 0; s->trailer_done = 0; s->any_block = 0; s->dict.len = 0; s->dict_pending = 0;
    s->status = s->wrap ? ST_INIT : ST_BUSY; s->last_flush = Z_NO_FLUSH;
    s->adler = 1; s->crc = 0; strm->adler = s->wrap == 2 ? 0 : 1; /* deflate.c:374-378 */
    return Z_OK;
}
EXPORT int deflateEnd(z_streamp strm)
{
    if (strm == Z_NULL || strm->state == Z_NULL || strm->state->kind != KIND_DEFLATE) return Z_STREAM_ERROR;
    const int busy = strm->state->status == ST_BUSY;
    state_free(strm);
    return busy ? Z_DATA_ERROR : Z_OK; /* deflate.c:886 */
}
/* deflate.c:315-354.  Served before the first byte of input (the reference also lets a raw stream replace its window later). */
EXPORT int deflateSetDictionary(z_streamp strm, const Bytef *d, uInt n)
{
    if (strm == Z_NULL || strm->state == Z_NULL || strm->state->kind != KIND_DEFLATE || d == Z_NULL) return Z_STREAM_ERROR;
    struct internal_state *s = strm->state;
    if (s->wrap == 2 || (s->wrap == 1 && s->status != ST_INIT) || s->any_block || s->in.len != 0 || strm->total_in != 0) return Z_STREAM_ERROR;
    if (s->wrap) strm->adler = adler32(strm->adler, d, n); /* becomes the DICTID of the header */
    if (n < 3) return Z_OK;                                /* shorter than MIN_MATCH: nothing to match against */
    const uInt keep = n > 32506u ? 32506u : n;            /* MAX_DIST: the tail of the dictionary */
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
    if ((level != s->level || strategy != s->strategy) && s->in.len != 0 && s->status != ST_FINISH) {
        rc = run_chunks(strm, s->in.p, s->in.len, 0);
        s->in.len = 0;
    }
    s->level = level; s->strategy = strategy;
    return rc;
}

/* level 0 needs no match finder or entropy coder: stored blocks are framing.  One chunk = the bytes the reference's
 * deflate_stored emits for a fresh stream of that chunk (deflate.c:1390-1439): blocks of at most 65531 bytes, the rest,
 * then the flush marker or, on the last chunk, the final bit. */
static int stored_block(bytebuf *out, const uint8_t *src, size_t len, int last)
{
    const uint8_t h[5] = {(uint8_t)(last ? 1 : 0), (uint8_t)len, (uint8_t)(len >> 8), (uint8_t)~len, (uint8_t)(~len >> 8)};
    return buf_put(out, h, 5) && buf_put(out, src, len);
}
static int stored_chunks(bytebuf *out, const uint8_t *src, size_t n, int final)
{
    size_t nchunks = n ? (n + CHUNK - 1) / CHUNK : 1;
    for (size_t k = 0; k < nchunks; k++) {
        const size_t lo = k * CHUNK, len = n - lo < CHUNK ? n - lo : CHUNK;
        const int last = final && k + 1 == nchunks;
        int ok;
        /* what deflate_stored emits for a fresh stream of `len` bytes: a block is cut at 65531 bytes (pending_buf_size - 5,
         * deflate.c:1397-1402,1420-1427) or as soon as it reaches MAX_DIST = 32506 bytes (:1431-1434), and the flush at the end
         * then closes whatever is left -- an EMPTY block after a MAX_DIST cut that took everything */
        if (len > 65531) ok = stored_block(out, src + lo, 65531, 0) && stored_block(out, src + lo + 65531, len - 65531, last);
        else if (len >= 32506) ok = stored_block(out, src + lo, len, 0) && stored_block(out, src + lo, 0, last);
        else ok = stored_block(out, src + lo, len, last);
        if (!ok) return 0;
        if (!last) { static const uint8_t marker[5] = {0, 0, 0, 0xff, 0xff}; if (!buf_put(out, marker, 5)) return 0; }
    }
    return 1;
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
            if (!stored_chunks(&s->out, src, take, first_final)) return Z_MEM_ERROR;
            s->adler = (uint32_t)adler32(s->adler, src, (uInt)take);
        } else {
            zgpu_engine *e = engine_get();
            if (!e) { strm->msg = g_engine_err; return Z_MEM_ERROR; }
            bytebuf w = {0};
            if (!buf_put(&w, s->dict.p, s->dict.len) || !buf_put(&w, src, take)) { free(w.p); return Z_MEM_ERROR; }
            const uint64_t cap = zgpu_deflate_bound(w.len, CHUNK);
            if (!buf_reserve(&s->out, cap)) { free(w.p); return Z_MEM_ERROR; }
            zgpu_deflate_params p = {s->level, CHUNK, first_final ? ZGPU_F_FINAL : 0u, ZGPU_LZ_AUTO, s->strategy, 0};
            zgpu_deflate_result r;
            pthread_mutex_lock(&g_lock);
            int rc = zgpu_deflate_dict_chunk_host(e, w.p, (uint32_t)w.len, (uint32_t)s->dict.len, &p, s->out.p + s->out.len, cap, &r);
            pthread_mutex_unlock(&g_lock);
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
        if (!stored_chunks(&s->out, src, n, final)) return Z_MEM_ERROR;
        for (size_t o = 0; o < n; o += 0x40000000u) {
            size_t m = n - o < 0x40000000u ? n - o : 0x40000000u;
            if (s->wrap == 2) s->crc = (uint32_t)crc32(s->crc, src + o, (uInt)m); else s->adler = (uint32_t)adler32(s->adler, src + o, (uInt)m);
        }
        s->any_block = 1; return Z_OK;
    }
    zgpu_engine *e = engine_get();
    if (!e) { strm->msg = g_engine_err; return Z_MEM_ERROR; }
    uint64_t cap = zgpu_deflate_bound(n, CHUNK);
    if (!buf_reserve(&s->out, cap)) return Z_MEM_ERROR;
    zgpu_deflate_params p = {s->level, CHUNK, (final ? ZGPU_F_FINAL : 0u) | (s->wrap == 2 ? ZGPU_F_CRC32 : 0u), ZGPU_LZ_AUTO, s->strategy, 0};
    zgpu_deflate_result r;
    pthread_mutex_lock(&g_lock);
    int rc = zgpu_deflate_host(e, src, n, &p, s->out.p + s->out.len, cap, NULL, &r);
    pthread_mutex_unlock(&g_lock);
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

    if (s->status == ST_INIT && s->wrap == 2) { /* gzip header without a gz_header, deflate.c:578-596; OS_CODE 3 as the reference builds on this host */
        const uint8_t h[10] = {31, 139, 8, 0, 0, 0, 0, 0, (uint8_t)(s->level == 9 ? 2 : (s->strategy >= Z_HUFFMAN_ONLY || s->level < 2) ? 4 : 0), 3};
        if (!buf_put(&s->out, h, 10)) return Z_MEM_ERROR;
        s->status = ST_BUSY;
    }
    if (s->status == ST_INIT) { /* zlib header, deflate.c:625-649 */
        unsigned hdr = (Z_DEFLATED + (7u << 4)) << 8, lf = (s->strategy >= Z_HUFFMAN_ONLY || s->level < 2) ? 0 : s->level < 6 ? 1 : s->level == 6 ? 2 : 3;
        hdr |= lf << 6;
        if (s->dict_pending) hdr |= 0x20; /* PRESET_DICT, deflate.c:641 */
        hdr += 31 - hdr % 31;
        uint8_t h[6] = {(uint8_t)(hdr >> 8), (uint8_t)hdr, (uint8_t)(strm->adler >> 24), (uint8_t)(strm->adler >> 16), (uint8_t)(strm->adler >> 8), (uint8_t)strm->adler};
        if (!buf_put(&s->out, h, s->dict_pending ? 6 : 2)) return Z_MEM_ERROR;
        strm->adler = 1; /* deflate.c:650 */
        s->status = ST_BUSY;
    }
    if (s->out.len - s->out_pos != 0) { /* deflate.c:757-768 */
        deliver(strm);
        if (strm->avail_out == 0) { s->last_flush = -1; return Z_OK; }
    } else if (strm->avail_in == 0 && flush <= old_flush && flush != Z_FINISH) { /* deflate.c:774-777 */
        strm->msg = ERR_MSG(Z_BUF_ERROR); return Z_BUF_ERROR;
    }
    if (s->status == ST_FINISH && strm->avail_in != 0) { strm->msg = ERR_MSG(Z_BUF_ERROR); return Z_BUF_ERROR; }

    if (strm->avail_in != 0 || s->in.len != 0 || (flush != Z_NO_FLUSH && s->status != ST_FINISH)) {
        /* take the caller's input (read_buf, deflate.c:956-981) */
        const uint8_t *src = strm->next_in; size_t n = strm->avail_in;
        strm->next_in += n; strm->total_in += n; strm->avail_in = 0;
        int rc = Z_OK;
        if (flush == Z_NO_FLUSH && s->dict_pending) {
            /* the first chunk behind a dictionary is 65536 - |dictionary| bytes: collect until it is complete */
            const size_t room = CHUNK - s->dict.len;
            if (!buf_put(&s->in, src, n)) rc = Z_MEM_ERROR;
            else if (s->in.len >= room) {
                rc = run_chunks(strm, s->in.p, room, 0);
                memmove(s->in.p, s->in.p + room, s->in.len - room); s->in.len -= room;
                if (rc == Z_OK && s->in.len >= CHUNK) {
                    size_t whole = s->in.len - s->in.len % CHUNK;
                    rc = run_chunks(strm, s->in.p, whole, 0);
                    memmove(s->in.p, s->in.p + whole, s->in.len - whole); s->in.len -= whole;
                }
            }
        } else if (flush == Z_NO_FLUSH) {
            /* only complete chunks are compressed now; the tail waits for more input */
            if (s->in.len == 0 && n >= CHUNK) { size_t whole = n - n % CHUNK; rc = run_chunks(strm, src, whole, 0); src += whole; n -= whole; }
            if (rc == Z_OK && n) { if (!buf_put(&s->in, src, n)) rc = Z_MEM_ERROR; }
            if (rc == Z_OK && s->in.len >= CHUNK) {
                size_t whole = s->in.len - s->in.len % CHUNK;
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
    if (!engine_get()) { strm->msg = g_engine_err; return Z_MEM_ERROR; }
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
    s->in.len = 0; s->out.len = 0; s->out_pos = 0; s->decoded = 0; s->status = ST_BUSY; s->adler = 1; s->crc = 0; s->next_try = 0; s->dict.len = 0; s->need_dict = 0; s->have_dict = 0;
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

/* Decode everything collected in s->in.  Returns Z_OK when decoded, Z_BUF_ERROR when the stream is visibly incomplete,
 * Z_DATA_ERROR / Z_MEM_ERROR otherwise.  out_hint: how much room the caller said it has (sizes the first attempt). */
static int decode_all(z_streamp strm, size_t out_hint)
{
    struct internal_state *s = strm->state;
    const uint8_t *p = s->in.p; size_t n = s->in.len, skip = 0;
    int gz = 0;
    if ((s->wrap & 2) && n >= 2 && p[0] == 31 && p[1] == 139) { /* gzip header, inflate.c:596-602, 634-759 */
        gz = 1;
        if (n < 10) return Z_BUF_ERROR;
        if (p[2] != Z_DEFLATED) { strm->msg = (char *)"unknown compression method"; return Z_DATA_ERROR; }
        if (p[3] & 0xe0) { strm->msg = (char *)"unknown header flags set"; return Z_DATA_ERROR; }
        const unsigned flg = p[3];
        skip = 10;
        if (flg & 4) { if (n < skip + 2) return Z_BUF_ERROR; skip += 2 + ((size_t)p[skip] | ((size_t)p[skip + 1] << 8)); if (n < skip) return Z_BUF_ERROR; }
        if (flg & 8) { while (skip < n && p[skip]) skip++; if (skip >= n) return Z_BUF_ERROR; skip++; }
        if (flg & 16) { while (skip < n && p[skip]) skip++; if (skip >= n) return Z_BUF_ERROR; skip++; }
        if (flg & 2) {
            if (n < skip + 2) return Z_BUF_ERROR;
            const uint32_t hc = (uint32_t)crc32(0, p, (uInt)skip) & 0xffffu;
            if (hc != ((uint32_t)p[skip] | ((uint32_t)p[skip + 1] << 8))) { strm->msg = (char *)"header crc mismatch"; return Z_DATA_ERROR; }
            skip += 2;
        }
        if (n < skip + 8 + 2) return Z_BUF_ERROR;
    } else if (s->wrap) { /* inflate.c:589-632 */
        if (n < 2) return Z_BUF_ERROR;
        if (!(s->wrap & 1) || (((unsigned)p[0] << 8) + p[1]) % 31) { strm->msg = (char *)"incorrect header check"; return Z_DATA_ERROR; }
        if ((p[0] & 15) != Z_DEFLATED) { strm->msg = (char *)"unknown compression method"; return Z_DATA_ERROR; }
        if ((unsigned)(p[0] >> 4) + 8 > (unsigned)s->level) { strm->msg = (char *)"invalid window size"; return Z_DATA_ERROR; }
        skip = 2;
        if (p[1] & 0x20) { /* DICTID follows the header, inflate.c:617-627 */
            if (n < 6) return Z_BUF_ERROR;
            s->dictid = ((uint32_t)p[2] << 24) | ((uint32_t)p[3] << 16) | ((uint32_t)p[4] << 8) | p[5];
            if (!s->have_dict) { strm->adler = s->dictid; s->need_dict = 1; return Z_NEED_DICT; }
            skip = 6;
        }
        if (n < skip + 4 + 2) return Z_BUF_ERROR;
    }
    const size_t tail = gz ? 8 : s->wrap ? 4 : 0;
    size_t body = n - skip - tail;
    if (body == 0) return Z_BUF_ERROR;
    zgpu_engine *e = engine_get();
    if (!e) { strm->msg = g_engine_err; return Z_MEM_ERROR; }
    size_t cap = out_hint > body * 4 ? out_hint : body * 4;
    if (cap < 65536) cap = 65536;
    for (;;) {
        s->out.len = 0; s->out_pos = 0;
        if (!buf_reserve(&s->out, cap)) return Z_MEM_ERROR;
        zgpu_inflate_result r;
        memset(&r, 0, sizeof r);
        pthread_mutex_lock(&g_lock);
        int rc = zgpu_inflate_set_dictionary(e, s->have_dict ? s->dict.p : NULL, s->have_dict ? (uint32_t)s->dict.len : 0u);
        if (rc == ZGPU_OK) rc = zgpu_inflate_stream_host(e, p + skip, body, s->out.p, cap, &r);
        if (s->have_dict) zgpu_inflate_set_dictionary(e, NULL, 0);
        pthread_mutex_unlock(&g_lock);
        if (rc == ZGPU_BUF_ERROR) { cap = r.out_bytes > cap ? (size_t)r.out_bytes : cap * 4; if (cap > ((size_t)1 << 40)) return Z_MEM_ERROR; continue; }
        if (rc == ZGPU_DATA_ERROR) {
            /* a body that stops inside a block is what a not-yet-complete stream looks like */
            const char *m = zgpu_inflate_message(r.error_msg);
            if (strcmp(m, "segment ends inside a block") == 0) return Z_BUF_ERROR;
            strm->msg = (char *)m; return Z_DATA_ERROR;
        }
        if (rc != ZGPU_OK) { strm->msg = (char *)zgpu_engine_error(e); return rc == ZGPU_MEM_ERROR ? Z_MEM_ERROR : Z_STREAM_ERROR; }
        s->out.len = r.out_bytes; s->adler = gz ? r.crc32 : r.adler32; /* strm->adler is the CRC for a gzip stream (inflate.c:602) */
        break;
    }
    if (gz) { /* inflate.c:1083-1112: CRC-32, then the length mod 2^32, both least significant byte first */
        const uint8_t *t = p + n - 8;
        const uint32_t want = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        const uint32_t wlen = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
        if (want != s->adler) { strm->msg = (char *)"incorrect data check"; s->out.len = 0; return Z_DATA_ERROR; }
        if (wlen != (uint32_t)s->out.len) { strm->msg = (char *)"incorrect length check"; s->out.len = 0; return Z_DATA_ERROR; }
    } else if (s->wrap) { /* inflate.c:1077-1098 */
        const uint8_t *t = p + n - 4;
        uint32_t want = ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | t[3];
        if (want != s->adler) { strm->msg = (char *)"incorrect data check"; s->out.len = 0; return Z_DATA_ERROR; }
    }
    s->decoded = 1;
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
    if (!s->decoded) {
        if (strm->avail_in) {
            if (!buf_put(&s->in, strm->next_in, strm->avail_in)) return Z_MEM_ERROR;
            strm->next_in += strm->avail_in; strm->total_in += strm->avail_in; strm->avail_in = 0;
        }
        /* Decode when the caller says this is everything (Z_FINISH) or has stopped supplying input.  A stream that is
         * not complete yet reads as "ends inside a block" and simply waits for more input. */
        if (flush == Z_FINISH || in0 == 0 || s->in.len < 4096 || s->in.len >= s->next_try) {
            int rc = decode_all(strm, strm->avail_out);
            if (rc == Z_BUF_ERROR) s->next_try = s->in.len + s->in.len / 4 + 1;
            if (rc == Z_DATA_ERROR || rc == Z_MEM_ERROR || rc == Z_NEED_DICT || rc == Z_STREAM_ERROR) { if (rc == Z_DATA_ERROR) s->status = ST_BAD; return rc; }
            if (rc == Z_BUF_ERROR) { /* incomplete: inflate.c:1150-1151 */
                if (flush == Z_FINISH || (in0 == 0 && out0 == strm->avail_out)) return Z_BUF_ERROR;
                return Z_OK;
            }
        }
    }
    if (s->decoded) {
        deliver(strm);
        strm->adler = s->adler;
        if (s->out.len - s->out_pos == 0) { s->status = ST_DONE; return Z_STREAM_END; }
        if (flush == Z_FINISH) return Z_BUF_ERROR; /* output space ran out (inflate.c:1150-1151) */
    }
    return Z_OK;
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
        const int decoded = st.state != Z_NULL && st.state->decoded;
        inflateEnd(&st);
        if (err == Z_NEED_DICT || (err == Z_BUF_ERROR && !decoded)) return Z_DATA_ERROR;
        return err == Z_OK ? Z_BUF_ERROR : err;
    }
    *destLen = st.total_out;
    return inflateEnd(&st);
}
